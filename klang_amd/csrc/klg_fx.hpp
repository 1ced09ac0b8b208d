// klang_amd/csrc/klg_fx.hpp — effect banks (Stereo::Effect instances: PingPong.k, Reverb.k).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/klang_mi355.h"
