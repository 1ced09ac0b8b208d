// oracle/ref/prelude.h — TEST INFRASTRUCTURE (reference build recipe), not product code.
//
// The reference header (/root/reference/klang.h, v0.7.8) only defines its
// platform macros for wasm / Apple / Win32 (klang.h:20-42) and omits a few
// standard includes on Linux.  This prelude supplies exactly those and nothing
// else, so the genuine header compiles unmodified with ROCm clang++.
// It is included *before* <klang.h> by every oracle/ref/ref_*.cpp.
#pragma once
#include <cstring>
#include <climits>
#include <cstdint>
#include <chrono>
#include <tuple>
#include <cmath>
#define THREAD_LOCAL thread_local
#define SQRT ::sqrt
#define SQRTF ::sqrtf
#define ABS ::abs
#define FABS ::fabsf
using std::isnan;
using std::isinf;
