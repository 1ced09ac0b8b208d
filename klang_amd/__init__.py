"""klang_amd — MI355X-native implementation of klang's per-block signal-graph evaluation path.

Python here is only the thin host binding used by tests and bench.py: `SynthBank` / `FxBank` wrap the
C-ABI of include/klang_mi355.h (libklang_mi355.so: C++ host logic + hand-written gfx950 kernels).
There is no CPU rendering path anywhere in this package.
"""
from ._lib import KlangError, lib, LIB_PATH  # noqa: F401
from .bank import SynthBank, FxBank, EventScript, PATCH_IDS, init  # noqa: F401
from .shard import ShardedFxBank, ShardedSynthBank, shard_range, owner_of  # noqa: F401
