#!/bin/bash
# klang_amd/csrc/build.sh — builds klang_amd/libklang_mi355.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off is REQUIRED: the reference path is bit-stable only without FMA contraction.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libklang_mi355.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -Bsymbolic: the library binds its own C++ inline functions (the graph-program parser of include/klang_mi355_graph.h is
# compiled into host programs too) instead of letting a host executable's copies interpose on them.
# -fno-slp-vectorize: where two independent fp32 operations should share a v_pk_* instruction the kernels say so (f2 vectors); what the SLP vectorizer packs
# on its own comes with the moves that build and split the pairs (a dependent v_pk_add_f32 itself waits 8 cycles like a plain one: tools/calib/issue_latency.hip):
# SuperSaw renders 2 - 3 % faster without it, nothing slower.
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wl,-Bsymbolic \
    -Wall -Wno-unused-function "$@" "$HERE/klg_api.hip" -o "$OUT"
echo "built $OUT"
# the deadline-measurement host (klang_amd/host/klang_deadline.cpp): a C++ real-time loop over the C-ABI, for bench.py's deadline legs
"$HIPCC" -O2 -std=c++17 -Wall "$HERE/../host/klang_deadline.cpp" -I"$HERE/../../include" -L"$HERE/.." -lklang_mi355 -Wl,-rpath,'$ORIGIN/..' -o "$HERE/../host/klang_deadline"
echo "built $HERE/../host/klang_deadline"
