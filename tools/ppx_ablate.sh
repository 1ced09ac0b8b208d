#!/bin/bash
# tools/ppx_ablate.sh — what paces klg_fx_pingpong_x's steps?  Builds the library with one part of the pipeline compiled out at a time (KLG_PPX_ABLATE:
# 1 no DC filter chain, 2 no output stores, 4 no ring requests, 8 no audio stage; results are then wrong — measurement only) into
# klang_amd/_ppx_ablate_<mask>.so (run HERE, where hipcc is), then on the GPU box:  tools/ppx_ablate.sh run  -> time per block and variant.
R="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" != run ]; then
	for m in ${PPX_MASKS:-0 1 2 3 4 8 15}; do bash $R/klang_amd/csrc/build.sh -DKLG_PPX_ABLATE=$m ${PPX_EXTRA} > /dev/null 2>&1 && cp $R/klang_amd/libklang_mi355.so $R/klang_amd/_ppx_ablate_$m.so; done
	bash $R/klang_amd/csrc/build.sh > /dev/null 2>&1          # the product library again
	exit 0
fi
cp $R/klang_amd/libklang_mi355.so /tmp/_keep.so
for m in ${PPX_MASKS:-0 1 2 3 4 8 15}; do
	cp $R/klang_amd/_ppx_ablate_$m.so $R/klang_amd/libklang_mi355.so
	echo "{\"ablate\": $m, \"time\": $(python $R/tools/pingpong_steady.py ${PPX_K:-4096} | tail -1)}"
done
cp /tmp/_keep.so $R/klang_amd/libklang_mi355.so
