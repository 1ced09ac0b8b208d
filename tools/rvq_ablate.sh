#!/bin/bash
# tools/rvq_ablate.sh — where do klg_fx_reverb_q's written bytes come from?  Builds the library with one group of the kernel's stores compiled out at a time
# (KLG_RVQ_ABLATE: 1 FilteredDelay pieces, 2 early-line pieces, 4 output block, 8 record write-back; results are then wrong — measurement only), into
# klang_amd/_rvq_ablate_<mask>.so (run HERE, where hipcc is), then on the GPU box:  tools/rvq_ablate.sh run  -> WRITE_SIZE / requests / kernel time per variant.
R="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" != run ]; then
	for m in ${RVQ_MASKS:-0 1 2 4 8 15}; do bash $R/klang_amd/csrc/build.sh -DKLG_RVQ_ABLATE=$m > /dev/null 2>&1 && cp $R/klang_amd/libklang_mi355.so $R/klang_amd/_rvq_ablate_$m.so; done
	bash $R/klang_amd/csrc/build.sh > /dev/null 2>&1          # the product library again
	exit 0
fi
cp $R/klang_amd/libklang_mi355.so /tmp/_keep.so
for m in ${RVQ_MASKS:-0 1 2 4 8 15}; do
	cp $R/klang_amd/_rvq_ablate_$m.so $R/klang_amd/libklang_mi355.so
	echo "{\"ablate\": $m, \"pmc\": $(python $R/tools/pmc_any.py klg_fx_reverb_q WRITE_SIZE TCC_EA0_WRREQ_sum,TCC_EA0_WRREQ_64B_sum TCP_TCC_WRITE_REQ_sum -- python $R/tools/fx_scale.py reverb 4096 2>/dev/null | tail -1), \"time\": $(KLG_FX_REVERB_EARLY=0 python $R/tools/fx_scale.py reverb 4096 | tail -1)}"
done
cp /tmp/_keep.so $R/klang_amd/libklang_mi355.so
