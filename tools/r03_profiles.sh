#!/bin/bash
# tools/r03_profiles.sh — everything profiles/r03_* is made from, in one gpurun call (run from the repo root on the GPU box).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
KLG_BENCH_PMC=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-configs > $O/bench_profiled.json 2> $O/bench_profiled.err
python $R/tools/kernel_summary.py $O/stats klg_render_sub2a_x2 375 20 375 > $O/kernel_summary.json 2>> $O/bench_profiled.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
# every config's leg under the kernel trace: rocprofv3's average per kernel beside the legs' kernel_ms_mean (dispatch-attached HIP events) — they must agree
KLG_BENCH_PMC=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_legs -- python $R/bench.py --no-cpu-baseline > $O/bench_legs_profiled.json 2> $O/bench_legs_profiled.err
cp $(find $O/stats_legs -name "*kernel_stats.csv" | head -1) $O/bench_legs_kernel_stats.csv 2>/dev/null
rm -rf $O/stats_legs
python $R/tools/bench_all.py --cpu-budget 2 > $O/bench_all.json 2> $O/bench_all.err
python $R/tools/fx_ablate.py > $O/fx_sizes.jsonl 2>&1
python $R/tools/pingpong_steady.py > $O/pingpong_steady.jsonl 2>&1
python $R/tools/reverb_modes.py 1024 4096 8192 > $O/reverb_modes.jsonl 2>&1
SQ1=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR
SQ2=SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_SCA,SQ_LDS_BANK_CONFLICT
# the FINAL synth kernels of configs 3 and 5
python $R/tools/pmc_any.py "PatchFM<4>" $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/bench_all.py --only cfg5 --cpu-budget 0.1 > $O/pmc_fm4.json 2>&1
python $R/tools/pmc_any.py "klg_render_supersaw_pairs" $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/bench_all.py --only cfg3 --cpu-budget 0.1 > $O/pmc_supersaw_pairs.json 2>&1
python $R/tools/pmc_any.py klg_fx_reverb_q $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE,TCC_HIT_sum,TCC_MISS_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum,TCP_TCC_WRITE_REQ_sum -- python $R/tools/fx_scale.py reverb 4096 > $O/pmc_reverb_q_4096.json 2>&1
python $R/tools/pmc_any.py klg_fx_pingpong_x $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/fx_scale.py pingpong 4096 > $O/pmc_pingpong_4096.json 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fx_stats -- python $R/tools/fx_scale.py pingpong 4096 16384 65536 reverb 1024 4096 8192 16384 > $O/fx_scale.jsonl 2> $O/fx_scale.err
cp $(find $O/fx_stats -name "*kernel_stats.csv" | head -1) $O/fx_kernel_stats.csv 2>/dev/null
rm -rf $O/stats $O/fx_stats
ls -la $O
