#!/bin/bash
# tools/r05_profiles.sh — what profiles/r05/* is made from, in one gpurun call (run from the repo root on the GPU box).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
KLG_BENCH_FULL=$O/bench_default_full.json python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
# the bench command under the kernel trace: rocprofv3's average of the dominant kernel over the timed launches must agree with roofline.kernel_ms
KLG_BENCH_PMC=0 KLG_BENCH_PMC_FX=0 KLG_BENCH_FULL=$O/bench_profiled_full.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-configs > $O/bench_profiled.json 2> $O/bench_profiled.err
python $R/tools/kernel_summary.py $O/stats klg_render_sub2a_x2 375 20 375 > $O/bench_kernel_summary.json 2>> $O/bench_profiled.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/stats
# every config's leg under the kernel trace (a PingPong span is ONE launch of `blocks_per_span` blocks)
KLG_BENCH_PMC=0 KLG_BENCH_PMC_FX=0 KLG_BENCH_FULL=$O/bench_legs_profiled_full.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_legs -- python $R/bench.py --no-cpu-baseline > $O/bench_legs_profiled.json 2> $O/bench_legs_profiled.err
cp $(find $O/stats_legs -name "*kernel_stats.csv" | head -1) $O/bench_legs_kernel_stats.csv 2>/dev/null
rm -rf $O/stats_legs
# Noise notes: this library and round 4's (rand() on the host), same box
for V in 1024 16384 262144; do python $R/tools/noise_bench.py $V 2>&1 | tail -1; done > $O/noise_bench.jsonl
for V in 1024 16384; do KLANG_MI355_LIB=$R/tools/_ab/klang_amd/libklang_mi355_r04.so python $R/tools/noise_bench.py $V 20 2>&1 | tail -1; done > $O/noise_bench_r04_library.jsonl
# SuperSaw: the three kernels at config 3's size and the sample-parallel one at scale; counters before / after
for L in 3 2 0; do KLG_SUPERSAW_LANES=$L python $R/tools/supersaw_bench.py 16384 | tail -1; done > $O/supersaw_kernels_16384.jsonl
for V in 65536 524288; do python $R/tools/supersaw_bench.py $V | tail -1; done > $O/supersaw_sp_scale.jsonl
SQ1=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR
SQ2=SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_SCA,SQ_LDS_BANK_CONFLICT
KLG_SUPERSAW_LANES=3 python $R/tools/pmc_any.py klg_render_supersaw_sp $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE -- python $R/tools/supersaw_bench.py 16384 > $O/pmc/pmc_supersaw_sp_16384.json 2>&1
KLG_SUPERSAW_LANES=2 python $R/tools/pmc_any.py klg_render_supersaw_pairs $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE -- python $R/tools/supersaw_bench.py 16384 > $O/pmc/pmc_supersaw_pairs_16384.json 2>&1
for V in 1024 2048 4096; do for SP in 1 0; do KLG_SUB2A_SP=$SP python $R/tools/sub2a_bench.py $V | tail -1; done; done > $O/sub2a_small_banks.jsonl
python $R/tools/pmc_any.py klg_render_sub2a_sp $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE -- python $R/tools/sub2a_bench.py 1024 > $O/pmc/pmc_sub2a_sp_1024.json 2>&1
python $R/tools/pmc_any.py klg_rand_fill $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE -- python $R/tools/noise_bench.py 16384 > $O/pmc/pmc_rand_fill_16384.json 2>&1
ls -la $O $O/pmc
# PingPong: the driver-shaped leg five times in one process (run-to-run spread), spans at rest and with converging smoothers, traffic of a 64-block span
cd $R
python tools/pingpong_leg_repeat.py 5 > $O/pingpong_leg_repeat.txt 2>/dev/null
(FX_SPAN_BLOCKS=1,25,64,256 python tools/fx_span_bench.py 4096 pingpong; FX_SPAN_FRESH=2 FX_SPAN_BLOCKS=25 python tools/fx_span_bench.py 4096 pingpong; FX_SPAN_BLOCKS=64 python tools/fx_span_bench.py 8192 pingpong; FX_SPAN_FRESH=2 FX_SPAN_BLOCKS=25 python tools/fx_span_bench.py 8192 pingpong) 2>/dev/null | grep effect > $O/pingpong_spans.jsonl
cd /tmp
FX_SPAN_BLOCKS=64 python $R/tools/pmc_any.py klg_fx_pingpong_x FETCH_SIZE WRITE_SIZE -- python $R/tools/fx_span_bench.py 4096 pingpong > $O/pmc/pmc_pingpong_4096_spans64.json 2>&1
ls -la $O $O/pmc
