#!/bin/bash
# tools/r04_profiles.sh — everything profiles/r04_* is made from, in one gpurun call (run from the repo root on the GPU box).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
KLG_BENCH_PMC=0 KLG_BENCH_PMC_FX=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-configs > $O/bench_profiled.json 2> $O/bench_profiled.err
python $R/tools/kernel_summary.py $O/stats klg_render_sub2a_x2 375 20 375 > $O/bench_kernel_summary.json 2>> $O/bench_profiled.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/stats
# every config's leg under the kernel trace: rocprofv3's average per kernel beside the legs' kernel_ms_mean (a PingPong span is ONE launch of `blocks_per_span` blocks)
KLG_BENCH_PMC=0 KLG_BENCH_PMC_FX=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_legs -- python $R/bench.py --no-cpu-baseline > $O/bench_legs_profiled.json 2> $O/bench_legs_profiled.err
cp $(find $O/stats_legs -name "*kernel_stats.csv" | head -1) $O/bench_legs_kernel_stats.csv 2>/dev/null
rm -rf $O/stats_legs
# effect banks: spans, recorded (staged / one lane per instance) against hand-written, dials off their defaults
python $R/tools/fx_span_bench.py 4096 > $O/fx_spans.jsonl 2>&1
python $R/tools/fx_span_bench.py 16384 pingpong >> $O/fx_spans.jsonl 2>&1
python $R/tools/pingpong_recorded_bench.py 4096 8192 16384 65536 > $O/pingpong_recorded.jsonl 2>&1
python $R/tools/reverb_recorded_bench.py 256 1024 4096 > $O/reverb_recorded.jsonl 2>&1
KLG_BENCH_SKIP_PLAIN=1 python $R/tools/reverb_recorded_bench.py 16384 >> $O/reverb_recorded.jsonl 2>&1
python $R/tools/pingpong_dials_bench.py 4096 > $O/pingpong_dials.jsonl 2>&1
( python $R/tools/staged_stamp.py pingpong 4096 2>&1 | grep stamps | tail -1; KLG_FX_STAGED_PIPE=0 python $R/tools/staged_stamp.py pingpong 4096 2>&1 | grep stamps | tail -1; python $R/tools/staged_stamp.py reverb 4096 2>&1 | grep stamps | tail -1 ) > $O/staged_stamps.txt
( cd $R && python -m pytest tests/test_gpu_fx_staged.py -m gpu -q -s -k which_example 2>&1 | grep -E "staged:|one lane per instance:|passed|failed" ) > $O/staged_forms.txt
python $R/tools/bench_all.py --cpu-budget 2 > $O/bench_all.json 2> $O/bench_all.err
python $R/tools/staged_examples_bench.py 4096 65536 2>&1 | grep -v amdgpu.ids > $O/staged_examples.jsonl
SQ1=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR
SQ2=SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_SCA,SQ_LDS_BANK_CONFLICT
# the staged kernels of the two recorded config-4 effects, and the hand-written ones as they run now (PingPong: 64-block spans)
python $R/tools/pmc_any.py klg_fx_staged $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/staged_sweep.py pingpong 4096 16,32 > $O/pmc/pmc_staged_pingpong_4096.json 2>&1
python $R/tools/pmc_any.py klg_fx_staged $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/staged_sweep.py reverb 4096 16,32 > $O/pmc/pmc_staged_reverb_4096.json 2>&1
FX_SPAN_BLOCKS=64 python $R/tools/pmc_any.py klg_fx_pingpong_x $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/fx_span_bench.py 4096 pingpong > $O/pmc/pmc_pingpong_4096_spans64.json 2>&1
FX_SPAN_BLOCKS=1 python $R/tools/pmc_any.py klg_fx_reverb_q $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/fx_span_bench.py 4096 reverb > $O/pmc/pmc_reverb_q_4096.json 2>&1
python $R/tools/pmc_any.py "PatchFM<4>" $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/bench_all.py --only cfg5 --cpu-budget 0.1 > $O/pmc/pmc_fm4.json 2>&1
python $R/tools/pmc_any.py "klg_render_supersaw_pairs" $SQ1 $SQ2 FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -- python $R/tools/bench_all.py --only cfg3 --cpu-budget 0.1 > $O/pmc/pmc_supersaw_pairs.json 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fx_stats -- python $R/tools/fx_span_bench.py 4096 > $O/fx_spans_profiled.jsonl 2> $O/fx_spans_profiled.err
cp $(find $O/fx_stats -name "*kernel_stats.csv" | head -1) $O/fx_kernel_stats.csv 2>/dev/null
rm -rf $O/fx_stats
ls -la $O $O/pmc
