#!/bin/bash
# tools/r06_final.sh — the round's closing measurements in ONE gpurun call (outputs under gpurun_out/r06/, copied to profiles/r06/ by hand):
#   the GPU suite, the default bench line, rocprofv3 kernel stats of the headline and of the legs, PMC instruction counts of the FM kernel, the small-bank tables
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -q > /tmp/suite_full.txt 2>&1; grep -aE "[0-9]+ (passed|failed)|^FAILED|^ERROR" /tmp/suite_full.txt | tail -6 > $O/final_gpu_suite.txt   # (device printf of the stamp tests lands behind pytest's summary: not `tail`)
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.txt 2>&1
python bench.py > $O/final_bench_default.json 2> $O/final_bench_default.err; cp gpurun_out/bench_full.json $O/final_bench_default_full.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_head -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-configs > $GRAFT_REPO_ROOT/$O/final_bench_profiled.json 2>/tmp/prof_head.err )
python tools/kernel_summary.py /tmp/prof_head klg_render_sub2a_x2 375 20 375 $O/final_bench_kernel_stats.csv > $O/final_bench_kernel_summary.json 2>&1
( cd /tmp && export TMPDIR=/tmp && KLG_BENCH_PMC=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_legs -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/final_bench_legs_profiled.json 2>/tmp/prof_legs.err )
python tools/kernel_summary.py /tmp/prof_legs klg_fx_pingpong_x 0 0 1 $O/final_bench_legs_kernel_stats.csv > /dev/null 2>&1
python tools/pmc_any.py "PatchFM<4>" "SQ_INSTS_VALU,SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY" -- python $GRAFT_REPO_ROOT/tools/fm_leg.py 131072 > $O/pmc_fm4.json 2>&1
python tools/pingpong_leg_repeat.py 5 > $O/pingpong_leg_repeat.txt 2>&1
python tools/small_banks.py > $O/small_banks.jsonl 2>&1
python tools/recorded_small_banks.py > $O/recorded_small_banks.jsonl 2>&1
ls -la $O
