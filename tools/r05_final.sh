R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R; (time python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/gpu_suite_head.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
cd /tmp && export TMPDIR=/tmp
KLG_BENCH_FULL=$O/bench_default_full.json python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
KLG_BENCH_PMC=0 KLG_BENCH_PMC_FX=0 KLG_BENCH_FULL=$O/bench_profiled_full.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-configs > $O/bench_profiled.json 2> $O/bench_profiled.err
python $R/tools/kernel_summary.py $O/stats klg_render_sub2a_x2 375 20 375 > $O/bench_kernel_summary.json 2>> $O/bench_profiled.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/stats
tail -3 $O/gpu_suite_head.txt; cat $O/smoke.txt | tail -2; cat $O/bench_kernel_summary.json | head -c 600
