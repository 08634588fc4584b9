cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/r02_g_bench_default.json 2> gpurun_out/r02_g_bench_default.err
tail -c 600 gpurun_out/r02_g_bench_default.json | head -c 300; echo
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g -o r02g -- python3 $R/bench.py --no-extras > $R/gpurun_out/prof_g.log 2>&1
f=$(find $R/gpurun_out/prof_g -name "*kernel_stats.csv" | head -1)
echo stats $f
[ -n "$f" ] && head -25 "$f" > $R/gpurun_out/r02_g_bench_default_kernel_stats.txt
rm -rf $R/gpurun_out/prof_g
head -12 $R/gpurun_out/r02_g_bench_default_kernel_stats.txt
