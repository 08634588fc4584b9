cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_w && mkdir -p $R/gpurun_out/prof_w
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload train > $R/gpurun_out/prof_w.log 2>&1
f=$(find $R/gpurun_out/prof_w -name "*.db" | head -1)
n=$(python3 -c "import sqlite3;print(sqlite3.connect('$f').execute('select count(*) from kernels').fetchone()[0])")
echo total $n
python3 $R/tools/rocpd_window.py $f $((n - 560)) 60
rm -rf $R/gpurun_out/prof_w
