cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tn && mkdir -p $R/gpurun_out/prof_tn
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tn -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload train > $R/gpurun_out/prof_tn.log 2>&1
f=$(find $R/gpurun_out/prof_tn -name "*.db" | head -1)
python3 $R/tools/rocpd_calls.py $f gemm_tn | tail -11
rm -rf $R/gpurun_out/prof_tn
