#!/bin/bash
# On the GPU box: kernel timeline of a window of the training iteration (usage: prof_timeline.sh BACK_US SPAN_US)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tl && mkdir -p $R/gpurun_out/prof_tl
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --workload train > $R/gpurun_out/prof_tl.log 2>&1
f=$(find $R/gpurun_out/prof_tl -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $f $1 $2
rm -rf $R/gpurun_out/prof_tl
