#!/bin/bash
# On the GPU box: kernel timeline of the END of the last training iteration (usage: prof_train_tail.sh [back_us] [span_us])
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ttl && mkdir -p $R/gpurun_out/prof_ttl
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ttl -- python3 $R/tools/train_timeline.py > $R/gpurun_out/prof_ttl.log 2>&1
f=$(find $R/gpurun_out/prof_ttl -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $f ${1:-1500} ${2:-1500}
rm -rf $R/gpurun_out/prof_ttl
