#!/bin/bash
# On the GPU box: per-queue busy time and gaps of the LAST training iteration (usage: prof_train_tl.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ttl && mkdir -p $R/gpurun_out/prof_ttl
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ttl -- python3 $R/tools/train_timeline.py > $R/gpurun_out/prof_ttl.log 2>&1
f=$(find $R/gpurun_out/prof_ttl -name "*.db" | head -1)
python3 $R/tools/rocpd_phases.py $f
rm -rf $R/gpurun_out/prof_ttl
