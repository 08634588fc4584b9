#!/bin/bash
# On the GPU box: kernel timeline of the two-stream forward (usage: prof_two_stream.sh BACK_US SPAN_US [env...])
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ts && mkdir -p $R/gpurun_out/prof_ts
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ts -- python3 $R/tools/two_stream_timeline.py > $R/gpurun_out/prof_ts.log 2>&1
f=$(find $R/gpurun_out/prof_ts -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $f $1 $2
rm -rf $R/gpurun_out/prof_ts
