#!/bin/bash
# On the GPU box: kernel-trace stats of the default bench under the caller's environment
# (usage: VAR=1 bash tools/prof_env.sh TAG [bench args]) -> gpurun_out/prof_TAG_stats.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_$TAG && mkdir -p $O/prof_$TAG
rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -- python3 $R/bench.py "$@" > $O/prof_$TAG.log 2>&1
f=$(find $O/prof_$TAG -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $f > $O/prof_${TAG}_stats.txt 2>&1
rm -rf $O/prof_$TAG
tail -1 $O/prof_$TAG.log | cut -c1-110
head -12 $O/prof_${TAG}_stats.txt | cut -c1-20,110-160
