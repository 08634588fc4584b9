#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats summary of the DEFAULT bench command
# (hipGraph replay) -> gpurun_out/prof_default_stats.txt + the bench JSON line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_default && mkdir -p $O/prof_default
rocprofv3 --kernel-trace --stats -d $O/prof_default -- python3 $R/bench.py "$@" > $O/prof_default.log 2>&1
f=$(find $O/prof_default -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $f > $O/prof_default_stats.txt 2>&1
rm -rf $O/prof_default
grep "\"metric\"" $O/prof_default.log | tail -1 > $O/prof_default_bench.json
head -30 $O/prof_default_stats.txt | cut -c1-160
