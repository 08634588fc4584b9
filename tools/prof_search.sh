#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of tools/search_profile.py (state_factored_search K = 40 on the full
# world: 146 replays of the 64-state step graph per minibatch) -> gpurun_out/prof_search_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_search && mkdir -p $O/prof_search
rocprofv3 --kernel-trace --stats -d $O/prof_search -- python3 $R/tools/search_profile.py "$@" > $O/prof_search.log 2>&1
f=$(find $O/prof_search -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $f > $O/prof_search_stats.txt 2>&1
rm -rf $O/prof_search
head -45 $O/prof_search_stats.txt | cut -c1-170
