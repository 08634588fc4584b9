#!/bin/bash
# On the GPU box: kernel timeline of the LAST training iteration (two-stream backward) -> gpurun_out/train_timeline.txt
# (start offset, duration, queue per dispatch).  `bash tools/prof_train_timeline.sh graph [back_us span_us]`: the
# iteration as hipGraph replays (tools/train_graph_timeline.py); `eager`: issued launch by launch (train_timeline.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
prog=train_timeline.py; [ "$1" = graph ] && prog=train_graph_timeline.py
rm -rf $O/prof_tl && mkdir -p $O/prof_tl
rocprofv3 --kernel-trace -d $O/prof_tl -- python3 $R/tools/$prog > $O/prof_tl.log 2>&1
f=$(find $O/prof_tl -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $f ${2:-5200} ${3:-5200} > $O/train_timeline_$1.txt 2>&1
rm -rf $O/prof_tl
tail -2 $O/prof_tl.log; wc -l $O/train_timeline_$1.txt
