#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the speaker's training iteration (tools/speaker_train_time.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_spktrain && mkdir -p $O/prof_spktrain
rocprofv3 --kernel-trace --stats -d $O/prof_spktrain -- python3 $R/tools/speaker_train_time.py > $O/prof_spktrain.log 2>&1
f=$(find $O/prof_spktrain -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $f > $O/prof_spktrain_stats.txt 2>&1
rm -rf $O/prof_spktrain
tail -1 $O/prof_spktrain.log
head -32 $O/prof_spktrain_stats.txt | cut -c1-150
