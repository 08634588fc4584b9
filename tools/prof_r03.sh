#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of the training iteration (tools/train_profile.py) and of the
# speaker batch (tools/speaker_bench.py), straight on `python3 <script>` after `--`.
#   -> gpurun_out/r03_train_kernel_stats.txt, gpurun_out/r03_speaker_kernel_stats.txt (+ the scripts' own logs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for what in train speaker; do
  if [ $what = train ]; then script=$R/tools/train_profile.py; else script=$R/tools/speaker_bench.py; fi
  rm -rf $O/prof_$what && mkdir -p $O/prof_$what
  rocprofv3 --kernel-trace --stats -d $O/prof_$what -- python3 $script > $O/r03_${what}_run.log 2>&1
  f=$(find $O/prof_$what -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $f > $O/r03_${what}_kernel_stats.txt 2>&1
  rm -rf $O/prof_$what
done
head -40 $O/r03_train_kernel_stats.txt | cut -c1-175
head -24 $O/r03_speaker_kernel_stats.txt | cut -c1-175
tail -32 $O/r03_train_run.log
