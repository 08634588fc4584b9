#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of the speaker sweep with ONE and with TWO streams (120 minibatches
# each): does the second stream's first kernel (gather_path_actions_kernel) wait?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for ns in 1 2; do
  rm -rf $O/prof_sweep_$ns && mkdir -p $O/prof_sweep_$ns
  SF_SWEEP_STREAMS=$ns SF_SWEEP_BATCHES=120 rocprofv3 --kernel-trace --stats -d $O/prof_sweep_$ns -- python3 $R/tools/speaker_sweep_streams.py > $O/prof_sweep_$ns.log 2>&1
  f=$(find $O/prof_sweep_$ns -name "*.db" | head -1)
  python3 $R/tools/rocpd_stats.py $f > $O/prof_sweep_${ns}_stats.txt 2>&1
  rm -rf $O/prof_sweep_$ns
  echo "== streams $ns"; grep "streams" $O/prof_sweep_$ns.log | tail -2; head -12 $O/prof_sweep_${ns}_stats.txt | cut -c1-150
done
