#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_beam && mkdir -p $O/prof_beam
rocprofv3 --kernel-trace --stats -d $O/prof_beam -- python3 $R/tools/beam_step_profile.py > $O/prof_beam.log 2>&1
f=$(find $O/prof_beam -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $f > $O/prof_beam_stats.txt 2>&1
rm -rf $O/prof_beam
tail -1 $O/prof_beam.log
head -24 $O/prof_beam_stats.txt | cut -c1-150
