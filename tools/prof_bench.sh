#!/bin/bash
# usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench args...]  -> gpurun_out/prof_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
rm -rf $R/gpurun_out/prof_$tag && mkdir -p $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra "$@" > $R/gpurun_out/prof_$tag.log 2>&1
f=$(find $R/gpurun_out/prof_$tag -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $f > $R/gpurun_out/prof_$tag.txt 2>&1
rm -rf $R/gpurun_out/prof_$tag
tail -2 $R/gpurun_out/prof_$tag.log | cut -c1-300
