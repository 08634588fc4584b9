#!/bin/bash
# On the GPU box: HBM-side traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of EVERY kernel of the headline
# rollout (bench.py --no-extras), per dispatch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bench
rm -rf $O && mkdir -p $O
pass() { # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -- python3 $R/bench.py --no-extras --steps 5 --warmup 2 > $O/$n.log 2>&1
  f=$(find $O/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/rocpd_pmc.py $f sf:: > $O/$n.txt 2>&1; fi
  rm -rf $O/$n
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cat $O/fetch.txt $O/write.txt | head -120
