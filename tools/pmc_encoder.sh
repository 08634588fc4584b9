#!/bin/bash
# On the GPU box: PMC passes over the encoder recurrent step (tools/encoder_only.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_enc
rm -rf $O && mkdir -p $O
pass() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -- python3 $R/tools/encoder_only.py > $O/$n.log 2>&1
  f=$(find $O/$n -name "*.db" | head -1); python3 $R/tools/rocpd_pmc.py $f lstm_step > $O/$n.txt 2>&1; rm -rf $O/$n
}
pass fetch FETCH_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
cat $O/fetch.txt $O/tcc.txt $O/tcp.txt
tail -3 $O/tcp.log
