#!/bin/bash
# On the GPU box: PMC passes over the persistent encoder launch (tools/encoder_only.py runs the follower
# encoder at B=100, T=80 a few times).  Separate passes per counter group (MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_persist
rm -rf $O && mkdir -p $O
pass() { # name counters...
  n=$1; shift
  REPS=6 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -- python3 $R/tools/encoder_only.py > $O/$n.log 2>&1
  f=$(find $O/$n -name "*.db" | head -1); python3 $R/tools/rocpd_pmc.py $f enc_persist_kernel > $O/$n.txt 2>&1; rm -rf $O/$n
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass sq1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES
cat $O/fetch.txt $O/write.txt $O/tcc.txt $O/sq1.txt
tail -2 $O/sq1.log
