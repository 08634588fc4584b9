#!/bin/bash
# On the GPU box: PMC passes over the decoder-LSTM-gate product alone (tools/gemm_only.py).
# Separate passes per counter group (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm
rm -rf $O && mkdir -p $O
pass() { # name counters...
  n=$1; shift
  REPS=40 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -- python3 $R/tools/gemm_only.py > $O/$n.log 2>&1
  f=$(find $O/$n -name "*.db" | head -1); python3 $R/tools/rocpd_pmc.py $f gemm > $O/$n.txt 2>&1; rm -rf $O/$n
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
pass sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
REPS=40 rocprofv3 --kernel-trace --stats -d $O/st -- python3 $R/tools/gemm_only.py > $O/st.log 2>&1
f=$(find $O/st -name "*.db" | head -1); python3 $R/tools/rocpd_stats.py $f > $O/stats.txt 2>&1; rm -rf $O/st
cat $O/fetch.txt $O/write.txt $O/sq1.txt $O/sq2.txt; head -5 $O/stats.txt
