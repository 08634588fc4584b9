#!/bin/bash
# On the GPU box: PMC passes over the persistent decode launch (tools/mega_only.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mega
rm -rf $O && mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt
pass() { # name counters...
  n=$1; shift
  REPS=4 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -- python3 $R/tools/mega_only.py > $O/$n.log 2>&1
  f=$(find $O/$n -name "*.db" | head -1); python3 $R/tools/rocpd_pmc.py $f mega_kernel > $O/$n.txt 2>&1; rm -rf $O/$n
}
pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
pass sq1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM
pass sq3 SQ_WAIT_ANY SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cat $O/names.txt; for n in ic sq1 sq2 sq3 fetch write; do echo "== $n"; cat $O/$n.txt; tail -1 $O/$n.log; done
