#!/bin/bash
# On the GPU box: matrix-core counters of every kernel of the headline rollout (bench.py --no-extras) and of the
# training iteration (--workload train), per dispatch.  One SQ pass + GRBM (MI355X_MICROARCH.md: 8 SQ slots, GRBM
# independent); --kernel-trace only beside --pmc, the program directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mfma
rm -rf $O && mkdir -p $O
pass() { # name workload counters...
  n=$1; w=$2; shift; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -- python3 $R/bench.py --no-extras --no-cpu-baseline --workload $w --steps 5 --warmup 2 > $O/$n.log 2>&1
  f=$(find $O/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/rocpd_pmc.py $f sf:: > $O/$n.txt 2>&1; fi
  rm -rf $O/$n
}
pass rollout_mfma rollout SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pass train_mfma train SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
wc -l $O/*.txt; head -60 $O/rollout_mfma.txt | cut -c1-150
