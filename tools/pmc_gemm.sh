cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc2 && mkdir -p $R/gpurun_out/pmc2
REPS=40 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc2/a -- python3 $R/tools/gemm_only.py > $R/gpurun_out/pmc2/a.log 2>&1
REPS=40 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $R/gpurun_out/pmc2/b -- python3 $R/tools/gemm_only.py > $R/gpurun_out/pmc2/b.log 2>&1
REPS=40 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc2/c -- python3 $R/tools/gemm_only.py > $R/gpurun_out/pmc2/c.log 2>&1
for d in a b; do f=$(find $R/gpurun_out/pmc2/$d -name "*.db" | head -1); python3 $R/tools/rocpd_pmc.py $f gemm > $R/gpurun_out/pmc2/$d.txt 2>&1; done
f=$(find $R/gpurun_out/pmc2/c -name "*.db" | head -1); python3 $R/tools/rocpd_stats.py $f > $R/gpurun_out/pmc2/c.txt 2>&1
find $R/gpurun_out/pmc2 -name "*.db" -delete
tail -3 $R/gpurun_out/pmc2/a.log
