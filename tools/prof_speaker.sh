cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/speaker_bench.py 2>&1 | tail -2
rm -rf $R/gpurun_out/prof_spk && mkdir -p $R/gpurun_out/prof_spk
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_spk -- python3 $R/tools/speaker_bench.py > $R/gpurun_out/prof_spk.log 2>&1
f=$(find $R/gpurun_out/prof_spk -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $f > $R/gpurun_out/prof_spk.txt 2>&1
rm -rf $R/gpurun_out/prof_spk
head -16 $R/gpurun_out/prof_spk.txt | cut -c1-150
