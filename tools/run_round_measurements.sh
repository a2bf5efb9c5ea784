set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench8.json 2> gpurun_out/bench8.err; tail -2 gpurun_out/bench8.err
export TMPDIR=/tmp
rm -rf gpurun_out/prof8 && mkdir -p gpurun_out/prof8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof8/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-audio > gpurun_out/prof8/stats.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/prof8/pmc_$n -- python3 tools/run_sample_once.py 256 bf16 2 > gpurun_out/prof8/pmc_$n.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof8/audio_stats -- python3 tools/gpu_audio_perf.py 32 > gpurun_out/prof8/audio_stats.log 2>&1
timeout 100 python tools/gpu_phase_profile8.py 256 > gpurun_out/prof8/phase8.txt 2>&1
find gpurun_out/prof8 -name "*.csv" | head -30
