#!/bin/bash
# copy the newest outputs of tools/run_round_measurements.sh from gpurun_out/ into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
for n in FETCH_SIZE WRITE_SIZE TCC_HIT_sum SQ_WAVE_CYCLES SQ_INSTS_MFMA; do
  cp "$(ls -t gpurun_out/prof8/pmc_$n/runc/*_counter_collection.csv | head -1)" profiles/r01_pmc/${n}_counter_collection.csv
done
cp "$(ls -t gpurun_out/prof8/stats/runc/*_kernel_stats.csv | head -1)" profiles/r01_bench_kernel_stats.csv
tail -1 gpurun_out/bench8.json > profiles/r01_bench_line.json
grep -v libdrm gpurun_out/prof8/phase8.txt > profiles/r01_k_sample8_phase_timeline.txt
cp "$(ls -t gpurun_out/prof8/audio_stats/runc/*_kernel_stats.csv | head -1)" profiles/r01_audio_kernel_stats.csv
