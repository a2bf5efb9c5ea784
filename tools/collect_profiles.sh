#!/bin/bash
# copy the outputs of tools/run_round_measurements.sh from gpurun_out/<dir> into profiles/ (tracked): collect_profiles.sh <dir> <rNN>
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/$1; R=$2
mkdir -p profiles/${R}_pmc profiles/${R}_decode_pmc
for n in FETCH_SIZE WRITE_SIZE TCC_HIT_sum SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU; do
  f=$(ls -t $O/pmc_$n/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_pmc/${n}_counter_collection.csv
done
for n in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32; do
  f=$(ls -t $O/dpmc_$n/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_decode_pmc/${n}_counter_collection.csv
done
# identity of the kernels these passes counted (bench.py reports roofline.traffic only if the tree's sampler still has it); taken on the GPU box
# by run_round_measurements.sh from the very tree that was profiled
[ -f $O/kernel_id.json ] && cp $O/kernel_id.json profiles/${R}_pmc/kernel_id.json && cp $O/kernel_id.json profiles/${R}_decode_pmc/kernel_id.json
mkdir -p profiles/${R}_diffonly_pmc
for n in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32; do
  f=$(ls -t $O/npmc_$n/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" > profiles/${R}_diffonly_pmc/${n}_per_kernel.csv
done
f=$(ls -t $O/den_stats/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_diffonly_kernel_stats.csv
f=$(ls -t $O/den_staged_stats/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_diffonly_staged_kernel_stats.csv
[ -f $O/den_perf.txt ] && grep -v libdrm $O/den_perf.txt > profiles/${R}_diffonly_perf.txt
mkdir -p profiles/${R}_fp32x_pmc
for n in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES SQ_INSTS_MFMA; do
  f=$(ls -t $O/xpmc_$n/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_fp32x_pmc/${n}_counter_collection.csv
done
[ -f $O/kernel_id.json ] && cp $O/kernel_id.json profiles/${R}_fp32x_pmc/kernel_id.json
f=$(ls -t $O/x_stats/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${R}_fp32x_kernel_stats.csv
[ -f $O/fp32x_perf.txt ] && grep -v libdrm $O/fp32x_perf.txt > profiles/${R}_fp32x_perf.txt
[ -f $O/phase8x.txt ] && grep -v libdrm $O/phase8x.txt > profiles/${R}_k_sample8x_phase_timeline.txt
mkdir -p profiles/${R}_audio_pmc
for n in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES SQ_INSTS_MFMA; do
  f=$(ls -t $O/apmc_$n/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" > profiles/${R}_audio_pmc/${n}_per_kernel.csv
done
f=$(ls -t $O/stats/*/*_kernel_stats.csv | head -1); cp "$f" profiles/${R}_bench_kernel_stats.csv
f=$(ls -t $O/decode_stats/*/*_kernel_stats.csv | head -1); cp "$f" profiles/${R}_decode_kernel_stats.csv
f=$(ls -t $O/audio_stats/*/*_kernel_stats.csv | head -1); cp "$f" profiles/${R}_audio_kernel_stats.csv
f=$(ls -t $O/audio_one_stats/*/*_kernel_stats.csv | head -1); cp "$f" profiles/${R}_audio_one_encoder_kernel_stats.csv
grep -v libdrm $O/gemm_bench.txt > profiles/${R}_gemm_bench.txt
grep -v libdrm $O/audio_perf.txt > profiles/${R}_audio_perf.txt
tail -1 $O/bench.json > profiles/${R}_bench_line.json
tail -1 $O/train_bench.json > profiles/${R}_train_bench_line.json
[ -f $O/train_bench_eager.json ] && tail -1 $O/train_bench_eager.json > profiles/${R}_train_bench_line_eager_layers.json
[ -f $O/train_profile.txt ] && cp $O/train_profile.txt profiles/${R}_train_step_torch_profile.txt
[ -f $O/train_gemm_time.txt ] && grep -v libdrm $O/train_gemm_time.txt > profiles/${R}_train_gemm_per_call.txt
[ -f $O/train_bench_nograph.json ] && tail -1 $O/train_bench_nograph.json > profiles/${R}_train_bench_line_no_graph.json
[ -f $O/multigpu_preflight_world1.txt ] && grep multigpu_preflight $O/multigpu_preflight_world1.txt > profiles/${R}_multigpu_preflight_world1.json
[ -f $O/attend32_probe.txt ] && cp $O/attend32_probe.txt profiles/${R}_attend32_probe_runs.txt
[ -f $O/train_profile_vendor_gemm.txt ] && cp $O/train_profile_vendor_gemm.txt profiles/${R}_train_step_torch_profile_vendor_gemm.txt
[ -f $O/mfma_f32_rate_probe.txt ] && cp $O/mfma_f32_rate_probe.txt profiles/${R}_mfma_f32_rate_probe.txt
[ -f $O/mfma_valu_coissue_probe.txt ] && cp $O/mfma_valu_coissue_probe.txt profiles/${R}_mfma_valu_coissue_probe.txt
[ -f $O/mfma_valu_samewave_probe.txt ] && cp $O/mfma_valu_samewave_probe.txt profiles/${R}_mfma_valu_samewave_probe.txt
[ -f $O/train_batch_sweep.jsonl ] && python3 -c "import json,sys; [print(d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], round(d['samples_per_s'])) for d in map(json.loads, open(sys.argv[1]))]" $O/train_batch_sweep.jsonl > profiles/${R}_train_batch_sweep_latest.txt
[ -f $O/torch_eager_baseline.txt ] && cp $O/torch_eager_baseline.txt profiles/${R}_torch_eager_baseline.txt
[ -f $O/train_host_vs_device.txt ] && cp $O/train_host_vs_device.txt profiles/${R}_train_host_vs_device.txt
f=$(ls -t $O/train_stats/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/probes/train_host/kernel_stats_per_iteration.py "$f" 40 60 > profiles/${R}_train_kernels_per_iteration.txt
[ -f $O/train_attn_perf.txt ] && grep -v libdrm $O/train_attn_perf.txt > profiles/${R}_train_attention_per_call.txt
[ -f $O/train_bench_vendor_attn.json ] && tail -1 $O/train_bench_vendor_attn.json > profiles/${R}_train_bench_line_vendor_attention.json
[ -f $O/dec_perf.txt ] && grep -v libdrm $O/dec_perf.txt > profiles/${R}_trans_dec_sampler_perf.txt
grep -v libdrm $O/phase8.txt > profiles/${R}_k_sample8_phase_timeline.txt
grep -v libdrm $O/decode_perf.txt > profiles/${R}_decode_perf.txt
grep -v libdrm $O/batch_sweep.txt > profiles/${R}_batch_sweep.txt
cp $O/drift.json profiles/${R}_bf16_vs_fp32_drift.json
cp $O/pytest.txt profiles/${R}_gpu_pytest_tail.txt
