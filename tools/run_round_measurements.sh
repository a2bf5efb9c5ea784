# Round measurements on the GPU box (run through gpurun from the repo root): tests, bench lines, rocprofv3 kernel stats of
# the bench command, PMC passes (one counter group per run, no tracing domains mixed in), decode / audio / phase timelines.
# Usage: bash tools/run_round_measurements.sh <out dir under gpurun_out>     then tools/collect_profiles.sh <out dir> <rNN>
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-meas}
export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
python tools/kernel_id.py > $O/kernel_id.json     # what the PMC passes below count (collect_profiles.sh files it next to them)
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 300 python bench.py --config train --steps 60 --warmup 15 > $O/train_bench.json 2> $O/train_bench.err
AMUSE_TRAIN_FUSED=0 timeout 300 python bench.py --config train --steps 60 --warmup 15 > $O/train_bench_eager.json 2> $O/train_bench_eager.err
timeout 300 python tools/gpu_train_profile.py 32 $O/train_profile.txt > /dev/null 2>&1
timeout 200 python tools/gpu_train_attn_perf.py > $O/train_attn_perf.txt 2>&1
timeout 300 python bench.py --config train --steps 60 --warmup 15 --no-graph > $O/train_bench_nograph.json 2> /dev/null
for b in 32 64 128 256 512; do timeout 300 python bench.py --config train --train-batch $b --steps 30 --warmup 8 2>/dev/null | tail -1; done > $O/train_batch_sweep.jsonl   # samples/s against the batch (config 4 is 32)
timeout 300 python tools/probes/train_host/gemm_time.py > $O/train_gemm_time.txt 2>&1          # the library's own GEMMs against torch's, per call
timeout 300 python tools/multigpu_preflight.py --gpus 1 > $O/multigpu_preflight_world1.txt 2>&1   # RCCL world of one: init, barrier, all-reduce, all-gather, bitwise shard check
(cd tools/probes/attend_32x32 && for B in 256 1024 4096; do for f in ./attend32_v12_t*_probe; do timeout 120 $f $B 50; done; done) > $O/attend32_probe.txt 2>&1

[ -x tools/probes/mfma_f32_rate_probe ] && timeout 60 ./tools/probes/mfma_f32_rate_probe > $O/mfma_f32_rate_probe.txt 2>&1
[ -x tools/probes/mfma_valu_coissue_probe ] && timeout 60 ./tools/probes/mfma_valu_coissue_probe > $O/mfma_valu_coissue_probe.txt 2>&1     # does VALU work hide under fp32-input MFMAs? (another wave)
[ -x tools/probes/mfma_valu_samewave_probe ] && timeout 60 ./tools/probes/mfma_valu_samewave_probe > $O/mfma_valu_samewave_probe.txt 2>&1   # ... (the same wave)
for m in eager graph; do echo "== $m"; timeout 200 python tools/probes/train_host/host_vs_device.py $m 2>&1 | grep -v "libdrm\|Warning\|warn"; done > $O/train_host_vs_device.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 bench.py --config train --steps 30 --warmup 10 --no-graph > $O/train_stats.log 2>&1
timeout 200 python tools/gpu_encode_fp32x_ab.py > $O/encode_fp32x_ab.txt 2>&1
timeout 300 python tools/gpu_torch_eager_baseline.py 256 2>&1 | grep -v libdrm > $O/torch_eager_baseline.txt     # the reference's own form of the job (PyTorch-ROCm eager twins) on this GPU
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-extras > $O/stats.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_WAVE32_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/run_sample_once.py 256 bf16 2 > $O/pmc_$n.log 2>&1
done
# the fp32x parity mode (k_sample8x): kernel stats + the same counter groups
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_stats -- python3 tools/run_sample_once.py 256 fp32x 3 > $O/x_stats.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/xpmc_$n -- python3 tools/run_sample_once.py 256 fp32x 2 > $O/xpmc_$n.log 2>&1
done
timeout 200 python tools/gpu_fp32x_perf.py 1 256 768 > $O/fp32x_perf.txt 2>&1
timeout 100 python tools/gpu_phase_profile8.py 256 fp32x > $O/phase8x.txt 2>&1
# decode: kernel stats + counters of the fused kernel at 256 clips
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode_stats -- python3 tools/gpu_decode_perf.py 256 > $O/decode_stats.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/dpmc_$n -- python3 tools/gpu_decode_perf.py 256 > $O/dpmc_$n.log 2>&1
done
timeout 200 python tools/gpu_decode_perf.py > $O/decode_perf.txt 2>&1
# the diffusion_only denoiser (S = 304 per step): kernel stats + counters of the fused step kernel at 256 clips, staged kernels at 64; per-step times
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/den_stats -- python3 tools/gpu_den_once.py 256 bf16 10 > $O/den_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/den_staged_stats -- python3 tools/gpu_den_once.py 256 bf16 10 staged > $O/den_staged_stats.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/npmc_$n -- python3 tools/gpu_den_once.py 256 bf16 4 > $O/npmc_$n.log 2>&1
done
timeout 400 python tools/gpu_den_perf.py 64 256 > $O/den_perf.txt 2>&1
timeout 300 python tools/gpu_dec_perf.py > $O/dec_perf.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/audio_stats -- python3 tools/gpu_audio_perf.py 32 > $O/audio_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/audio_one_stats -- python3 tools/gpu_audio_one_encoder.py 32 > $O/audio_one_stats.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/apmc_$n -- python3 tools/gpu_audio_one_encoder.py 32 > $O/apmc_$n.log 2>&1
done
timeout 200 python tools/gpu_gemm_bench.py > $O/gemm_bench.txt 2>&1
timeout 200 python tools/gpu_audio_perf.py > $O/audio_perf.txt 2>&1
timeout 100 python tools/gpu_phase_profile8.py 256 > $O/phase8.txt 2>&1
timeout 300 python tests/tools/gpu_drift.py > $O/drift.txt 2>&1; cp gpurun_out/drift.json $O/drift.json
timeout 200 python tools/gpu_perf.py > $O/batch_sweep.txt 2>&1
find $O -name "*.csv" | head -40
