import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_collection_finish(session):
    """A GPU session that holds the two DDPM-1000 parity tests: their shared oracle run (~60 s of CPU, tests/tools/ddpm1000_job.py) starts NOW as a child process
    (CPU only: no GPU visible to it) and runs beside the GPU tests in front of them; tests/test_gpu_parity.py picks its result up, or computes it in line."""
    import subprocess
    import tempfile

    import torch
    if not torch.cuda.is_available() or getattr(pytest, "_amuse_ddpm1000", None) is not None:
        return
    if not any("gpu" in it.keywords and ("test_ddpm1000_fp32_single_clip_vs_oracle" in it.nodeid or "test_full_size_batch_values_against_the_oracle" in it.nodeid)
               for it in session.items):
        return
    out = Path(tempfile.mkdtemp(prefix="amuse_ddpm1000_")) / "ref.npy"
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="4")
    proc = subprocess.Popen([sys.executable, str(REPO / "tests" / "tools" / "ddpm1000_job.py"), str(out)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    pytest._amuse_ddpm1000 = (proc, out)


def pytest_sessionfinish(session, exitstatus):
    pre = getattr(pytest, "_amuse_ddpm1000", None)
    if pre is not None and pre[0].poll() is None:
        pre[0].kill()          # (the exact child this session started)


@pytest.fixture(scope="session")
def host_asan_build(tmp_path_factory):
    """tests/host_asan/build.sh once per session: the library's host code + the stubbed HIP runtime under ASan / UBSan -> directory holding `host_asan`
    (tests/test_host_asan.py) and `plan_sweep` (tests/test_plan_cpu.py)."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (shutil.which(hipcc) or os.path.exists(hipcc)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("host_asan")
    build = subprocess.run(["bash", str(REPO / "tests" / "host_asan" / "build.sh"), str(out)], capture_output=True, text=True, timeout=900)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-2000:]
    return out


def make_reference_tree(root: Path, n_infer_wavs: int = 2):
    """A reference-shaped checkout (configs/, scripts/overrides/, viz_dump/test/{speech,e_speech}) holding the slices of
    configs/base_new.json, diff_latent_v2.json and the override YAMLs that the infer / edit entry points read
    (scripts/main.py:226-268, trainer.py:500-543,1037-1075).  Paths inside the config are absolute, like the author's."""
    import json

    import numpy as np
    from scipy.io import wavfile
    for d in ("configs", "scripts/overrides", "viz_dump/test/speech", "viz_dump/test/e_speech", "data/BEAT-processed",
              "saved-models"):
        (root / d).mkdir(parents=True, exist_ok=True)
    sched = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    ldm = {"scheduler": dict(sched, set_alpha_to_one=True, steps_offset=0, num_inference_timesteps=10, eta=0.0),
           "noisy_scheduler": dict(sched, variance_type="fixed_small", clip_sample=False, prediction_type="epsilon")}
    json.dump(ldm, open(root / "configs/diff_latent_v2.json", "w"), indent=4)
    # diff_o.yaml restores the shipped sampler settings: proves the YAML is merged over the JSON
    (root / "scripts/overrides/diff_o.yaml").write_text(
        "scheduler:\n  set_alpha_to_one: False\n  steps_offset: 1\n  num_inference_timesteps: 50\n")
    use_false = {k: {"use": False, "overwrite": None} for k in ("style_transfer", "emotion_control", "content_control",
                                                               "style_Xemo_transfer")}
    base = {"DATA_PARAM": {"Bvh": {"train_pose_framelen": 300, "fps": 30, "bvh2smplbvh": True}},
            "TRAIN_PARAM": {"tag": "latent_diffusion", "seed": 2024, "pretrained_infer": False, "debug": True,
                            "motion_extractor": {"use": False, "metrics_only": False},
                            "wav_dtw_mfcc": {"ablation": None, "dataset_mean": -9.173025, "dataset_std": 5.062332,
                                             "frame_based_feats": True},
                            "latent_diffusion": {"smplx_data": True, "smplx_rep": "6D", "skip_trans": False,
                                                 "train_upper_body": False, "arch": "diff_latent_v2", "viz_type": "CaMN",
                                                 "shuffle_type": "actors", "pretrained_lpdm": "",
                                                 "pretrained_ast": "wav_dtw_mfcc_test", "pretrained_prior_lpdm_e": "best",
                                                 "pretrained_ldm_lpdm_e": "best"},
                            "test": dict(use_false, replication_times=1, diff_only=False,
                                         audio_list={"use": False, "short_audio_list": False}),
                            "baselines": {"renders": {"task": "custom_renders",
                                                      "custom_audios": str(root / "viz_dump/test/speech"),
                                                      "custom_renders": str(root / "viz_dump/test/gesture")}}}}
    json.dump(base, open(root / "configs/base_new.json", "w"), indent=4)
    common = ("DATA_PARAM:\n  Bvh:\n    bvh2smplbvh: False\nTRAIN_PARAM:\n  pretrained_infer: True\n"
              "  wav_dtw_mfcc:\n    ablation: full\n  latent_diffusion:\n    pretrained_lpdm: LPDM_test\n  test:\n")
    (root / "scripts/overrides/infer_gesture.yaml").write_text(
        common + "    audio_list:\n      use: True\n      short_audio_list: False\n")
    (root / "scripts/overrides/edit_gesture.yaml").write_text(
        common + "    emotion_control_list:\n      use: True\n      actor: miranda\n"
        f"      audios: {root / 'viz_dump/test/e_speech'}\n      renders: {root / 'viz_dump/test/e_gesture'}\n"
        "    audio_list:\n      use: False\n      short_audio_list: False\n")
    rng = np.random.default_rng(0)

    def wav(path, n=159744, rate=16000):
        t = np.arange(n) / rate
        x = 0.3 * np.sin(2 * np.pi * 220 * t * (1 + 0.1 * rng.standard_normal())) + 0.05 * rng.standard_normal(n)
        wavfile.write(path, rate, (x * 20000).astype(np.int16))
    for k in range(n_infer_wavs):
        wav(root / f"viz_dump/test/speech/scott_0_{k}_{k}.wav")
    wav(root / "viz_dump/test/e_speech/9_miranda_source.wav")
    wav(root / "viz_dump/test/e_speech/9_miranda_target.wav")
    return root
