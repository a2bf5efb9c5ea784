"""CPU: `bench.py --gpus N` and `python -m amuse_amd.train_gesture --gpus N` typed directly start their own ranks
(amuse_amd/launch.py) - the parent never needs WORLD_SIZE, the children are real torch.distributed ranks."""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_torchrun_command_shape():
    from amuse_amd import launch
    cmd = launch.torchrun_command("bench.py", ["--gpus", "4"], 4, port=29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-3:] == ["bench.py", "--gpus", "4"]
    assert launch.torchrun_command("amuse_amd.train_gesture", [], 2, module=True)[-2:] == ["-m", "amuse_amd.train_gesture"]
    assert not launch.launched_by_torchrun() or "RANK" in os.environ


def test_bench_gpus_2_starts_its_own_ranks():
    """Without a GPU the CHILDREN refuse ("needs an MI355X") - the parent does not stop at a WORLD_SIZE check."""
    for extra in ([], ["--config", "train"]):
        r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", *extra], cwd=REPO,
                           env=_env(), capture_output=True, text=True, timeout=600)
        out = r.stdout + r.stderr
        assert r.returncode != 0
        assert out.count("needs an MI355X") >= 1, out[-2000:]          # (one line per rank, unless the launcher stops the second rank when the first has failed)
        assert "WORLD_SIZE" not in out.replace("WORLD_SIZE=2 ranks", "")


def test_train_gesture_main_is_data_parallel(tmp_path):
    """Two gloo ranks on the CPU through the module's own launcher: both train (no HIP sampler on the CPU), rank 0 reports and
    writes the checkpoints; weights are identical across ranks after the all-reduced steps (checked inside by the 2-rank test of
    tests/test_train_cpu.py - here: the entry point itself)."""
    r = subprocess.run([sys.executable, "-m", "amuse_amd.train_gesture", "--gpus", "2", "--batch", "2", "--iters-per-epoch", "2",
                        "--device", "cpu", "--out", str(tmp_path)], cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.count("[LPDM-T] Epoch: [1/1]") == 1                # rank 0 only
    assert len(list(tmp_path.glob("prior_model_NoOpt_*_e1.pt"))) == 1 and len(list(tmp_path.glob("latdiff_model_wOpt_*_e1.pt"))) == 1


def test_cli_fn_train_gesture_dispatches_to_the_trainer(tmp_path):
    """`main.py --fn train_gesture` (scripts/main.py:116-155): configuration-driven (batch size / epochs / save frequency from
    base_new.json merged with scripts/overrides/train_gesture.yaml, seed from TRAIN_PARAM), refuses a missing LMDB cache, trains on
    synthetic batches when asked, writes the reference-named checkpoints under <root>/saved-models."""
    import pytest
    from conftest import make_reference_tree
    from amuse_amd import main as cli
    root = make_reference_tree(tmp_path / "tree")
    ov = ("TRAIN_PARAM:\n  latent_diffusion:\n    batch_size: 2\n    n_epochs: 1\n    model_save_freq: 1\n    lr_base: 0.0003\n"
          "    vtex_displacement: {vtex}\n    optimizer_name: {opt}\n  diffusion:\n"
          "    lmdb_cache: BEAT-cache/2023-10-28_30F_fing_smplx_MOSH_identity_v1_feat_based_300\n")
    (root / "scripts/overrides/train_gesture.yaml").write_text(ov.format(vtex="False", opt="adamw"))
    (root / "scripts/overrides/diff_o.yaml").write_text("losses:\n  LAMBDA_KL: 0.002\n  LAMBDA_REC: 1.0\n  LAMBDA_GEN: 1.0\n  LAMBDA_LATENT: 1.0\n"
                                                        "  LAMBDA_JOINT: 1.0\n  LAMBDA_PRIOR: 0.0\n  stage: vae_diffusion\n  train_lpdm:\n    version: v0\n"
                                                        "  use_recons_joints: true\n  predict_epsilon: true\n")
    with pytest.raises(SystemExit, match="LMDB cache"):
        cli.main(["--fn", "train_gesture", "--root", str(root), "--device", "cpu"])
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(["--fn", "train_gesture", "--root", str(root), "--device", "cpu", "--synthetic", "--iters-per-epoch", "2"]) == 0
    # the non-default configuration reaches the trainer: learning rate, the ablation variant of the LMDB id, the loss weights of
    # configs/diff_latent_v2.json merged with diff_o.yaml (advisor finding, round 3)
    assert "lr 0.0003, ablation kind identity" in buf.getvalue() and "LAMBDA_KL 0.002" in buf.getvalue(), buf.getvalue()[-800:]
    assert len(list((root / "saved-models").glob("latdiff_model_wOpt_*_e1.pt"))) == 1
    # what this path cannot do is refused loudly, not silently dropped: the vertex-displacement terms the shipped override asks for ...
    (root / "scripts/overrides/train_gesture.yaml").write_text(ov.format(vtex="True", opt="adamw"))
    with pytest.raises(SystemExit, match="vtex_displacement"):
        cli.main(["--fn", "train_gesture", "--root", str(root), "--device", "cpu", "--synthetic", "--iters-per-epoch", "1"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(["--fn", "train_gesture", "--root", str(root), "--device", "cpu", "--synthetic", "--iters-per-epoch", "1", "--skip-vtex-loss"]) == 0
    assert "WARNING: vtex_displacement" in buf.getvalue()
    # ... and another optimizer
    (root / "scripts/overrides/train_gesture.yaml").write_text(ov.format(vtex="False", opt="sgd"))
    with pytest.raises(SystemExit, match="optimizer_name"):
        cli.main(["--fn", "train_gesture", "--root", str(root), "--device", "cpu", "--synthetic"])
    # the inference entry points still refuse a training configuration (pretrained_infer false), as scripts/main.py:126 does
    (root / "scripts/overrides/infer_gesture.yaml").write_text("TRAIN_PARAM:\n  pretrained_infer: False\n")
    with pytest.raises(AssertionError, match="mismatch"):
        cli.main(["--fn", "infer_gesture", "--root", str(root), "--random-init"])


def test_multigpu_preflight_protocol_on_gloo():
    """tools/multigpu_preflight.py (what to run first when a multi-GPU node appears): its launcher + step protocol with 2 gloo ranks on the CPU -
    init, barrier, the 6,835,661-float all-reduce of train_gesture, the 3 x 256-float all-gather of the all-pairs edit batch; one JSON record, rc 0."""
    r = subprocess.run([sys.executable, str(REPO / "tools" / "multigpu_preflight.py"), "--gpus", "2", "--cpu"], cwd=REPO, env=_env(),
                       capture_output=True, text=True, timeout=600)
    rec = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"tool": "multigpu_preflight"')]
    assert r.returncode == 0 and len(rec) == 1, (r.stdout[-1500:], r.stderr[-1500:])
    assert rec[0]["ok"] and rec[0]["world"] == 2 and [s["step"] for s in rec[0]["steps"]] == ["init", "barrier", "all_reduce", "all_gather"]
    assert all(s["ok"] for s in rec[0]["steps"]) and rec[0]["steps"][2]["floats"] == 6835661
