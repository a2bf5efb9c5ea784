"""GPU: train_gesture's step with the in-loop sampler on the HIP kernels, and amuse_update_weights behind it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_update_weights_equals_a_fresh_context():
    """amuse_update_weights re-packs in place: afterwards the context is indistinguishable from one created on the new
    weights (sampling, decode, encode; both precisions; the schedule is re-applied)."""
    from amuse_amd import scheduler as sch, weights as wts
    from amuse_amd.engine import HipEngine
    w0 = (wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    w1 = (wts.make_denoiser_weights(1), wts.make_prior_weights(1))
    a, b = HipEngine(*w0), HipEngine(*w1)
    for e in (a, b):
        e.set_schedule(sch.ddim_table())
    gen = torch.Generator().manual_seed(0)
    c, em, s, x = (torch.randn(100, n, generator=gen) for n in (256, 256, 256, 128))
    before = a.sample(c, em, s, "bf16", x_init=x)
    a.update_weights(*w1)
    for prec in ("fp32", "bf16", "fp32x"):
        la, lb = a.sample(c, em, s, prec, x_init=x), b.sample(c, em, s, prec, x_init=x)
        assert torch.equal(la, lb), prec
        da, db = a.vae_decode(la, None, prec, return_feats=True), b.vae_decode(lb, None, prec, return_feats=True)   # fused path in bf16
        assert torch.equal(da["feats"], db["feats"]) and torch.equal(da["poses"], db["poses"]), prec
        assert torch.equal(a.vae_decode(la[:4], [300, 17, 160, 1], prec)["poses"], b.vae_decode(lb[:4], [300, 17, 160, 1], prec)["poses"])
        f = torch.randn(2, 300, 333, generator=gen)
        assert torch.equal(a.vae_encode(f, None, prec)["mu"], b.vae_encode(f, None, prec)["mu"])
    assert not torch.equal(before, a.sample(c, em, s, "bf16", x_init=x))
    # a partial update (bf16 streams only): the bf16 mode is on the new weights (the fp32 mode is then undefined, amuse_hip.h)
    a.update_weights(*w0, what=2)
    assert torch.equal(a.sample(c, em, s, "bf16", x_init=x), before)
    with pytest.raises(Exception):
        a.update_weights(None, None)
    a.close(); b.close()


def test_update_weights_device_equals_the_host_path():
    """amuse_update_weights_device (a gather kernel per packed image, maps learnt from probe runs of the builders) leaves the
    context bitwise where amuse_update_weights / a fresh context on the same values leave it: every mask, both networks, either alone."""
    from amuse_amd import scheduler as sch, weights as wts
    from amuse_amd.engine import HipEngine, flatten_state_dict
    w0 = (wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    w1 = (wts.make_denoiser_weights(1), wts.make_prior_weights(1))
    flat = lambda w: (torch.from_numpy(flatten_state_dict(w[0], wts.denoiser_param_spec())).cuda(),
                      torch.from_numpy(flatten_state_dict(w[1], wts.prior_param_spec())).cuda())
    f0, f1 = flat(w0), flat(w1)
    a, b = HipEngine(*w0), HipEngine(*w1)
    for e in (a, b):
        e.set_schedule(sch.ddim_table())
    gen = torch.Generator().manual_seed(0)
    c, em, s, x = (torch.randn(100, n, generator=gen) for n in (256, 256, 256, 128))
    before = a.sample(c, em, s, "bf16", x_init=x)
    a.update_weights_device(*f1)
    for prec in ("fp32", "bf16", "fp32x"):
        la, lb = a.sample(c, em, s, prec, x_init=x), b.sample(c, em, s, prec, x_init=x)
        assert torch.equal(la, lb), prec
        da, db = a.vae_decode(la, None, prec, return_feats=True), b.vae_decode(lb, None, prec, return_feats=True)
        assert torch.equal(da["feats"], db["feats"]) and torch.equal(da["poses"], db["poses"]), prec
        assert torch.equal(a.vae_decode(la[:4], [300, 17, 160, 1], prec)["poses"], b.vae_decode(lb[:4], [300, 17, 160, 1], prec)["poses"])
        f = torch.randn(2, 300, 333, generator=gen)
        assert torch.equal(a.vae_encode(f, None, prec)["mu"], b.vae_encode(f, None, prec)["mu"])
        assert torch.equal(a.denoise_step(x, 501, c, em, s, prec), b.denoise_step(x, 501, c, em, s, prec))
    # bf16 streams only, the denoiser alone, the prior alone
    a.update_weights_device(*f0, what=2)
    assert torch.equal(a.sample(c, em, s, "bf16", x_init=x), before)
    a.update_weights_device(f1[0], None, what=2)
    assert torch.equal(a.sample(c, em, s, "bf16", x_init=x), b.sample(c, em, s, "bf16", x_init=x))
    lat = b.sample(c, em, s, "bf16", x_init=x)
    assert not torch.equal(a.vae_decode(lat, None, "bf16")["poses"], b.vae_decode(lat, None, "bf16")["poses"])   # (prior still w0)
    a.update_weights_device(None, f1[1], what=2)
    assert torch.equal(a.vae_decode(lat, None, "bf16")["poses"], b.vae_decode(lat, None, "bf16")["poses"])
    # the fp32x mask alone: the split-fp16 streams of the sampler and of the prior's decoder
    a.update_weights_device(*f0, what=8)
    a0 = HipEngine(*w0)
    a0.set_schedule(sch.ddim_table())
    l0 = a0.sample(c, em, s, "fp32x", x_init=x)
    assert torch.equal(a.sample(c, em, s, "fp32x", x_init=x), l0)
    assert torch.equal(a.vae_decode(l0[:8], None, "fp32x")["poses"], a0.vae_decode(l0[:8], None, "fp32x")["poses"])   # k_vae_rows<f16x2>
    assert torch.equal(a.vae_decode(l0, None, "fp32x")["poses"], a0.vae_decode(l0, None, "fp32x")["poses"])           # 100 clips: k_vae_rows8x's stream
    a0.close()
    with pytest.raises(Exception):
        a.update_weights_device(None, None)
    with pytest.raises(ValueError):
        a.update_weights_device(f1[0][:-1], None)
    a.close(); b.close()


def test_train_step_with_hip_inner_sampler():
    from amuse_amd.engine import HipEngine
    from amuse_amd.train_gesture import build_trainer, synthetic_batch
    torch.manual_seed(0)
    tr = build_trainer("cuda:0")
    batch = synthetic_batch(32, 3, "cuda:0")
    l0 = float(tr.train_step(batch))
    ld = {k: float(v) for k, v in tr.lpdm_losses.compute().items()}
    assert np.isfinite(l0) and ld["gen_feature"] > 0 and np.isfinite(ld["gen_feature"])
    assert abs(ld["total"] - (ld["recons_feature"] + 1e-4 * ld["kl_motion"] + ld["inst_loss"] + ld["gen_feature"])) < 1e-4 * max(1.0, ld["total"])
    for _ in range(3):
        tr.train_step(batch)
    assert float(tr.lpdm_losses.compute()["recons_feature"]) < ld["recons_feature"]      # the step optimises
    # the in-loop sampler runs on the CURRENT weights: after its next call (which re-packs on the GPU straight from the
    # trainer's flat parameter buffer) its context equals a fresh one built from the modules' state dicts ...
    s = tr.inner_sampler
    assert s.flat is not None and s.on_device
    s(batch["ld_audio_con"], batch["ld_audio_emo"], batch["ld_audio_sty"], 32)
    fresh = HipEngine(s._den_state(), s._prior_state())
    fresh.set_schedule(s.engine.schedule)
    fresh.set_decode_path("fused")                    # (the in-loop sampler pins the per-clip decode kernel: HipInnerSampler.__init__)
    lat_a = s.engine.sample(batch["ld_audio_con"], batch["ld_audio_emo"], batch["ld_audio_sty"], "bf16", seed=1)
    lat_b = fresh.sample(batch["ld_audio_con"], batch["ld_audio_emo"], batch["ld_audio_sty"], "bf16", seed=1)
    assert torch.equal(lat_a, lat_b) and len(s.sync_ms) >= 4
    assert torch.equal(s.engine.vae_decode(lat_a, None, "bf16")["poses"], fresh.vae_decode(lat_b, None, "bf16")["poses"])
    # ... and so does the host path
    s.engine.update_weights(s._den_state(), s._prior_state(), what=2)
    assert torch.equal(s.engine.sample(batch["ld_audio_con"], batch["ld_audio_emo"], batch["ld_audio_sty"], "bf16", seed=1), lat_b)
    fresh.close()


def test_gpu_gradients_equal_cpu_gradients():
    """The step's forward + backward on the GPU (rocBLAS GEMMs, the fused scaled-dot-product kernels, gradients handed over and
    packed into the bucket) against the same trainer on the CPU: losses and the flat gradient, dropout off, explicit draws."""
    from amuse_amd.train_gesture import build_trainer, synthetic_batch
    B = 4
    g = torch.Generator().manual_seed(11)
    batch = synthetic_batch(B, 7)
    noise, ts = torch.randn(B, 1, 128, generator=g), torch.randint(0, 1000, (B,), generator=g)
    e1, e2 = torch.randn(1, B, 128, generator=g), torch.randn(1, B, 128, generator=g)
    out = {}
    for dev in ("cpu", "cuda:0"):
        tr = build_trainer(dev, use_hip_sampler=False, dropout=0.0)
        b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        loss = tr.forward_losses(b, noise=noise.to(dev), timesteps=ts.to(dev), eps_enc=e1.to(dev), eps_inf=e2.to(dev))
        tr.backward_into_bucket(loss)
        out[dev] = (float(loss.detach()), tr.flat_grad.detach().cpu().clone(), tr)
    (lc, gc, _), (lg, gg, trg) = out["cpu"], out["cuda:0"]
    assert abs(lc - lg) < 2e-5 * max(1.0, abs(lc))
    scale = float(gc.abs().max())
    assert scale > 0 and float((gc - gg).abs().max()) < 2e-4 * scale
    # every p.grad points into the bucket again, and a step moves the flat parameter buffer
    off = 0
    for p in trg.params:
        assert p.grad.data_ptr() == trg.flat_grad.data_ptr() + 4 * off
        off += p.numel()
    before = trg.flat_param.clone()
    trg.lpdm_opt.step()
    assert not torch.equal(before, trg.flat_param)


def test_inner_sampler_on_its_own_stream_changes_nothing(monkeypatch):
    """The no-gradient half (DDIM-50 + decode on the HIP kernels) runs on a side stream beside the forward pass (GestureTrainer.forward_losses); with the
    same seeds the iteration's loss terms - gen_feature, which only the sampler feeds, first of all - equal the in-line order's, step after step."""
    from amuse_amd import train_ops
    from amuse_amd.train_gesture import build_trainer, synthetic_batch
    res = {}
    for side in ("1", "0"):
        torch.manual_seed(3)
        train_ops._OFFSET[0] = 0
        tr = build_trainer("cuda:0", seed=1, sampler_stream=side == "1")
        terms = []
        for i in range(3):
            tr.train_step(synthetic_batch(8, 10 + i, "cuda:0"))
            terms.append({k: float(v) for k, v in tr.lpdm_losses.compute().items()})
        res[side] = terms
        assert (tr._side_stream is not None) == (side == "1")
    for a, b in zip(res["1"], res["0"]):
        assert a["gen_feature"] == b["gen_feature"] and a["gen_feature"] > 0
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-6 * max(1.0, abs(b[k])), k



def test_denoiser_chain_on_its_own_stream_changes_nothing():
    """The Denoiser's chain (the no-gradient encode that feeds it, its forward and backward pass) is issued on a stream of its own beside the prior's
    (GestureTrainer.forward_losses; scratch lane 1 of the library): with the same seeds, the iteration's loss terms AND every parameter after three optimizer
    steps are BITWISE those of the in-line order - same kernels, same operands, another stream."""
    from amuse_amd import train_ops
    from amuse_amd.train_gesture import build_trainer, synthetic_batch
    res = {}
    for den in (True, False):
        torch.manual_seed(3)
        train_ops._OFFSET[0] = 0
        tr = build_trainer("cuda:0", seed=1, denoiser_stream=den, grads_mode="sink")
        terms = []
        for i in range(3):
            tr.train_step(synthetic_batch(32, 10 + i, "cuda:0"))
            terms.append({k: float(v) for k, v in tr.lpdm_losses.compute().items()})
        torch.cuda.synchronize()
        res[den] = (terms, tr.flat_param.detach().clone(), tr.flat_grad.detach().clone())
        assert (tr._den_stream is not None) == den
    for a, b in zip(res[True][0], res[False][0]):
        assert a == b
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
    assert float(res[True][2].abs().max()) > 0


def test_train_mode_inner_sampler_on_the_gpu():
    """The opt-in reference-semantics inner sampler (train_gesture.TrainModeInnerSampler) on the GPU: in eval mode its DDIM-50 latents are the persistent
    HIP sampler's fp32 latents for the SAME clips (both start from the library's counter-based normals: <= 1e-4) and its features the HIP decode's; in
    train mode the layers' counter-based dropout masks are live (another result, fresh on every call), and an iteration trains with it."""
    from amuse_amd.train_gesture import HipInnerSampler, TrainModeInnerSampler, build_trainer, synthetic_batch
    torch.manual_seed(0)
    tr = build_trainer("cuda:0", inner="train")
    s = tr.inner_sampler
    assert isinstance(s, TrainModeInnerSampler) and s.engine is not None
    batch = synthetic_batch(4, 5, "cuda:0")
    con, emo, sty = batch["ld_audio_con"], batch["ld_audio_emo"], batch["ld_audio_sty"]
    hip = HipInnerSampler(tr.model, "cuda:0", precision="fp32")
    for m in tr.model.values():
        m.eval()
    feats, lat = s(con, emo, sty, 4, return_latents=True)              # clips 0..3 of seed 2024
    lat_hip = hip.engine.sample(con, emo, sty, "fp32", seed=2024, clip_index0=0)
    assert float((lat - lat_hip).abs().max()) < 1e-4
    f_hip = hip.engine.vae_decode(lat_hip, None, "fp32", return_feats=True)["feats"]
    assert float((feats - f_hip).abs().max()) < 2e-4
    for m in tr.model.values():
        m.train()
    s.clip_counter = 0
    f1 = s(con, emo, sty, 4)
    s.clip_counter = 0
    f2 = s(con, emo, sty, 4)
    assert float((f1 - feats).abs().max()) > 1e-3 and float((f1 - f2).abs().max()) > 1e-3 and bool(torch.isfinite(f1).all())
    l0 = float(tr.train_step(batch))
    assert np.isfinite(l0) and float(tr.lpdm_losses.compute()["gen_feature"]) > 0
    hip.engine.close()


def test_training_step_as_two_hip_graphs():
    """GestureTrainer.enable_graph: the iteration replayed as two HIP graphs around the all-reduce.  (1) Deterministic parts equal: with dropout off, the draws
    pinned (explicit noise / timesteps / rsample draws are eager calls; so the comparison is on the first replayed step after re-seeding torch's device generator
    the same way for the eager and the graphed trainer) the loss and the parameters after the step agree with the eager trainer to fp32 round-off.  (2) A replay is a NEW
    iteration: successive replays on the same batch give different losses (fresh noise / timesteps / dropout masks), the device-side dropout epoch and AdamW
    step count advance by one per replay, and the optimizer's state_dict reports the replayed steps.  (3) Eager allocations between replays change nothing."""
    from amuse_amd import train_gesture as tg, train_ops
    dev = torch.device("cuda:0")
    batches = [tg.synthetic_batch(8, 40 + i, dev) for i in range(3)]

    def run(graph: bool, dropout: float):
        torch.manual_seed(11)
        train_ops._OFFSET[0] = 0
        tr = tg.build_trainer(dev, seed=4, use_hip_sampler=True, dropout=dropout, grads_mode="sink")
        for i in range(2):
            tr.train_step(batches[i])
        if graph:
            assert tr.enable_graph(batches[2])
        losses = []
        for i in range(4):
            losses.append(float(tr.train_step(batches[i % 3])))
            junk = torch.randn(3_000_000, device=dev).mul_(2)      # an eager allocation between replays
            del junk
        torch.cuda.synchronize()
        sd = tr.lpdm_opt.state_dict()
        steps = {int(float(v["step"])) for v in sd["state"].values()}
        return tr, losses, steps

    tr_g, lg, steps_g = run(True, 0.1)
    assert tr_g._graph is not None and all(np.isfinite(lg)) and len(set(lg)) == len(lg)          # four different iterations
    assert steps_g == {2 + 4}                                                                    # 2 eager + 4 replays (a capture records, it does not run)
    assert int(tr_g.lpdm_opt._t_dev.item()) == 6 and tr_g.lpdm_losses.count == 6
    # the same batch twice in a row: a replay draws fresh noise, timesteps and masks
    a, b = float(tr_g.train_step(batches[0])), float(tr_g.train_step(batches[0]))
    assert a != b
    # other shapes fall back to the eager step and keep the step count consistent
    tr_g.train_step(tg.synthetic_batch(4, 99, dev))
    assert int(float(next(iter(tr_g.lpdm_opt.state_dict()["state"].values()))["step"])) == 9
    # training works: parameters moved and stayed finite
    assert bool(torch.isfinite(tr_g.flat_param).all())
    tr_e, le, steps_e = run(False, 0.1)
    assert steps_e == {6}
    # same number of optimizer steps on the same data from the same initial weights: the parameters of the graphed and the eager trainer agree in distribution
    # (different draws), i.e. the update magnitudes are of the same size
    d_g = float((tr_g.flat_param - tg.build_trainer(dev, seed=4, use_hip_sampler=False).flat_param).abs().mean())
    d_e = float((tr_e.flat_param - tg.build_trainer(dev, seed=4, use_hip_sampler=False).flat_param).abs().mean())
    assert 0.5 < d_g / d_e < 2.0, (d_g, d_e)
