"""End-to-end parity on the MI355X: two full G+D training iterations of the tiny fixture network against the
golden record taken from the reference (tests/golden/g8_tiny_e2e.npz), and one iteration at config 1 width
(32x32, bs 8) against the CPU oracle / the reference's recorded losses and norms (g11_config1.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, assert_close, load_golden

pytestmark = pytest.mark.gpu
T = torch.as_tensor
DEV = torch.device("cuda:0")


def sub(z, prefix):
    return {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}


def assert_step_close(got, want, lr, step, what):
    """Post-Nadam parameters, in units of the learning rate.  The first Nadam steps move every weight by
    ~lr * g / (|g| + eps), i.e. by about +-lr whatever the size of g: an element whose gradient is of the size of
    its own fp32 rounding noise can flip sign and land a full step away (the reference's fp32-vs-fp64 self-noise
    does the same).  So the bound is statistical: mean deviation <= 1e-4 lr, at most 0.1 % of the elements off by
    more than 1 % of a step, and nothing further away than a sign flip allows (x10 on the compounded 2nd step)."""
    if what.endswith("weight_u") or what.endswith("weight_v"):
        # power-iteration state, not an optimizer-driven quantity (G's u/v are never trainable; D's are overwritten
        # by the next forward): plain normalised tolerance
        return assert_close(got, want, 2e-5 if step == 1 else 3e-4, what)
    d = (torch.as_tensor(got).double() - torch.as_tensor(want).double()).abs()
    k = 1 if step == 1 else 10
    assert float(d.mean()) <= 1e-4 * lr * k, "%s: mean |delta| = %.3e lr" % (what, float(d.mean()) / lr)
    assert float((d > 0.01 * lr * k).double().mean()) <= 1e-3, "%s: %.3e of the elements off by > 1%% lr" % (
        what, float((d > 0.01 * lr * k).double().mean()))
    assert float(d.max()) <= 2.5 * lr, "%s: max |delta| = %.3e lr" % (what, float(d.max()) / lr)


def _build_tiny(z, batched_sn=False, concurrent_d=False, stacked_d=False, overlap_wgrad=False):
    from locate_amd import Discriminator, Generator, Nadam, NetConfig, TrainStep
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    G, D = Generator(cfg), Discriminator(cfg)
    G.load_state_dict(sub(z, "G/sd0/"))
    D.load_state_dict(sub(z, "D/sd0/"))
    G.noise = T(z["G/noise"])
    dev = torch.device("cuda:0")
    G, D = G.to(dev), D.to(dev)
    G.batched_spectral_norm = D.batched_spectral_norm = batched_sn
    step = TrainStep(G, D, Nadam(G.parameters(), lr=cfg.glr, betas=(cfg.beta1, cfg.beta2)),
                     Nadam(D.parameters(), lr=cfg.dlr, betas=(cfg.beta1, cfg.beta2)), concurrent_d=concurrent_d,
                     stacked_d=stacked_d, overlap_wgrad=overlap_wgrad)
    return cfg, G, D, step, dev


@pytest.mark.parametrize("batched_sn,concurrent_d,stacked_d,overlap_wgrad", [
    (False, False, False, False), (True, False, False, False), (True, True, False, False), (True, False, True, False),
    (False, False, False, True),      # three separate D passes: weight gradients accumulate across them at the join
    (True, False, True, True)])
def test_tiny_two_steps_golden(batched_sn, concurrent_d, stacked_d, overlap_wgrad):
    """Two full iterations against the reference's record, in every launch mode: per-layer spectral norm, batched
    spectral norm, three-stream D-step, the D-step's three passes stacked into one [3B] pass, and the weight gradients
    on a second stream (ops.weight_grad_stream)."""
    z = load_golden("g8_tiny_e2e")
    cfg, G, D, step, dev = _build_tiny(z, batched_sn, concurrent_d, stacked_d, overlap_wgrad)
    assert step.stacked_d == stacked_d
    for it in (1, 2):
        p = "step%d/" % it
        # capture gradients and pre-step u/v through hooks on the optimizers' step
        rec = {}
        d_step_orig = step.dis_opt.step

        def d_step_hook(*a, **k):
            rec["d_grads"] = {kk: q.grad.detach().cpu().clone() for kk, q in D.named_parameters() if q.grad is not None}
            rec["d_uv"] = {kk: v.detach().cpu().clone() for kk, v in D.state_dict().items()
                           if kk.endswith("weight_u") or kk.endswith("weight_v")}
            out = d_step_orig(*a, **k)
            rec["d_post"] = {kk: v.detach().cpu().clone() for kk, v in D.state_dict().items()}
            return out
        g_step_orig = step.gen_opt.step

        def g_step_hook(*a, **k):
            rec["g_grads"] = {kk: q.grad.detach().cpu().clone() for kk, q in G.named_parameters() if q.grad is not None}
            rec["g_none"] = sorted(kk for kk, q in G.named_parameters() if q.grad is None and q.requires_grad)
            return g_step_orig(*a, **k)
        step.dis_opt.step, step.gen_opt.step = d_step_hook, g_step_hook
        out = step(T(z[p + "latent"]).to(dev), T(z[p + "real"]).to(dev), T(z[p + "aug"]).to(dev))
        step.dis_opt.step, step.gen_opt.step = d_step_orig, g_step_orig
        otol, gtol = (2e-5, 2e-4) if it == 1 else (3e-4, 3e-3)
        for k in ("generated", "fake", "d_true", "d_gen", "d_error", "penalty", "g_error"):
            assert_close(out[k].detach().cpu().reshape(z[p + k].shape), z[p + k], otol, p + k)
        for net, grads in (("D", rec["d_grads"]), ("G", rec["g_grads"])):
            want = sub(z, p + net + "/grad/")
            assert set(grads) == set(want), set(grads) ^ set(want)
            for k, v in want.items():
                assert_close(grads[k], v, gtol, p + net + " grad " + k)
        assert rec["g_none"] == sorted(z[p + "G/none_grads"].tolist())
        for k, v in sub(z, p + "D/sd_pre_step/").items():
            assert_close(rec["d_uv"][k], v, otol, p + k)
        for k, v in sub(z, p + "D/sd_post_step/").items():
            assert_step_close(rec["d_post"][k], v, cfg.dlr, it, p + "D post " + k)
        gsd = G.state_dict()
        for k, v in sub(z, p + "G/sd_post_step/").items():
            assert_step_close(gsd[k].cpu(), v, cfg.glr, it, p + "G post " + k)
        dsd = D.state_dict()
        for k, v in sub(z, p + "D/sd_end/").items():
            assert_close(dsd[k].cpu(), v, 10 * otol, p + "end " + k)
    from locate_amd import parameter_count
    assert parameter_count(D) == int(z["meta/d_param_count_after"])
    G.eval()
    with torch.no_grad():
        img = G(T(z["sample/latent"]).to(dev))
    assert_close(img.cpu(), z["sample/image"], 1e-3)


@pytest.mark.parametrize("name,S,B,ff,stacked,switches", [
    ("g11_config1", 32, 8, 8, False, {}),         # BASELINE configs[0]: 32x32, batch 8, full width
    ("g11_config1", 32, 8, 8, True, {}),
    ("g14_config2_64", 64, 64, 8, True, {}),      # configs[1]: the benchmark workload itself, from the reference
    ("g12_config3_128", 128, 2, 8, True, {}),     # configs[3]: the reference's default 128x128 architecture (C up to 1536)
    ("g13_256_narrow", 256, 2, 1, True, {}),      # configs[4]'s 256x256 architecture at 1/8 width (attention over N = 65 536)
    ("g20_256_full", 256, 2, 8, True, {}),        # configs[4]'s architecture at FULL width: G 225.9 M / D 186.5 M parameters,
                                                  # ConvTranspose [3072, 3072, 4, 4], N = 65 536 softmax rows at C = 48
    # the libs/config.py switches the shipped defaults leave off
    ("g15_depth2_32", 32, 4, 2, True, dict(depth=2)),
    ("g16_depth3_fm2_32", 32, 4, 2, False, dict(depth=3, feature_multiplier=2)),
    ("g16_depth3_fm2_32", 32, 4, 2, True, dict(depth=3, feature_multiplier=2)),
    ("g17_separable_32", 32, 4, 4, False, dict(separable=True)),
    ("g17_separable_32", 32, 4, 4, True, dict(separable=True)),
    ("g18_separable_depth2_fm2_64", 64, 2, 2, True, dict(separable=True, depth=2, feature_multiplier=2)),
    ("g19_separable_128", 128, 2, 8, True, dict(separable=True)),   # configs[3] read as depthwise-factorised blocks
])
def test_full_architectures_step_vs_reference_record(name, S, B, ff, stacked, switches):
    """Seeded construction + one step of the full architectures; losses, output and per-tensor gradient / post-step
    norms against what the reference produced (oracle/gen_golden.py: g11 - g14)."""
    from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model
    z = load_golden(name)
    # the same build and inputs run through the reference in float64 as well (oracle/gen_golden.py f64): where that record
    # exists, gradients and post-step norms are held to the float64 values - the truth - with the bound every tensor has always
    # had, widened at most to three times what the reference's OWN fp32 run deviates from its float64 run
    z64 = load_golden(name + "_f64") if os.path.exists(os.path.join(GOLDEN_DIR, name + "_f64.npz")) else None
    cfg = NetConfig(image_size=S, base_feature_factor=ff, **switches)
    torch.manual_seed(cfg.seed)
    dev = torch.device("cuda:0")
    G, GO = get_model(Generator(cfg), cfg.glr, dev)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
    G.batched_spectral_norm = D.batched_spectral_norm = stacked
    latent = torch.randn(B, S)
    real = torch.randn(B, 3, S, S).clamp(-1, 1)
    aug = torch.randn(B, 3, S, S).clamp(-1, 1)
    np.testing.assert_array_equal(latent[0, :4].numpy(), z["after_build_rng_check"])
    step = TrainStep(G, D, GO, DO)
    assert step.stacked_d == stacked
    rec = {}
    d_orig, g_orig = DO.step, GO.step
    # the gates' d(gamma) = sum x^2 g: their sum of MAGNITUDES is recorded beside them, so that the comparison below can
    # be made relative to what the sum is conditioned on instead of to its (cancelled) value
    from locate_amd import ops
    magnitude = {}
    gate_backward = ops.GateFn.backward

    def observed_gate_backward(ctx, g):
        x, _, gamma = ctx.saved_tensors
        magnitude[gamma.data_ptr()] = magnitude.get(gamma.data_ptr(), 0.0) + float((x.double() ** 2 * g.double()).abs().sum())
        return gate_backward(ctx, g)

    def d_hook():
        rec["d"] = {k: float(p.grad.double().norm()) for k, p in D.named_parameters() if p.grad is not None}
        rec["d_mag"] = {k: magnitude.get(p.data_ptr(), 0.0) for k, p in D.named_parameters() if k.endswith("gamma")}
        magnitude.clear()
        return d_orig()

    def g_hook():
        rec["g"] = {k: float(p.grad.double().norm()) for k, p in G.named_parameters() if p.grad is not None}
        rec["g_mag"] = {k: magnitude.get(p.data_ptr(), 0.0) for k, p in G.named_parameters() if k.endswith("gamma")}
        magnitude.clear()
        return g_orig()
    DO.step, GO.step = d_hook, g_hook
    ops.GateFn.backward = staticmethod(observed_gate_backward)
    try:
        out = step(latent.to(dev), real.to(dev), aug.to(dev))
    finally:
        ops.GateFn.backward = staticmethod(gate_backward)
    for k in ("d_true", "d_gen", "d_error", "penalty", "g_error"):
        assert_close(out[k].detach().cpu().reshape(z[k].shape), z[k], 5e-5, k)
    assert_close(out["fake"].flatten()[:16].cpu(), z["fake_first"], 5e-5)
    assert abs(float(out["fake"].double().norm()) - float(z["fake_norm"])) <= 1e-4 * float(z["fake_norm"])
    for tag, got in (("D", rec["d"]), ("G", rec["g"])):
        keys = z[tag + "/grad_keys"].tolist()
        assert sorted(keys) == sorted(got)
        want = dict(zip(keys, z[tag + "/grad_norms"]))
        scale = max(want.values())
        if z64 is not None:
            ref32 = dict(zip(z64["f32/%s/grad_keys" % tag].tolist(), z64["f32/%s/grad_norms" % tag]))
            ref64 = dict(zip(z64["f64/%s/grad_keys" % tag].tolist(), z64["f64/%s/grad_norms" % tag]))
            assert sorted(ref64) == sorted(keys)
            for k in keys:
                # gates' d(gamma) = sum x^2 g included, at the same 5e-4 as every other tensor: measured against float64 the
                # kernels' sums are 3-10x closer than the reference's own fp32 run (tools/post_step_probe.py: 1.8e-5 ... 1.5e-4
                # of the value at 128 x 128, the reference's fp32 run 2e-4 ... 1.4e-3)
                allowed = max(5e-4 * max(ref64[k], 1e-3 * scale), 3.0 * abs(ref32[k] - ref64[k]))
                assert abs(got[k] - ref64[k]) <= allowed, (tag, k, got[k], ref64[k], ref32[k], allowed)
            continue
        for k in keys:
            # d(gamma) = sum x^2 g over a whole activation is a scalar with heavy cancellation: at batch 2 even the CPU
            # oracle - the same ATen kernels as the reference, merely composed differently - deviates by 6.5e-5 on the
            # generator's gammas (1e-6 elsewhere); tools/full_arch_errors.py lists the per-tensor deviations.  The kernel
            # computing the sum is held to 2e-5 on full-size random operands (test_gpu_ops.py::
            # test_residual_gate_large_vs_oracle) and to 3e-4 element by element on the tiny network's recorded gradients
            # (test_tiny_two_steps_golden); what is left here is upstream fp32 rounding amplified by the cancellation, and
            # any re-association upstream moves it (128x128 architecture, block_1, C = 768 at 8x8, batch 2: 5e-4 ... 2e-3
            # of the VALUE across kernel variants that all pass every other check).  So a gamma gradient must be within
            # 1e-3 of its value OR within 2e-5 of the sum of the magnitudes of its terms - the bound every other tensor of
            # this test is held to, applied to what the sum is conditioned on.
            tol = 1e-3 if k.endswith("gamma") else 5e-4
            allowed = tol * max(want[k], 1e-3 * scale)
            if k.endswith("gamma"):
                allowed = max(allowed, 2e-5 * rec[tag.lower() + "_mag"].get(k, 0.0))
            assert abs(got[k] - want[k]) <= allowed, (tag, k, got[k], want[k], allowed)
    for tag, net in (("D", D), ("G", G)):
        sd = net.state_dict()
        post32 = dict(zip(z[tag + "/post_keys"].tolist(), z[tag + "/post_norms"]))
        post64 = post32 if z64 is None else dict(zip(z64["f64/%s/post_keys" % tag].tolist(), z64["f64/%s/post_norms" % tag]))
        for k, w32 in post32.items():
            w = post64[k]
            if tag == "D" and (k.endswith("weight_u") or k.endswith("weight_v")):
                continue   # D's u/v were advanced once more by the G-step's D forward after the record point
            # absolute term: 1 % of one learning-rate step.  The first Nadam step moves an element by lr * g / (|g| + 1e-8):
            # a full +-lr unless |g| is within a few 1e-8 of zero, where the step - and with it the norm of a freshly
            # initialised all-zero bias (128 elements, norm = 11.3 lr) - carries the gradient's rounding noise
            lr = cfg.dlr if tag == "D" else cfg.glr
            allowed = max(1e-4 * max(w, 1e-6) + 0.01 * lr, 3.0 * abs(w32 - w))
            assert abs(float(sd[k].double().norm()) - w) <= allowed, (tag, k, float(sd[k].double().norm()), w, w32)


@pytest.mark.parametrize("S,B,ff,stacked,switches", [
    (16, 1, 1, False, {}),                          # smallest image the architecture admits (3 blocks, no attention in G), batch 1
    (16, 3, 2, True, {}),
    (32, 5, 2, True, {}),                           # odd batch
    (32, 1, 1, True, dict(separable=True)),         # batch 1 through the stacked D-step with the grouped kernels
    (64, 3, 1, True, dict(depth=2)),
    (32, 7, 1, False, dict(separable=True, feature_multiplier=3)),
    (64, 64, 8, True, {}),                          # BASELINE configs[1], the benchmark workload itself at full width: EVERY
                                                    # gradient element by element (the reference record g14 holds norms only)
])
def test_step_vs_oracle_on_unusual_shapes(S, B, ff, stacked, switches):
    """One full G+D step against the CPU oracle, element by element (outputs, every gradient, post-step weights), on
    shapes none of the reference records covers: the smallest image size, batch 1, odd batches, mixed switches - and on the
    benchmark workload itself (64 x 64, batch 64, full width)."""
    from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, init
    from oracle import locate_oracle as O
    cfg = NetConfig(image_size=S, base_feature_factor=ff, **switches)
    torch.manual_seed(1000 + S + B)
    G, D = Generator(cfg), Discriminator(cfg)
    G.apply(init)
    D.apply(init)
    latent = torch.randn(B, S)
    real = torch.randn(B, 3, S, S).clamp(-1, 1)
    aug = torch.randn(B, 3, S, S).clamp(-1, 1)
    ocfg = O.NetConfig(image_size=S, base_feature_factor=ff, **switches)
    PG = O.make_params({k: v.clone() for k, v in G.state_dict().items()})
    PD = O.make_params({k: v.clone() for k, v in D.state_dict().items()})
    want = O.train_step(PG, PD, G.noise.clone(), O.Nadam(ocfg.glr, (ocfg.beta1, ocfg.beta2)),
                        O.Nadam(ocfg.dlr, (ocfg.beta1, ocfg.beta2)), latent, real, aug, ocfg)
    # the same step in float64: the yardstick for the one ill-conditioned output, the consistency penalty
    PG64 = O.make_params({k: v.clone().double() for k, v in G.state_dict().items()})
    PD64 = O.make_params({k: v.clone().double() for k, v in D.state_dict().items()})
    want64 = O.train_step(PG64, PD64, G.noise.clone().double(), O.Nadam(ocfg.glr, (ocfg.beta1, ocfg.beta2)),
                          O.Nadam(ocfg.dlr, (ocfg.beta1, ocfg.beta2)), latent.double(), real.double(), aug.double(), ocfg)
    from locate_amd import Nadam
    dev = torch.device("cuda:0")
    G, D = G.to(dev), D.to(dev)
    G.batched_spectral_norm = D.batched_spectral_norm = stacked
    GO = Nadam(G.parameters(), lr=cfg.glr, betas=(cfg.beta1, cfg.beta2))
    DO = Nadam(D.parameters(), lr=cfg.dlr, betas=(cfg.beta1, cfg.beta2))
    step = TrainStep(G, D, GO, DO)
    rec = {}
    d_orig, g_orig = DO.step, GO.step

    def d_hook():
        rec["d"] = {k: p.grad.detach().cpu().clone() for k, p in D.named_parameters() if p.grad is not None}
        out = d_orig()
        rec["d_post"] = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
        return out

    def g_hook():
        rec["g"] = {k: p.grad.detach().cpu().clone() for k, p in G.named_parameters() if p.grad is not None}
        return g_orig()
    DO.step, GO.step = d_hook, g_hook
    out = step(latent.to(dev), real.to(dev), aug.to(dev))
    for k in ("generated", "fake", "d_true", "d_gen", "d_error", "g_error"):
        assert_close(out[k].detach().cpu().reshape(want[k].shape), want[k], 3e-5, k)
    # the consistency penalty 100 (mean D(real) - mean D(aug))^2 (grad_penalty.py:1-2) squares the DIFFERENCE delta of two
    # discriminator outputs, and delta is tiny (1e-4 ... 1e-3 of the outputs): deviations of the outputs that pass the 3e-5
    # check above with room to spare do not cancel in it - the fp32 oracle itself deviates from its float64 run by up to 4.5e-5
    # of the penalty.  It is held to the FLOAT64 value with: 3e-5 like everything else, or three times the fp32 oracle's own
    # deviation, or what the outputs' own 3e-5 allowance explains to first order
    # (|d penalty| <= 200 |delta| (|d mean_real| + |d mean_aug|) <= 400 |delta| max|d - d64|) with the deviation of the outputs
    # taken at the FIXED 3e-5 allowance they were just held to (3e-5 max|d_true64|) - never at the deviation this build happens
    # to show: the bound does not grow with the error under test.
    p64, p32, got = float(want64["penalty"]), float(want["penalty"]), float(out["penalty"])
    allowance = 3e-5 * float(want64["d_true"].abs().max())
    explained = 400.0 * (p64 / 100.0) ** 0.5 * allowance
    assert abs(got - p64) <= max(3e-5 * abs(p64), 3.0 * abs(p32 - p64), explained), ("penalty", got, p32, p64, allowance, explained)
    for tag, got, ref in (("D", rec["d"], want["d_grads"]), ("G", rec["g"], want["g_grads"])):
        assert sorted(got) == sorted(ref), tag
        for k, v in ref.items():
            # d(gamma) = sum x^2 g: a scalar with heavy cancellation (see test_full_architectures_step_vs_reference_record)
            assert_close(got[k], v, 3e-4, tag + " grad " + k)
    for k, v in want["d_post_step"].items():
        assert_step_close(rec["d_post"][k], v, cfg.dlr, 1, "D post " + k)
    gsd = G.state_dict()
    for k, v in want["g_post_step"].items():
        assert_step_close(gsd[k].cpu(), v, cfg.glr, 1, "G post " + k)


@pytest.mark.parametrize("mode", ["stacked", "stacked+wgrad-stream", "three-streams", "stacked+early-second-pass"])
def test_graph_replay_equals_eager(mode, monkeypatch):
    """hipGraph replay of the captured phases must produce the same trajectory as eager launches - BIT FOR BIT: the kernels are
    deterministic (no atomics on values; split-K partial sums are added in a fixed order) and a replay launches exactly the
    kernels the eager iteration does.  Modes: the stacked D-step (what bench.py runs), the same with the weight gradients on
    a second stream, and the three-stream D-step, whose three discriminator passes are parallel branches of ONE graph and must
    not share split-K arrival counters (ops._counters); and the opt-in schedule LOCATE_G2_EARLY=1 - the generator's two power
    iterations hoisted into a graph of their own, both generator passes replayed concurrently, the second on its own counter
    lane (graph.py)."""
    from locate_amd.graph import GraphedTrainStep
    z = load_golden("g8_tiny_e2e")
    kw = {"stacked": dict(stacked_d=True), "stacked+wgrad-stream": dict(stacked_d=True, overlap_wgrad=True),
          "three-streams": dict(concurrent_d=True), "stacked+early-second-pass": dict(stacked_d=True)}[mode]
    if mode == "stacked+early-second-pass":
        monkeypatch.setenv("LOCATE_G2_EARLY", "1")
    cfg, G1, D1, step1, dev = _build_tiny(z, True, **kw)    # same arithmetic, launched eagerly
    _, G2, D2, step2, _ = _build_tiny(z, True, **kw)
    lat, real, aug = (T(z["step1/" + k]).to(dev) for k in ("latent", "real", "aug"))
    runner = GraphedTrainStep(step2, lat, real, aug, warmup=2)      # 2 eager iterations, then capture (no execution)
    assert (runner.begin_graph is not None) == (mode == "stacked+early-second-pass")
    for _ in range(2):
        out1 = step1(lat, real, aug)
    for _ in range(3):                                               # three replays
        out1 = step1(lat, real, aug)
        out2 = runner.replay()
    torch.cuda.synchronize()
    for k in ("d_error", "g_error", "fake"):
        assert torch.equal(out2[k], out1[k]), k
    for (k, a), (_, b) in zip(D1.state_dict().items(), D2.state_dict().items()):
        assert torch.equal(a, b), "D " + k
    for (k, a), (_, b) in zip(G1.state_dict().items(), G2.state_dict().items()):
        assert torch.equal(a, b), "G " + k


def test_benchmark_mode_replay_equals_eager_at_full_width():
    """The launch mode bench.py times - hipGraph replay of the stacked D-step, the G-step's generator pass on a second stream,
    weight gradients in line, in-launch split-K combines, tall tiles, the fp16-piece and window forms - at the benchmark's own
    size (BASELINE configs[1]: 64 x 64, batch 64, full width, built from the reference record g14's seed and inputs): two eager
    iterations + two replays against four eager iterations, BIT FOR BIT (losses, generated batch, every parameter of both
    networks), and the first iteration's losses against what the reference itself produced (g14_config2_64, main.py:142-172)."""
    from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, ops
    from locate_amd.graph import GraphedTrainStep
    z = load_golden("g14_config2_64")
    S, B = 64, 64
    cfg = NetConfig(image_size=S)
    dev = torch.device("cuda:0")

    def build():
        torch.manual_seed(cfg.seed)
        G, GO = get_model(Generator(cfg), cfg.glr, dev)
        D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
        G.batched_spectral_norm = D.batched_spectral_norm = True
        latent = torch.randn(B, S)
        real = torch.randn(B, 3, S, S).clamp(-1, 1)
        aug = torch.randn(B, 3, S, S).clamp(-1, 1)
        np.testing.assert_array_equal(latent[0, :4].numpy(), z["after_build_rng_check"])
        return G, D, TrainStep(G, D, GO, DO, overlap_wgrad=False), latent.to(dev), real.to(dev), aug.to(dev)

    G1, D1, step1, lat, real, aug = build()
    G2, D2, step2, _, _, _ = build()
    f16_before, win_before = dict(ops.F16_CALLS), ops.WIN_CALLS[0]
    first = step1(lat, real, aug)
    for k in ("d_true", "d_gen", "d_error", "penalty", "g_error"):
        assert_close(first[k].detach().cpu().reshape(z[k].shape), z[k], 5e-5, k)
    assert all(ops.F16_CALLS[k] > f16_before[k] for k in ("fwd", "dgrad", "wgrad")), "the fp16-piece form is live at this size"
    assert ops.WIN_CALLS[0] > win_before, "the window form is live at this size"
    for _ in range(3):
        out1 = step1(lat, real, aug)
    runner = GraphedTrainStep(step2, lat, real, aug, warmup=2)          # bench.py's default: two eager iterations, capture, replay
    for _ in range(2):
        out2 = runner.replay()
    torch.cuda.synchronize()
    for k in ("d_error", "g_error", "penalty", "fake"):
        assert torch.equal(out2[k], out1[k]), k
    for (k, a), (_, b) in zip(D1.state_dict().items(), D2.state_dict().items()):
        assert torch.equal(a, b), "D " + k
    for (k, a), (_, b) in zip(G1.state_dict().items(), G2.state_dict().items()):
        assert torch.equal(a, b), "G " + k


def test_checkpoint_reference_file_reproduces_the_generated_batch(tmp_path):
    """A folder as the reference writes it (torch.save(state_dict) -> netG.torch / netD.torch, main.py:235-236) loads into
    the mirror; with the noise map (which the reference forgets to save, models.py:59) the first generated batch of the
    g8 record comes out."""
    from locate_amd import Discriminator, Generator, NetConfig, load_checkpoint, save_checkpoint
    z = load_golden("g8_tiny_e2e")
    for tag, name in (("G", "netG.torch"), ("D", "netD.torch")):
        torch.save({k[len(tag + "/sd0/"):]: T(z[k]) for k in z.files if k.startswith(tag + "/sd0/")}, tmp_path / name)
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    torch.manual_seed(5)
    G, D = Generator(cfg).to(DEV), Discriminator(cfg).to(DEV)
    rec = load_checkpoint(str(tmp_path), G, D)
    assert not rec["noise_restored"]
    with torch.no_grad():
        G.noise.copy_(T(z["G/noise"]))
    with torch.no_grad():
        img = G(T(z["step1/latent"]).to(DEV))
        d_true = D(T(z["step1/real"]).to(DEV))
    assert_close(img.cpu(), z["step1/generated"], 2e-5, "generated from a reference-format checkpoint")
    assert_close(d_true.cpu().view(-1), z["step1/d_true"], 2e-5, "d_true from a reference-format checkpoint")
    # and our own folder carries the noise map
    out = tmp_path / "own"
    save_checkpoint(str(out), G, D)
    torch.manual_seed(6)
    G2, D2 = Generator(cfg).to(DEV), Discriminator(cfg).to(DEV)
    assert load_checkpoint(str(out), G2, D2)["noise_restored"]
    assert torch.equal(G2.noise, G.noise) and G2.noise.is_cuda
    # G's u/v advanced by one forward before the save: both models continue from that state, with the same (deterministic)
    # kernels - bit for bit the same image
    with torch.no_grad():
        img2, img1 = G2(T(z["step1/latent"]).to(DEV)), G(T(z["step1/latent"]).to(DEV))
    assert torch.equal(img1, img2)


def test_resume_from_checkpoint_continues_the_reference_trajectory(tmp_path):
    """Step 1, save (weights, noise, Nadam moments and float64 schedules, the u/v-trainable flag), load into freshly
    built networks and optimizers, step 2: the same values as the uninterrupted run recorded from the reference."""
    from locate_amd import load_checkpoint, save_checkpoint
    z = load_golden("g8_tiny_e2e")
    cfg, G, D, step, dev = _build_tiny(z, True, False, True, False)
    step(*(T(z["step1/" + k]).to(dev) for k in ("latent", "real", "aug")))
    save_checkpoint(str(tmp_path), G, D, step.gen_opt, step.dis_opt)
    cfg2, G2, D2, step2, _ = _build_tiny(z, True, False, True, False)
    with torch.no_grad():
        G2.noise.zero_()
    rec = load_checkpoint(str(tmp_path), G2, D2, step2.gen_opt, step2.dis_opt)
    assert rec == {"noise_restored": True, "gen_opt_restored": True, "dis_opt_restored": True, "dis_uv_trainable": True}
    for opt in (step2.gen_opt, step2.dis_opt):
        assert all(st["sched"].dtype == torch.float64 and st["sched"].is_cuda for st in opt.state.values())
    out = step2(*(T(z["step2/" + k]).to(dev) for k in ("latent", "real", "aug")))
    for k in ("generated", "fake", "d_true", "d_gen", "d_error", "penalty", "g_error"):
        assert_close(out[k].detach().cpu().reshape(z["step2/" + k].shape), z["step2/" + k], 3e-4, "resumed step2/" + k)
    dsd = D2.state_dict()
    for k, v in sub(z, "step2/D/sd_end/").items():
        assert_close(dsd[k].cpu(), v, 3e-3, "resumed end " + k)
