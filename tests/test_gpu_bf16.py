"""The bf16-operand variant of the contractions (BASELINE.json configs[1] names bf16; the reference itself is fp32 only,
libs/config.py:10-11,75-78) against the SAME fp32 fixtures the fp32-faithful path is held to.

What the variant is: `net.set_precision("bf16")` - every dense contraction (conv / transposed conv / 1x1 / linear, forward,
input gradient and weight gradient) rounds both operands to nearest-even bf16 and issues one v_mfma_f32_32x32x16_bf16 per
slice with fp32 accumulation (six in the fp32-faithful path); storage, InPlaceNorm statistics, sigma of the spectral norm,
RootTanh, softmax, the gates, the losses and Nadam stay fp32.

STATED TOLERANCES (normalised max error = max|got - want| / max|want| against the reference's fp32 values):
    one contraction, K up to 3072            y, dx 8e-3      dw 8e-3
    tiny network, step 1 (g8)                outputs 3e-2    gradients 1.5e-1 (5e-1 for the cancellation-heavy scalars
                                                             gamma / u / v)
    benchmark workload (g14: 64x64, batch 64) losses 3e-2    per-tensor gradient norms 2e-1
A bf16 product carries a relative rounding error of up to 2^-8 per operand; the sums average it down, the ~20 layers of
a pass and the global-statistics norms compound it.  The figures above are ~3x what was measured on MI355X (recorded in
profiles/r02_bf16_tolerances.txt), i.e. regression bounds, not accuracy claims; the fp32-faithful path's own bounds are
2e-5 / 2e-4 on the same fixtures (tests/test_gpu_e2e.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu
T = torch.as_tensor
DEV = torch.device("cuda:0")


def sub(z, prefix):
    return {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}


@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,H,W", [
    ("conv", 32, 32, 5, 2, 2, 3, 12, 12), ("conv", 48, 48, 3, 1, 1, 2, 10, 10), ("conv", 100, 200, 1, 1, 0, 5, 6, 6),
    ("convT", 96, 96, 4, 2, 1, 2, 8, 8), ("convT", 192, 192, 4, 2, 1, 1, 5, 3), ("convT", 96, 48, 1, 1, 0, 2, 8, 8),
    ("conv", 272, 256, 3, 1, 1, 8, 16, 16), ("conv", 3, 3, 5, 2, 2, 4, 16, 16), ("conv", 512, 1, 1, 1, 0, 8, 1, 1),
])
def test_bf16_contraction_vs_fp32_reference(kind, cin, cout, k, s, p, B, H, W):
    from locate_amd import ops
    torch.manual_seed(cin + cout + k)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = torch.randn(wshape) / (cin * k * k) ** 0.5
    x = torch.randn(B, cin, H, W)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p) if kind == "conv" else F.conv_transpose2d(xr, wr, None, s, p)
    g = torch.randn_like(yr)
    yr.backward(g)
    rt = ops.Runtime()
    rt.precision = 1
    wg, xg = w.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    # u = v = 0: the rank-1 spectral-norm term dsigma u v^T of dW_bar vanishes, what is left is the contraction's own G / sigma
    u, v = torch.zeros(wshape[0], device=DEV), torch.zeros(w.numel() // wshape[0], device=DEV)
    sigma, wv = torch.tensor([1.0, 1.0], device=DEV), torch.zeros(wshape[0], device=DEV)
    y = ops.SNConvFn.apply(xg, wg, u, v, None, sigma, wv, ops.ConvSpec(kind, k, k, s, p, p), rt)
    y.backward(g.to(DEV))
    assert_close(y.detach().cpu(), yr.detach(), 8e-3, "bf16 y")
    assert_close(xg.grad.cpu(), xr.grad, 8e-3, "bf16 dx")
    assert_close(wg.grad.cpu(), wr.grad, 8e-3, "bf16 dw")
    # and it IS a different arithmetic: the fp32-faithful path is ~1000x closer
    y32 = ops.SNConvFn.apply(xg.detach(), wg.detach(), u, v, None, sigma, wv, ops.ConvSpec(kind, k, k, s, p, p))
    e32 = float((y32.cpu() - yr.detach()).abs().max() / yr.detach().abs().max())
    e16 = float((y.detach().cpu() - yr.detach()).abs().max() / yr.detach().abs().max())
    assert e32 < 2e-5 and (e16 > 20 * e32 or e16 < 1e-6)


def _tiny(z, precision):
    from locate_amd import Discriminator, Generator, Nadam, NetConfig, TrainStep
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    G, D = Generator(cfg), Discriminator(cfg)
    G.load_state_dict(sub(z, "G/sd0/"))
    D.load_state_dict(sub(z, "D/sd0/"))
    G.noise = T(z["G/noise"])
    G, D = G.to(DEV), D.to(DEV)
    G.set_precision(precision)
    D.set_precision(precision)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    step = TrainStep(G, D, Nadam(G.parameters(), lr=cfg.glr, betas=(cfg.beta1, cfg.beta2)),
                     Nadam(D.parameters(), lr=cfg.dlr, betas=(cfg.beta1, cfg.beta2)))
    return cfg, G, D, step


def test_bf16_tiny_step_within_stated_tolerance_of_the_fp32_record():
    z = load_golden("g8_tiny_e2e")
    cfg, G, D, step = _tiny(z, "bf16")
    rec = {}
    d_orig, g_orig = step.dis_opt.step, step.gen_opt.step

    def d_hook(*a, **k):
        rec["d"] = {kk: q.grad.detach().cpu().clone() for kk, q in D.named_parameters() if q.grad is not None}
        return d_orig(*a, **k)

    def g_hook(*a, **k):
        rec["g"] = {kk: q.grad.detach().cpu().clone() for kk, q in G.named_parameters() if q.grad is not None}
        return g_orig(*a, **k)
    step.dis_opt.step, step.gen_opt.step = d_hook, g_hook
    out = step(*(T(z["step1/" + k]).to(DEV) for k in ("latent", "real", "aug")))
    for k in ("generated", "fake", "d_true", "d_gen", "d_error", "penalty", "g_error"):
        assert_close(out[k].detach().cpu().reshape(z["step1/" + k].shape), z["step1/" + k], 3e-2, "bf16 step1/" + k)
    for net, grads in (("D", rec["d"]), ("G", rec["g"])):
        want = sub(z, "step1/" + net + "/grad/")
        assert set(grads) == set(want)
        for k, v in want.items():
            scalarish = k.endswith(("gamma", "weight_u", "weight_v"))
            assert_close(grads[k], v, 5e-1 if scalarish else 1.5e-1, "bf16 step1 " + net + " grad " + k)
    assert all(torch.isfinite(p).all() for p in list(G.parameters()) + list(D.parameters()))


def test_bf16_benchmark_workload_vs_reference_record():
    """configs[1] as named: 64x64 RGB, batch 64, bf16 operands - one step against the record the reference produced in fp32
    (g14): losses and per-tensor gradient norms."""
    from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model
    z = load_golden("g14_config2_64")
    cfg = NetConfig(image_size=64)
    torch.manual_seed(cfg.seed)
    G, GO = get_model(Generator(cfg), cfg.glr, DEV)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, DEV)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    G.set_precision("bf16")
    D.set_precision("bf16")
    B, S = 64, 64
    latent = torch.randn(B, S)
    real = torch.randn(B, 3, S, S).clamp(-1, 1)
    aug = torch.randn(B, 3, S, S).clamp(-1, 1)
    np.testing.assert_array_equal(latent[0, :4].numpy(), z["after_build_rng_check"])
    step = TrainStep(G, D, GO, DO)
    rec = {}
    d_orig, g_orig = DO.step, GO.step

    def d_hook():
        rec["d"] = {k: float(p.grad.double().norm()) for k, p in D.named_parameters() if p.grad is not None}
        return d_orig()

    def g_hook():
        rec["g"] = {k: float(p.grad.double().norm()) for k, p in G.named_parameters() if p.grad is not None}
        return g_orig()
    DO.step, GO.step = d_hook, g_hook
    out = step(latent.to(DEV), real.to(DEV), aug.to(DEV))
    for k in ("d_true", "d_gen", "d_error", "penalty", "g_error"):
        assert_close(out[k].detach().cpu().reshape(z[k].shape), z[k], 3e-2, "bf16 " + k)
    assert abs(float(out["fake"].double().norm()) - float(z["fake_norm"])) <= 1e-2 * float(z["fake_norm"])
    worst = 0.0
    for tag, got in (("D", rec["d"]), ("G", rec["g"])):
        want = dict(zip(z[tag + "/grad_keys"].tolist(), z[tag + "/grad_norms"]))
        scale = max(want.values())
        for k, w in want.items():
            err = abs(got[k] - w) / max(w, 1e-3 * scale)
            worst = max(worst, err)
            assert err <= 2e-1, (tag, k, got[k], w)
    print("bf16 config 2: worst relative gradient-norm deviation %.3e" % worst)
