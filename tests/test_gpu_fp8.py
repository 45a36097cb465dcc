"""The fp8 variant of the contractions (BASELINE.json configs[4]: "fp8 weights + activations on CDNA4 fp8 MFMA"; the reference
itself is fp32 only, libs/config.py:10-11,75-78) against the fp32 / float64 records the fp32-faithful path is held to.

What the variant is: `net.set_precision("fp8")` - every dense contraction rounds BOTH operands to OCP e4m3 (4 significant bits)
after scaling each tensor by a power of two into e4m3's range, multiplies on v_mfma_f32_32x32x16_fp8_fp8 (forward / input
gradient, csrc/convfp8.hip; the weight gradient feeds the same e4m3 values through the bf16 MFMA, in which they are exact) and
accumulates in fp32; storage, InPlaceNorm statistics, sigma, RootTanh, softmax, gates, losses and Nadam stay fp32.

PARITY OF THE PLUMBING IS EXACT: on operands that are e4m3 numbers already (small integers) every product and every partial sum is
exact, so the kernels must reproduce the fp32 reference BIT FOR BIT - `torch.equal`, no tolerance
(test_fp8_contractions_are_exact_on_e4m3_operands).

STATED TOLERANCES for real-valued operands, in units of e4m3's unit roundoff u = 2^-4 (normalised max error =
max|got - want| / max|want| against the reference's fp32 values; they follow from the format, not from this build's measurements):
    one contraction                                   2 u   = 0.125   (two rounded operands per product: relative error <= 2u + u^2)
    one step, outputs and losses                      8 u   = 0.5     (generator and discriminator in series: ~40 rounded
                                                                      contractions whose errors add like a random walk, 2u sqrt(40) / 1.6)
    one step, per-tensor gradient norms               8 u   = 0.5     (forward AND backward operands rounded)
Measured on MI355X (printed with pytest -s): one contraction 0.03 ... 0.05; outputs 0.17 (64 x 64, batch 64), 0.25 (tiny network),
0.30 (256 x 256, batch 2); gradient norms 0.04 ... 0.1.  The fp32-faithful path's own bounds on the same records are 2e-5 ... 5e-4:
this is a coarse format, the point of the bounds is that the plumbing (scales, panels, accumulation) adds nothing on top of it."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN_DIR, assert_close, load_golden

pytestmark = pytest.mark.gpu
T = torch.as_tensor
DEV = torch.device("cuda:0")
U = 2.0 ** -4

CASES = [
    ("conv", 32, 32, 5, 2, 2, 3, 12, 12), ("conv", 48, 48, 3, 1, 1, 2, 10, 10), ("conv", 100, 200, 1, 1, 0, 5, 6, 6),
    ("convT", 96, 96, 4, 2, 1, 2, 8, 8), ("convT", 192, 192, 4, 2, 1, 1, 6, 4), ("convT", 96, 48, 1, 1, 0, 2, 8, 8),
    ("conv", 272, 256, 3, 1, 1, 8, 16, 16), ("conv", 3, 3, 5, 2, 2, 4, 16, 16), ("convT", 768, 384, 1, 1, 0, 4, 8, 8),
    ("conv", 64, 64, 5, 2, 2, 12, 16, 16),
]


def _run(kind, k, s, p, x, w, g, precision):
    from locate_amd import ops
    rt = ops.Runtime()
    rt.precision = precision
    rt.defer_finalisers = False
    wshape = tuple(w.shape)
    wg, xg = w.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    # u = v = 0: the rank-1 spectral-norm term of dW_bar vanishes, what is left is the contraction's own G / sigma
    u, v = torch.zeros(wshape[0], device=DEV), torch.zeros(w.numel() // wshape[0], device=DEV)
    sigma, wv = torch.tensor([1.0, 1.0], device=DEV), torch.zeros(wshape[0], device=DEV)
    y = ops.SNConvFn.apply(xg, wg, u, v, None, sigma, wv, ops.ConvSpec(kind, k, k, s, p, p), rt)
    y.backward(g.to(DEV))
    return y.detach().cpu(), xg.grad.cpu(), wg.grad.cpu()


def _reference(kind, k, s, p, x, w, g):
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, p) if kind == "conv" else F.conv_transpose2d(xr, wr, None, s, p)
    yr.backward(g)
    return yr.detach(), xr.grad, wr.grad


@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,H,W", CASES)
def test_fp8_contractions_are_exact_on_e4m3_operands(kind, cin, cout, k, s, p, B, H, W):
    """Integers in [-15, 15] are e4m3 numbers after the power-of-two scaling (4 significant bits), their products and sums are exact
    in fp32: forward, input gradient and weight gradient must equal the fp32 reference bit for bit."""
    from locate_amd import ops
    torch.manual_seed(cin + cout + k)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = torch.randint(-15, 16, wshape).float()
    x = torch.randint(-15, 16, (B, cin, H, W)).float()
    y0 = F.conv2d(x, w, None, s, p) if kind == "conv" else F.conv_transpose2d(x, w, None, s, p)
    g = torch.randint(-15, 16, tuple(y0.shape)).float()
    want = _reference(kind, k, s, p, x, w, g)
    before = ops.FP8_CALLS[0]
    got = _run(kind, k, s, p, x, w, g, 3)
    assert ops.FP8_CALLS[0] == before + 2
    for name, a, b in zip(("y", "dx", "dw"), got, want):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,H,W", CASES)
def test_fp8_contraction_vs_fp32_reference(kind, cin, cout, k, s, p, B, H, W):
    torch.manual_seed(cin + cout + k)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = torch.randn(wshape) / (cin * k * k) ** 0.5
    x = torch.randn(B, cin, H, W)
    y0 = F.conv2d(x, w, None, s, p) if kind == "conv" else F.conv_transpose2d(x, w, None, s, p)
    g = torch.randn_like(y0) * 1e-3
    want = _reference(kind, k, s, p, x, w, g)
    got = _run(kind, k, s, p, x, w, g, 3)
    errs = []
    for name, a, b in zip(("y", "dx", "dw"), got, want):
        assert_close(a, b, 2 * U, "fp8 " + name)
        errs.append(float((a - b).abs().max() / b.abs().max()))
    print("fp8 %s %d->%d %dx%d: normalised max errors y %.3f dx %.3f dw %.3f (bound %.3f)" % ((kind, cin, cout, k, k) + tuple(errs) + (2 * U,)))
    # and it IS the coarser arithmetic: bf16 operands are ~16x closer
    y16 = _run(kind, k, s, p, x, w, g, 1)[0]
    e16 = float((y16 - want[0]).abs().max() / want[0].abs().max())
    assert errs[0] > 4 * e16 or cin * k * k < 100


def sub(z, prefix):
    return {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}


def test_fp8_tiny_step_within_stated_tolerance_of_the_fp32_record():
    from locate_amd import Discriminator, Generator, Nadam, NetConfig, TrainStep
    z = load_golden("g8_tiny_e2e")
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    G, D = Generator(cfg), Discriminator(cfg)
    G.load_state_dict(sub(z, "G/sd0/"))
    D.load_state_dict(sub(z, "D/sd0/"))
    G.noise = T(z["G/noise"])
    G, D = G.to(DEV), D.to(DEV)
    G.set_precision("fp8")
    D.set_precision("fp8")
    G.batched_spectral_norm = D.batched_spectral_norm = True
    step = TrainStep(G, D, Nadam(G.parameters(), lr=cfg.glr, betas=(cfg.beta1, cfg.beta2)),
                     Nadam(D.parameters(), lr=cfg.dlr, betas=(cfg.beta1, cfg.beta2)))
    out = step(*(T(z["step1/" + k]).to(DEV) for k in ("latent", "real", "aug")))
    worst = 0.0
    for k in ("generated", "fake", "d_true", "d_gen", "d_error", "g_error"):
        want = z["step1/" + k]
        err = float((out[k].detach().cpu().reshape(want.shape) - T(want)).abs().max() / np.abs(want).max())
        worst = max(worst, err)
        assert err <= 8 * U, ("fp8 step1/" + k, err)
    print("fp8 tiny step: worst normalised output error %.3f (bound %.3f)" % (worst, 8 * U))
    assert all(torch.isfinite(q).all() for q in list(G.parameters()) + list(D.parameters()))


@pytest.mark.parametrize("name,S,B", [("g14_config2_64", 64, 64), ("g20_256_full", 256, 2)])
def test_fp8_full_architectures_vs_reference_record(name, S, B):
    """One step with fp8 operands of the benchmark workload (64 x 64, batch 64) and of configs[4]'s 256 x 256 architecture at full
    width (G 225.9 M / D 186.5 M parameters, ConvTranspose [3072, 3072, 4, 4]; the record's batch 2) against what the reference
    produced in fp32 - and, where the float64 record exists, against float64: losses and per-tensor gradient norms."""
    from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, ops
    z = load_golden(name)
    z64 = load_golden(name + "_f64") if os.path.exists(os.path.join(GOLDEN_DIR, name + "_f64.npz")) else None
    cfg = NetConfig(image_size=S)
    torch.manual_seed(cfg.seed)
    G, GO = get_model(Generator(cfg), cfg.glr, DEV)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, DEV)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    G.set_precision("fp8")
    D.set_precision("fp8")
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
    before = ops.FP8_CALLS[0]
    out = step(latent.to(DEV), real.to(DEV), aug.to(DEV))
    assert ops.FP8_CALLS[0] > before + 50
    worst_out = 0.0
    for k in ("d_true", "d_gen", "d_error", "g_error"):
        want = z[k]
        err = float((out[k].detach().cpu().reshape(want.shape) - T(want)).abs().max() / max(np.abs(want).max(), 1e-6))
        worst_out = max(worst_out, err)
        assert err <= 8 * U, ("fp8 " + k, err)
    assert abs(float(out["fake"].double().norm()) - float(z["fake_norm"])) <= 4 * U * float(z["fake_norm"])
    worst = 0.0
    for tag, got in (("D", rec["d"]), ("G", rec["g"])):
        want = dict(zip(z[tag + "/grad_keys"].tolist(), z[tag + "/grad_norms"]))
        if z64 is not None:
            want = dict(zip(z64["f64/%s/grad_keys" % tag].tolist(), z64["f64/%s/grad_norms" % tag]))
        scale = max(want.values())
        for k, w in want.items():
            if k.endswith(("gamma", "weight_u", "weight_v")):
                continue          # scalars / vectors with heavy cancellation (sum x^2 g, d sigma): not a measure of the format
            err = abs(got[k] - w) / max(w, 1e-2 * scale)
            worst = max(worst, err)
            assert err <= 8 * U, (tag, k, got[k], w)
    print("fp8 %s: worst output error %.3f (bound %.2f), worst gradient-norm deviation %.3f (bound %.2f)" % (name, worst_out, 8 * U, worst, 8 * U))
    assert all(torch.isfinite(q).all() for q in list(G.parameters()) + list(D.parameters()))
