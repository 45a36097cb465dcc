"""Parity of the gfx950 kernels (through the C ABI and the autograd layer) against the golden fixtures
generated from the reference and against the CPU oracle on seeded inputs.  Needs an MI355X: run with -m gpu.
Tolerances: normalised max error (max |got - want| / max |want|); fp32 per-op 1e-5 ... 1e-4 as stated per test
(the reference's own fp32-vs-fp64 noise is <= 7e-7 on outputs, <= 1.3e-5 on gradients);
bit-exact for indexing ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, load_golden, rel_err

pytestmark = pytest.mark.gpu

T = torch.as_tensor


def dev():
    return torch.device("cuda:0")


def G_(a):
    return T(a).to(dev())


def sub(z, prefix):
    return {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}


def test_device_is_gfx950_and_library_loaded():
    from locate_amd._lib import LIB_PATH, require_gpu
    arch, cus, wave = require_gpu()
    assert arch.startswith("gfx950") and wave == 64 and cus >= 200, (arch, cus, wave)
    maps = open("/proc/self/maps").read()
    assert LIB_PATH in maps


# ------------------------------------------------------------------------------------------------ G1
def test_roottanh_golden_and_large():
    from locate_amd import ops
    from oracle import locate_oracle as O
    z = load_golden("g1_roottanh_f32")
    x = G_(z["x"]).requires_grad_(True)
    y = ops.root_tanh(x)
    y.backward(G_(z["g"]))
    assert torch.isfinite(x.grad).all()
    assert_close(y.cpu(), z["y"], 2e-6, "y")
    assert_close(x.grad.cpu(), z["dx"], 2e-6, "dx")
    z64 = load_golden("g1_roottanh_f64")     # fp64 truth: the fp32 kernel must be as close as the reference's fp32
    x = G_(z64["x"].astype(np.float32)).requires_grad_(True)
    y = ops.root_tanh(x)
    y.backward(G_(z64["g"].astype(np.float32)))
    assert_close(y.cpu(), z64["y"], 2e-6)
    assert_close(x.grad.cpu(), z64["dx"], 2e-6)
    # odd length (tail path) and a full-size activation
    for n in (1, 7, 1023, 64 * 96 * 64 * 64):
        torch.manual_seed(n)
        xc = torch.randn(n) * 4
        gc = torch.randn(n)
        xr = xc.clone().requires_grad_(True)
        O.root_tanh(xr).backward(gc)
        xg = xc.to(dev()).requires_grad_(True)
        yg = ops.root_tanh(xg)
        yg.backward(gc.to(dev()))
        assert_close(yg.cpu(), O.root_tanh(xc), 2e-6, "n=%d" % n)
        assert_close(xg.grad.cpu(), xr.grad, 3e-6, "n=%d grad" % n)


def test_tanh():
    from locate_amd import ops
    torch.manual_seed(3)
    xc = torch.randn(5000) * 3
    xg = xc.to(dev()).requires_grad_(True)
    y = ops.tanh(xg)
    g = torch.randn(5000)
    y.backward(g.to(dev()))
    assert_close(y.cpu(), torch.tanh(xc), 2e-6)
    assert_close(xg.grad.cpu(), g * (1 - torch.tanh(xc) ** 2), 5e-6)


# ------------------------------------------------------------------------------------------------ G2
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_inplace_norm_golden(case):
    from locate_amd import ops
    z = load_golden("g2_inplace_norm")
    for mode, yname in (("w", "weight"), ("s", "scale")):
        p = "%s_%s_" % (case, mode)
        x = G_(z[p + "x"]).requires_grad_(True)
        y = G_(z[p + yname]).requires_grad_(True)
        b = G_(z[p + "bias"]).requires_grad_(True)
        out = ops.inplace_norm(x, y, b)
        out.backward(G_(z[p + "g"]))
        assert_close(out.cpu(), z[p + "out"], 1e-5, p + "out")
        assert_close(x.grad.cpu(), z[p + "dx"], 3e-5, p + "dx")
        assert_close(y.grad.cpu(), z[p + "d" + yname], 3e-5, p + "dy")
        assert_close(b.grad.cpu(), z[p + "dbias"], 1e-5, p + "db")


def test_inplace_norm_fused_activation_and_big():
    from locate_amd import ops
    from oracle import locate_oracle as O
    torch.manual_seed(11)
    # (the 2x2 ... 8x8 planes take the several-planes-per-wave backward, norm_bwd_plane_small_kernel: one to 32 lanes per plane,
    # plane counts that do not fill the last wave)
    for shape, per_sample in (((4, 6, 5, 7), False), ((8, 48, 64, 64), True), ((64, 768, 2, 2), True), ((3, 5, 1, 1), False),
                              ((5, 7, 2, 4), False), ((6, 33, 4, 4), True), ((3, 11, 6, 6), False), ((7, 13, 8, 8), True),
                              ((2, 9, 8, 16), False), ((96, 64, 1, 1), True), ((7, 33, 1, 1), False)):
        B, C = shape[:2]
        x = torch.randn(shape) * 1.7 + 0.4
        y = torch.randn(B if per_sample else 1, C, 1, 1)
        b = torch.randn(1, C, 1, 1)
        g = torch.randn(shape)
        xr, yr, br = (t.clone().requires_grad_(True) for t in (x, y, b))
        ref = O.root_tanh(O.inplace_norm(xr, yr, br))
        ref.backward(g)
        xg, yg, bg = (t.to(dev()).requires_grad_(True) for t in (x, y, b))
        out = ops.inplace_norm(xg, yg, bg, True)
        out.backward(g.to(dev()))
        assert_close(out.cpu(), ref, 1e-5, "act %s" % (shape,))
        assert_close(xg.grad.cpu(), xr.grad, 5e-5, "dx %s" % (shape,))
        assert_close(yg.grad.cpu(), yr.grad, 5e-5, "dy %s" % (shape,))
        assert_close(bg.grad.cpu(), br.grad, 5e-5, "db %s" % (shape,))


# ------------------------------------------------------------------------------------------------ G3
def test_residual_gate_golden():
    from locate_amd import ops
    z = load_golden("g3_residual")
    for p in ("full_", "bc_"):
        x = G_(z[p + "x"]).requires_grad_(True)
        a = G_(z[p + "a"]).requires_grad_(True)
        gamma = G_(z[p + "gamma"]).requires_grad_(True)
        out = ops.residual_gate(x, a, gamma)
        out.backward(G_(z[p + "g"]))
        assert_close(out.cpu(), z[p + "out"], 1e-6)
        assert_close(x.grad.cpu(), z[p + "dx"], 1e-6)
        assert_close(a.grad.cpu(), z[p + "da"], 1e-5)
        assert_close(gamma.grad.cpu(), z[p + "dgamma"], 1e-5)     # x^2 g, as the reference codes it


def test_residual_gate_large_vs_oracle():
    from locate_amd import ops
    from oracle import locate_oracle as O
    torch.manual_seed(5)
    # (planes of 4 ... 128 elements: gate_bwd_small_kernel, one to 32 lanes per plane)
    for shape, bc in (((64, 48, 64, 64), False), ((64, 192, 16, 16), True), ((8, 512, 1, 1), False), ((48, 512, 2, 2), True),
                      ((5, 7, 2, 2), False), ((6, 33, 4, 4), True), ((3, 11, 6, 6), False), ((7, 13, 8, 8), True), ((2, 9, 8, 16), False),
                      ((9, 5, 2, 4), True), ((64, 512, 1, 1), True), ((3, 64, 1, 1), True)):
        x = torch.randn(shape)
        a = torch.randn(shape[0], shape[1], 1, 1) if bc else torch.randn(shape)
        gamma = torch.tensor([[3.0]])
        g = torch.randn(shape)
        xr, ar, gr = (t.clone().requires_grad_(True) for t in (x, a, gamma))
        ref = O.residual_gate(xr, ar.expand_as(xr), gr)
        ref.backward(g)
        xg, ag, gg = (t.to(dev()).requires_grad_(True) for t in (x, a, gamma))
        out = ops.residual_gate(xg, ag, gg)
        out.backward(g.to(dev()))
        assert_close(out.cpu(), ref, 1e-6)
        assert_close(xg.grad.cpu(), xr.grad, 1e-6)
        assert_close(ag.grad.cpu(), ar.grad, 2e-5)
        assert_close(gg.grad.cpu(), gr.grad, 2e-5)


# ------------------------------------------------------------------------------------------------ G4
def _sn_inner(name):
    nn = torch.nn
    return {
        "conv5s2": lambda: nn.Conv2d(4, 6, 5, stride=2, padding=2, bias=False),
        "conv3": lambda: nn.Conv2d(5, 3, 3, stride=1, padding=1, bias=False),
        "conv1x1b": lambda: nn.Conv2d(4, 7, 1),
        "convT4s2": lambda: nn.ConvTranspose2d(6, 6, 4, stride=2, padding=1, bias=False),
        "convT1x1": lambda: nn.ConvTranspose2d(6, 3, 1, bias=False),
        "conv1d": lambda: nn.Conv1d(8, 8, 1, bias=False),
        "convS1": lambda: nn.Conv2d(8, 2, (4, 1), bias=False),
        "conv1S": lambda: nn.Conv2d(2, 2, (1, 4), bias=False),
        "linear": lambda: nn.Linear(10, 6),
    }[name]()


SN_NAMES = ["conv5s2", "conv3", "conv1x1b", "convT4s2", "convT1x1", "conv1d", "convS1", "conv1S", "linear"]


@pytest.mark.parametrize("name", SN_NAMES)
@pytest.mark.parametrize("uvg", [False, True])
def test_spectral_norm_layers_golden(name, uvg):
    """Three forwards then ONE backward (the D-step pattern): outputs, u/v state after every forward, dx, dW_bar,
    dbias and - once u, v are trainable (main.py:172) - du, dv."""
    from locate_amd import SpectralNorm
    z = load_golden("g4_spectral_norm")
    tag = name + ("_uvg" if uvg else "")
    mod = SpectralNorm(_sn_inner(name))
    mod.load_state_dict(sub(z, tag + "/sd0/"))
    mod = mod.to(dev())
    if uvg:
        mod.requires_grad_(True)
    x = G_(z[tag + "/x"]).requires_grad_(True)
    outs = []
    for k in range(3):
        y = mod(x)
        outs.append(y)
        assert_close(y.cpu(), z[tag + "/y%d" % k], 2e-5, "y%d" % k)
        assert_close(mod.module.weight_u.cpu(), z[tag + "/u%d" % k], 1e-5, "u%d" % k)
        assert_close(mod.module.weight_v.cpu(), z[tag + "/v%d" % k], 1e-5, "v%d" % k)
    sum((o * G_(z[tag + "/g%d" % k])).sum() for k, o in enumerate(outs)).backward()
    assert_close(x.grad.cpu(), z[tag + "/dx"], 3e-5, "dx")
    want = sub(z, tag + "/grad/")
    got = {k: p.grad.cpu() for k, p in mod.named_parameters() if p.grad is not None}
    assert set(got) == set(want), set(got) ^ set(want)
    for k in want:
        # du, dv are dsigma * (W v, W^T u) with dsigma = -<G, W_bar> / sigma^2, a sum with heavy cancellation: rounding
        # differences of 1e-7 in G (summation order; measured against float64 in tools/conv_accuracy.py) show up
        # amplified ~1000x in these two, in the reference's own fp32 arithmetic as much as in ours
        assert_close(got[k], want[k], 5e-4 if k.endswith(("weight_u", "weight_v")) else 1e-4, k)


def test_batched_spectral_norm_equals_per_layer():
    from locate_amd import Discriminator, NetConfig
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    torch.manual_seed(4)
    D1 = Discriminator(cfg).to(dev())
    D2 = Discriminator(cfg).to(dev())
    D2.load_state_dict(D1.state_dict())
    D2.batched_spectral_norm = True
    x = torch.randn(4, 3, 32, 32, device=dev())
    for _ in range(2):
        y1, y2 = D1(x), D2(x)
    assert_close(y2.cpu(), y1.cpu(), 1e-6)
    for (k, a), (_, b) in zip(D1.state_dict().items(), D2.state_dict().items()):
        assert_close(b.cpu(), a.cpu(), 1e-6, k)
    y1.sum().backward()
    y2.sum().backward()
    for (k, a), (_, b) in zip(D1.named_parameters(), D2.named_parameters()):
        if a.grad is not None:
            assert_close(b.grad.cpu(), a.grad.cpu(), 1e-5, k)


@pytest.mark.parametrize("size,ff,B,switches", [(32, 1, 4, {}), (32, 4, 2, {}), (64, 1, 2, {}),
                                                (32, 4, 2, dict(separable=True, depth=2, feature_multiplier=2))])
def test_stacked_discriminator_pass_equals_three_calls(size, ff, B, switches):
    """D over a [3B] batch with stacked=3 == three D calls in order (outputs, u/v state, all gradients incl. u/v)."""
    from locate_amd import Discriminator, NetConfig
    cfg = NetConfig(image_size=size, base_feature_factor=ff, **switches)
    torch.manual_seed(11)
    D1 = Discriminator(cfg).to(dev())
    D2 = Discriminator(cfg).to(dev())
    D2.load_state_dict(D1.state_dict())
    D1.batched_spectral_norm = D2.batched_spectral_norm = True
    D1.requires_grad_(True)          # u, v trainable like after the reference's main.py:172
    D2.requires_grad_(True)
    xs = [torch.randn(B, 3, size, size, device=dev()) * (1 + k) for k in range(3)]     # different statistics per call
    gs = [torch.randn(B, 1, 1, 1, device=dev()) for _ in range(3)]
    ys = [D1(x) for x in xs]
    torch.autograd.backward(ys, [g.view_as(y) for g, y in zip(gs, ys)])
    y_all = D2(torch.cat(xs), stacked=3)
    y_all.backward(torch.cat(gs).view_as(y_all))
    assert_close(y_all.detach().cpu(), torch.cat(ys).detach().cpu(), 2e-5, "outputs")
    for (k, a), (_, b) in zip(D1.state_dict().items(), D2.state_dict().items()):
        assert_close(b.cpu(), a.cpu(), 1e-6, k)
    for (k, a), (_, b) in zip(D1.named_parameters(), D2.named_parameters()):
        assert (a.grad is None) == (b.grad is None), k
        if a.grad is not None:
            # du, dv carry dsigma = -<G, W_bar> / sigma^2 (heavy cancellation, see test_spectral_norm_layers_golden) and the
            # two paths sum <G, W_bar> in different orders (weight side vs activation side)
            assert_close(b.grad.cpu(), a.grad.cpu(), 1e-3 if k.endswith(("weight_u", "weight_v")) else 2e-4, "grad " + k)


# ---------------------------------------------------------------------- dense contractions vs torch CPU
CONV_CASES = [
    # kind, Cin, Cout, k, stride, pad, B, H, W
    ("conv", 3, 3, 5, 2, 2, 4, 16, 16),       # D stem conv_0 (M = 3)
    ("conv", 32, 32, 5, 2, 2, 3, 12, 12),
    ("conv", 64, 64, 5, 2, 2, 2, 9, 7),       # odd sizes
    ("conv", 48, 48, 3, 1, 1, 2, 10, 10),     # G head conv_0
    ("conv", 48, 3, 1, 1, 0, 2, 16, 16),      # G head conv_1 (M = 3)
    ("conv", 512, 1, 1, 1, 0, 8, 1, 1),       # D head conv_1 (M = 1, N = 8)
    ("conv", 100, 200, 1, 1, 0, 5, 6, 6),     # M = 200 (two 128 tiles, ragged), K = 100 (ragged K)
    ("convT", 64, 64, 4, 2, 1, 3, 2, 2),      # G block 0
    ("convT", 96, 96, 4, 2, 1, 2, 8, 8),      # BM = 96 tile
    ("convT", 192, 192, 4, 2, 1, 1, 5, 3),    # odd sizes, BM = 96 x 2
    ("convT", 130, 130, 4, 2, 1, 1, 4, 4),    # ragged M
    ("convT", 96, 48, 1, 1, 0, 2, 8, 8),      # transposed 1x1 (weights [C_in, C_out, 1, 1])
    ("conv", 48, 48, 1, 1, 0, 32, 64, 64),    # pointwise streaming kernel (self-attention gate at 64x64), MT = 48
    ("conv", 48, 3, 1, 1, 0, 130, 64, 64),    # pointwise, M = 3, two pixels per thread
    ("conv", 50, 64, 1, 1, 0, 128, 32, 34),   # pointwise, ragged K (50), MT = 64
    ("convT", 32, 20, 1, 1, 0, 128, 32, 32),  # pointwise through the adjoint panel, MT = 32
    # small / unequal output maps (OW = 4, 12, 8; OH != OW), ragged M
    ("convT", 64, 64, 4, 2, 1, 4, 4, 4),      # OW = 4
    ("conv", 40, 24, 3, 1, 1, 3, 12, 12),     # OW = 12
    ("conv", 16, 130, 5, 2, 2, 5, 16, 16),    # OW = 8, ragged M (130), stride 2 with padding
    ("conv", 24, 32, 3, 1, 1, 2, 4, 8),       # OH != OW
    # boundaries of the K pipeline (the kernels prefetch two stages past the last one into the panel's zero tail):
    ("conv", 272, 256, 3, 1, 1, 8, 16, 16),   # K = 2448 = 16 * 153 exactly, 19 uneven K splits (17 x 9 stages + 1 x 1)
    ("conv", 260, 64, 3, 1, 1, 2, 8, 8),      # ragged K = 2340, split count capped at 64: trailing splits own no stage
    ("convT", 296, 64, 4, 2, 1, 2, 4, 4),     # adjoint phases with K = 1184 = Kpad exactly (no padding rows before the tail)
    ("conv", 128, 128, 1, 1, 0, 4, 8, 8),     # K = Kpad = 128: 8 stages, the split threshold
    ("conv", 342, 192, 3, 1, 1, 3, 6, 8),     # weight gradient on the tall 192 x 128 tile (M % 192 == 0, 25 column tiles): ragged R = 3078, odd batch
    ("convT", 384, 96, 4, 2, 1, 2, 4, 4),     # ... transposed (R's rows are the transposed layer's inputs: M = 384, two tall row tiles)
    # maps so small that most taps only ever see padding (the contraction runs over the useful taps only)
    ("conv", 64, 64, 5, 2, 2, 6, 2, 2),       # 2x2 -> 1x1: 4 of 25 taps (the discriminator's last block)
    ("conv", 32, 48, 3, 1, 1, 5, 1, 1),       # 3x3 p1 on a 1x1 map: the centre tap only (the discriminator's head)
    # 1x1 layers with <= 128 channels over many pixels: the weight gradient straight from global-memory fragments (pw_wgrad_kernel)
    ("conv", 96, 48, 1, 1, 0, 16, 32, 32),    # two column tiles
    ("conv", 40, 100, 1, 1, 0, 8, 32, 32),    # two row tiles, ragged both ways
    ("conv", 24, 32, 1, 1, 0, 64, 6, 12),     # H * W = 72: a multiple of 8 but not of 16 (a step straddles two images)
    ("conv", 3, 29, 1, 1, 0, 12, 64, 64),     # the discriminator's first skip conv: 32 x 32 tiles, mostly masked
    # 1x1 maps: the direct fp32 kernels (skinny_rows_kernel / skinny_wgrad_kernel)
    ("conv", 256, 192, 1, 1, 0, 64, 1, 1),    # a style linear: 8 reduction chunks, one per wave
    ("conv", 832, 384, 1, 1, 0, 64, 1, 1),    # 26 chunks: the prefetch loop; 6 column tiles
    ("conv", 112, 48, 1, 1, 0, 64, 1, 1),     # ragged reduction (112 = 3.5 chunks), one partial column tile
    ("conv", 12, 48, 1, 1, 0, 64, 1, 1),      # reduction shorter than a chunk
    ("conv", 100, 7, 1, 1, 0, 5, 1, 1),       # ragged everything: 5 batch rows (tiles of 4), 7 outputs
    ("convT", 64, 96, 1, 1, 0, 6, 1, 1),      # transposed layer on a 1x1 map (weights [C_in, C_out])
    ("conv", 768, 12, 1, 1, 0, 192, 1, 1),    # batch 192
    ("conv", 16, 16, 5, 2, 2, 3, 3, 3),       # 3x3 -> 2x2: 4 of 5 rows / columns
    ("convT", 32, 32, 4, 2, 1, 3, 1, 1),      # transposed 1x1 -> 2x2: one tap per sub-pixel phase
    ("conv", 24, 160, 5, 2, 2, 6, 2, 2),      # one output pixel, 128+ rows: weight gradient through LDS tiles, rows assembled in LDS
    ("conv", 40, 136, 3, 1, 1, 5, 1, 1),      # ... the head's shape (one useful tap of nine)
    ("conv", 16, 130, 4, 2, 1, 4, 3, 3),      # ... a 3x3 map (9 pixels: not whole channels per 64 columns - scattered stores)
    ("conv", 200, 144, 1, 1, 0, 7, 1, 1),     # 1x1 map, 128+ rows
    ("conv", 8, 8, 5, 2, 2, 2, 1, 5),         # 1 x 5 map: one useful row, all five columns
    # tall 192 x 128 tiles (M a multiple of 192 and >= 512 tiles): adjoint phases and the regular direction
    ("convT", 192, 192, 4, 2, 1, 64, 16, 16),
    ("conv", 48, 192, 3, 1, 1, 64, 32, 32),
    ("convT", 384, 192, 1, 1, 0, 64, 32, 32), # 1x1 transposed, M = 192 (dgrad direction runs with M = 384: two tall tiles)
]


@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,H,W", CONV_CASES)
def test_conv_igemm_vs_cpu(kind, cin, cout, k, s, p, B, H, W):
    """Forward, data gradient and weight gradient of the implicit-GEMM kernels against ATen CPU convs, through
    the SpectralNorm module (so 1/sigma folding and the SN backward are covered too)."""
    from locate_amd import SpectralNorm
    from oracle import locate_oracle as O
    torch.manual_seed(cin * 1000 + cout + k)
    nn = torch.nn
    inner = (nn.Conv2d if kind == "conv" else nn.ConvTranspose2d)(cin, cout, k, stride=s, padding=p, bias=False)
    mod = SpectralNorm(inner)
    sd = {kk: v.clone() for kk, v in mod.state_dict().items()}
    x = torch.randn(B, cin, H, W)
    P = O.make_params(sd)
    xr = x.clone().requires_grad_(True)
    w = O.sn_weight(P, "module.")
    yr = F.conv2d(xr, w, None, s, p) if kind == "conv" else F.conv_transpose2d(xr, w, None, s, p)
    g = torch.randn_like(yr)
    yr.backward(g)
    mod = mod.to(dev())
    xg = x.to(dev()).requires_grad_(True)
    yg = mod(xg)
    yg.backward(g.to(dev()))
    assert_close(yg.cpu(), yr, 2e-5, "y")
    assert_close(xg.grad.cpu(), xr.grad, 2e-5, "dx")
    assert_close(mod.module.weight_bar.grad.cpu(), P["module.weight_bar"].grad, 5e-5, "dw")
    assert_close(mod.module.weight_u.cpu(), P["module.weight_u"], 1e-5, "u")


@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,H,W", CONV_CASES)
def test_conv_fp16_pieces_vs_cpu(kind, cin, cout, k, s, p, B, H, W, monkeypatch):
    """The same comparison with the contractions in their fp16-piece form (two scaled fp16 pieces per operand, three MFMAs per
    slice; csrc/conv.hip precision 2) - held to the SAME bounds as the three-piece bf16 form above.  The largest-magnitude
    words that select this form normally come from the producing kernels; here they are attached to the raw test tensors, and
    the work threshold is lifted so that every geometry takes the path."""
    from locate_amd import SpectralNorm, ops
    from oracle import locate_oracle as O
    monkeypatch.setattr(ops, "F16_MIN_FLOPS", 0.0)
    torch.manual_seed(cin * 1000 + cout + k)
    nn = torch.nn
    inner = (nn.Conv2d if kind == "conv" else nn.ConvTranspose2d)(cin, cout, k, stride=s, padding=p, bias=False)
    mod = SpectralNorm(inner)
    sd = {kk: v.clone() for kk, v in mod.state_dict().items()}
    x = torch.randn(B, cin, H, W) * 3.0
    P = O.make_params(sd)
    xr = x.clone().requires_grad_(True)
    w = O.sn_weight(P, "module.")
    yr = F.conv2d(xr, w, None, s, p) if kind == "conv" else F.conv_transpose2d(xr, w, None, s, p)
    g = torch.randn_like(yr) * 1e-4          # gradient-sized values: far below fp16's normal range without the scaling
    yr.backward(g)
    mod = mod.to(dev())
    xg = ops.tag_amax(x.to(dev()).requires_grad_(True))
    before = dict(ops.F16_CALLS)
    yg = mod(xg)
    yg.backward(ops.tag_amax(g.to(dev())))
    # the weight gradient of a layer with one output pixel is a plain fp32 outer-product kernel at every precision setting (inside a
    # network it is queued for the pass's batched launch and never reaches the piece-form selection; a stand-alone layer asks for
    # the form and the library still takes the fp32 kernel)
    one_pixel = (tuple(yr.shape[2:]) if kind == "conv" else (H, W)) == (1, 1)      # output side of the regular conv R
    assert all(ops.F16_CALLS[kk] == before[kk] + 1 for kk in ("fwd", "dgrad")), (before, ops.F16_CALLS)
    assert ops.F16_CALLS["wgrad"] - before["wgrad"] in ((0, 1) if one_pixel else (1,)), (before, ops.F16_CALLS)
    assert_close(yg.cpu(), yr, 2e-5, "y")
    assert_close(xg.grad.cpu(), xr.grad, 2e-5, "dx")
    assert_close(mod.module.weight_bar.grad.cpu(), P["module.weight_bar"].grad, 5e-5, "dw")
    assert_close(mod.module.weight_u.cpu(), P["module.weight_u"], 1e-5, "u")


# Window form of the contractions (csrc/convwin.hip): geometries whose 128- / 256-column tiles are whole rows or whole images.
# kind, Cin, Cout, k, stride, pad, B, H, W - every tile shape (192 / 128 rows x 128 columns, 96 / 64 / 32 rows x 256, the 64 / 32 x 128
# tiles of single-tap layers), one and two slices per stage, multi-image and multi-row windows, stride-2 gathers (parity
# de-interleave), partial last tiles, split-K with the in-launch combine
WINDOW_CASES = [
    ("convT", 64, 64, 4, 2, 1, 16, 4, 4),       # 2x2-tap phases, eight images per tile, 64 x 256 tiles
    ("convT", 192, 192, 4, 2, 1, 4, 16, 16),    # tall 192 x 128 tiles, eight rows per tile (one halo row)
    ("convT", 128, 96, 4, 2, 1, 3, 8, 8),       # 96 x 256 tiles (adjoint) / 128 x 128 (regular), a partial last tile
    ("conv", 48, 48, 3, 1, 1, 2, 64, 64),       # the generator's 3x3 head: nine taps (ten units, one slice per stage), 64 x 256
    ("conv", 32, 32, 5, 2, 2, 6, 32, 32),       # 5x5 stride 2: 25 taps, de-interleaved window columns, 32 x 256 tiles
    ("conv", 64, 64, 5, 2, 2, 24, 8, 8),        # ... on small maps: 16 images per tile, windows loaded in parts from the item table
    ("conv", 128, 128, 5, 2, 2, 40, 4, 4),      # ... 2x2 outputs: 64 images per tile, a partial last tile
    ("conv", 256, 128, 1, 1, 0, 8, 16, 16),     # single tap: 64 x 128 / 32 x 128 tiles, four channel groups per stage
    ("convT", 384, 192, 1, 1, 0, 4, 8, 8),      # transposed single tap, split-K (few tiles)
    ("conv", 96, 160, 1, 1, 0, 3, 2, 64),       # single tap on a 2 x 64 map (flat pixel runs), ragged M
    ("conv", 48, 48, 1, 1, 0, 5, 2, 2),         # 4-pixel maps: 32 images per tile, partial tile
    ("convT", 24, 12, 1, 1, 0, 8, 16, 16),      # single tap, THREE 8-channel groups under four per stage: the padded group's window
                                                # chunks are read against zero weights and must be finite (cleared LDS; was NaN)
    ("conv", 40, 96, 1, 1, 0, 4, 16, 16),       # single tap, M = 96: the panel is 96 columns wide - 32-row tiles, not 64; five groups
    ("convT", 192, 96, 1, 1, 0, 64, 32, 32),    # ... at the size where the 64-row tile would have been picked
]


@pytest.mark.parametrize("pieces", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,H,W", WINDOW_CASES)
def test_conv_window_form_vs_cpu(kind, cin, cout, k, s, p, B, H, W, pieces, monkeypatch):
    """Forward and input gradient through the window kernels (forced wherever the geometry has the form), weight gradient through
    its own kernels, against ATen CPU convs through the SpectralNorm module - at the bounds of the gather kernels' test above, in
    both fp32-faithful piece forms."""
    from locate_amd import SpectralNorm, ops
    from oracle import locate_oracle as O
    monkeypatch.setattr(ops, "WIN_MODE", 2)
    monkeypatch.setattr(ops, "F16_MIN_FLOPS", 0.0)
    ops._WIN_CACHE.clear()
    torch.manual_seed(cin * 1000 + cout + k)
    nn = torch.nn
    inner = (nn.Conv2d if kind == "conv" else nn.ConvTranspose2d)(cin, cout, k, stride=s, padding=p, bias=False)
    mod = SpectralNorm(inner)
    sd = {kk: v.clone() for kk, v in mod.state_dict().items()}
    x = torch.randn(B, cin, H, W) * 3.0
    P = O.make_params(sd)
    xr = x.clone().requires_grad_(True)
    w = O.sn_weight(P, "module.")
    yr = F.conv2d(xr, w, None, s, p) if kind == "conv" else F.conv_transpose2d(xr, w, None, s, p)
    g = torch.randn_like(yr) * (1e-4 if pieces == "f16x2" else 1.0)
    yr.backward(g)
    mod = mod.to(dev())
    xg = x.to(dev()).requires_grad_(True)
    gg = g.to(dev())
    if pieces == "f16x2":
        xg, gg = ops.tag_amax(xg), ops.tag_amax(gg)
    # which directions have the form in this piece format (three-piece windows of the largest cases exceed the LDS budget)
    spec = ops.ConvSpec(kind, k, k, s, p, p)
    geom, _ = spec.geometry(tuple(x.shape), (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k))
    fmt = 2 if pieces == "f16x2" else 0
    expect = sum(int(ops.lib().locate_conv_win_ok(ops._geom(geom), adj | fmt, 2, 16) > 0) for adj in (0, 1))
    assert expect >= 1
    before = ops.WIN_CALLS[0]
    yg = mod(xg)
    yg.backward(gg)
    assert ops.WIN_CALLS[0] == before + expect, "every direction that has the window form took it"
    assert_close(yg.cpu(), yr, 2e-5, "y")
    assert_close(xg.grad.cpu(), xr.grad, 2e-5, "dx")
    assert_close(mod.module.weight_bar.grad.cpu(), P["module.weight_bar"].grad, 5e-5, "dw")
    ops._WIN_CACHE.clear()


def test_conv_window_form_stacked_calls_and_repack(monkeypatch):
    """The window form under the step's own conditions: three stacked calls with their own 1/sigma per batch third (the epilogue's
    group scales), a bias, an output written into a channel slice, and the panel re-packed after the weights changed (the batched
    re-packing with and without the optimizer's absmax words) - each against the gather kernels on the same operands."""
    from locate_amd import ops
    L = ops.lib()
    torch.manual_seed(5)
    B, C, M, H = 24, 64, 48, 8
    spec = ops.ConvSpec("conv", 5, 5, 2, 2, 2)
    x = ops.tag_amax((torch.randn(B, C, H, H) * 2).to(dev()))
    w = (torch.randn(M, C, 5, 5) * 0.05).to(dev()).requires_grad_(True)
    bias = torch.randn(M).to(dev())
    sigma = torch.tensor([[2.0, 0.5], [4.0, 0.25], [0.5, 2.0]], device=dev())
    geom, out_shape = spec.geometry(tuple(x.shape), tuple(w.shape))
    garr = ops._geom(geom)

    def run(mode):
        monkeypatch.setattr(ops, "WIN_MODE", mode)
        ops._WIN_CACHE.clear()
        buf = torch.zeros(B, M + 8, out_shape[2], out_shape[3], device=dev())
        y = ops._conv_apply(x, w.detach(), w, spec, geom, garr, sigma, bias, out_shape, 0, ops._amax_of(x), buf[:, 8:])
        return y.clone()

    monkeypatch.setattr(ops, "F16_MIN_FLOPS", 0.0)
    before = ops.WIN_CALLS[0]
    y_win = run(2)
    assert ops.WIN_CALLS[0] == before + 1
    y_gather = run(0)
    assert_close(y_win.cpu(), y_gather.cpu(), 2e-6, "stacked window vs gather")
    # the weights change (as after an optimizer step): both panels are stale and re-packed in one batched launch
    with torch.no_grad():
        w.mul_(1.5).add_(0.01)
    ops.refresh_panels([w])
    y_win2 = run(2)
    y_gather2 = run(0)
    assert_close(y_win2.cpu(), y_gather2.cpu(), 2e-6, "after re-packing")
    assert float((y_win2 - y_win).abs().max()) > 1e-3
    ops._WIN_CACHE.clear()


@pytest.mark.parametrize("shape", [(2, 768, 4, 4), (3, 384, 8, 8), (2, 96, 32, 32), (1, 6, 5, 7), (2, 10, 3, 6), (5, 4, 1, 2)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_pool_upsample_equals_the_two_launches(shape):
    """FeaturePooling(C / 2) + bilinear x2 upsample in one launch each way (the generator's skip branch, libs/scale.py:7-16,37-38)
    against the two separate ops: output and input gradient bit for bit - with and without a second consumer of the input adding
    into the same gradient buffer (ops.fork).  Sizes walk the tile / four-wide / scalar kernels and odd widths."""
    from locate_amd import ops
    torch.manual_seed(2)
    x0 = torch.randn(shape)
    B, C, H, W = shape
    g0 = torch.randn(B, C // 2, 2 * H, 2 * W)
    for forked in (False, True):
        res = []
        for fused in (True, False):
            x = x0.to(dev()).requires_grad_(True)
            src = x * 1.0
            if forked:
                src, other = ops.fork(src)
            y = ops.pool_upsample(src, C // 2) if fused else ops.upsample2x(ops.feature_pool(src, C // 2))
            loss_extra = (ops.root_tanh(other) * 0.25).sum() if forked else 0.0
            ops.reset_backward_state()
            torch.autograd.backward([y, loss_extra] if forked else [y], [g0.to(dev()), torch.ones((), device=dev())] if forked else [g0.to(dev())])
            res.append((y.detach().clone(), x.grad.clone()))
        assert torch.equal(res[0][0], res[1][0]), ("forward", forked)
        assert torch.equal(res[0][1], res[1][1]), ("input gradient", forked)
        assert float(res[0][0].abs().max()) > 0 and float(res[0][1].abs().max()) > 0


ACT_LINK_CASES = [
    # (conv class name, cin, mid, cout, kernel, stride, pad, batch, H)  - conv_0 (k x k) -> RootTanh -> conv_1 (1 x 1)
    ("ConvTranspose2d", 192, 192, 96, 4, 2, 1, 64, 16),      # tall tiles, four sub-pixel phases; conv_1 single-tap window form
    ("ConvTranspose2d", 64, 64, 64, 4, 2, 1, 8, 2),          # 2x2 -> 4x4: the staged small-plane epilogue
    ("Conv2d", 48, 48, 48, 3, 1, 1, 16, 32),                 # 3x3; conv_1 through the narrow pointwise stream (M, K <= 64, >= 131072 pixels)
    ("Conv2d", 64, 64, 64, 5, 2, 2, 24, 16),                 # the discriminator's 5x5 stride-2 conv, batch 24
    ("Conv2d", 256, 256, 256, 5, 2, 2, 6, 4),                # deep layer: split-K with the separate reduction launch
    ("Conv2d", 40, 72, 24, 3, 1, 1, 3, 7),                   # ragged everything
    ("Conv2d", 128, 128, 64, 1, 1, 0, 16, 1),                # 1x1 maps: the rows kernel on both sides
]


@pytest.mark.parametrize("case", ACT_LINK_CASES, ids=lambda c: "%s-%d-%d-%d-k%d-s%d-b%d-%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[7], c[8]))
@pytest.mark.parametrize("form", ["f16x2", "bf16x3"])
def test_linked_activation_equals_separate_launches(case, form, monkeypatch):
    """conv_0 -> RootTanh -> conv_1 of a stage (libs/conv.py:19-20) with the activation written by conv_0's epilogue and its
    derivative applied by conv_1's input-gradient epilogue (ops.ActLink) against the same stage with the two RootTanh launches of
    their own: output, input gradient and every parameter gradient bit for bit (same products, same order, same roundings)."""
    from locate_amd import ops
    from locate_amd.nn import ActivatedBaseConv
    import copy
    cls, cin, mid, cout, k, s, pad, B, H = case
    monkeypatch.setattr(ops, "F16_MIN_FLOPS", 0.0 if form == "f16x2" else 1e30)
    monkeypatch.setattr(ops, "AMAX_MIN_NUMEL", [1])
    monkeypatch.setattr(ops, "ACT_LINKS", [True])
    monkeypatch.setattr(ops, "ACT_LINK_MAX_NUMEL", [1 << 40])
    torch.manual_seed(11)

    class Cfg:
        feature_multiplier = 1
        separable = False
    stage = ActivatedBaseConv(cin, cout, getattr(torch.nn, cls), kernel=k, stride=s, pad=pad, cfg=Cfg)
    if mid != cin:          # (the reference ties the bottleneck width to the input's; widen it for the ragged case)
        from locate_amd import SpectralNorm
        stage.conv_0 = SpectralNorm(getattr(torch.nn, cls)(cin, mid, k, s, pad, bias=False))
        stage.conv_1 = SpectralNorm(getattr(torch.nn, cls)(mid, cout, 1, 1, 0, bias=False))
    x0 = torch.randn(B, cin, H, H) * 1.5
    results = []
    for linked in (True, False):
        ops.ACT_LINKS[0] = linked
        st = copy.deepcopy(stage).to(dev())
        for prm in st.parameters():
            prm.requires_grad_(True)
        x = ops.tag_amax(x0.to(dev())).requires_grad_(True)
        before = (ops.F16_CALLS["fwd"], ops.F16_CALLS["dgrad"])
        y = st(x, pre_activated=True)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dev())
        ops.reset_backward_state()
        y.backward(ops.tag_amax(gy))
        torch.cuda.synchronize()
        if form == "f16x2" and H > 1:
            assert ops.F16_CALLS["fwd"] > before[0], "the fp16-piece form was not taken"
        results.append((y.detach().clone(), x.grad.clone(), [(n, prm.grad.clone()) for n, prm in st.named_parameters() if prm.grad is not None]))
    (ya, gxa, pa), (yb, gxb, pb) = results
    assert torch.isfinite(ya).all() and float(ya.abs().max()) > 0
    assert torch.equal(ya, yb), "stage output"
    assert torch.equal(gxa, gxb), "input gradient"
    assert [n for n, _ in pa] == [n for n, _ in pb] and len(pa) >= 4
    for (n, a), (_, b) in zip(pa, pb):
        assert torch.equal(a, b), n


def test_stale_largest_magnitude_tag_is_ignored(monkeypatch):
    """A tensor's largest-magnitude words are only valid for the data they were taken from: when autograd sums a second consumer's
    gradient INTO a tagged gradient (in place), the contraction that consumes the sum must not scale its fp16 pieces by the old
    maximum (a sum 4x larger overflows the high piece to inf).  The tag carries the tensor's version; a stale one is ignored and
    the contraction takes the six-product form.  Here: y = conv(x) feeds RootTanh (whose backward tags its gradient) AND a plain
    torch op whose gradient is 30x larger."""
    from locate_amd import SpectralNorm, ops
    from oracle import locate_oracle as O
    monkeypatch.setattr(ops, "F16_MIN_FLOPS", 0.0)
    t = ops.tag_amax(torch.randn(1 << 16, device=dev()))
    assert ops._amax_of(t) is not None
    t.add_(1.0)
    assert ops._amax_of(t) is None, "an in-place change invalidates the tag"
    torch.manual_seed(3)
    inner = torch.nn.Conv2d(32, 48, 3, stride=1, padding=1, bias=False)
    mod = SpectralNorm(inner)
    sd = {kk: v.clone() for kk, v in mod.state_dict().items()}
    x = torch.randn(8, 32, 32, 32)
    P = O.make_params(sd)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, O.sn_weight(P, "module."), None, 1, 1)
    g1, g2 = torch.randn_like(yr) * 1e-3, torch.randn_like(yr)
    (O.RootTanhFn.apply(yr) * g1).sum().backward(retain_graph=True)
    (yr * 30.0 * g2).sum().backward()
    mod = mod.to(dev())
    xg = ops.tag_amax(x.to(dev()).requires_grad_(True))
    yg = mod(xg)
    # RootTanh's branch is created last, so its backward runs first: its tagged gradient is the buffer the engine adds the other
    # branch's (much larger) gradient into
    big = yg * 30.0
    small = ops.root_tanh(yg)
    ((small * g1.to(dev())).sum() + (big * g2.to(dev())).sum()).backward()
    assert torch.isfinite(xg.grad).all() and torch.isfinite(mod.module.weight_bar.grad).all()
    assert_close(xg.grad.cpu(), xr.grad, 2e-5, "dx")
    assert_close(mod.module.weight_bar.grad.cpu(), P["module.weight_bar"].grad, 5e-5, "dw")


GROUPED_CASES = [
    # kind, Cin, mult, k, stride, pad, B, H, W        (groups = Cin: the SEPARABLE switch, libs/conv.py:17)
    ("conv", 3, 1, 5, 2, 2, 4, 16, 16),        # D stem conv_0
    ("conv", 32, 1, 5, 2, 2, 3, 12, 12),
    ("conv", 20, 2, 5, 2, 2, 2, 9, 7),         # FEATURE_MULTIPLIER = 2, odd sizes
    ("conv", 48, 1, 3, 1, 1, 2, 10, 10),       # G head conv_0
    ("conv", 12, 3, 5, 1, 2, 2, 8, 8),         # DEPTH > 1 middle stage, multiplier 3
    ("conv", 96, 1, 5, 2, 2, 40, 32, 32),      # many chunks in the weight-gradient reduction
    ("convT", 64, 1, 4, 2, 1, 3, 2, 2),        # G block 0
    ("convT", 24, 2, 4, 2, 1, 2, 8, 8),        # transposed with multiplier (weights [C_in, 2, 4, 4])
    ("convT", 10, 1, 4, 2, 1, 1, 5, 3),        # odd sizes
    ("full", 32, 8, 8, 1, 0, 5, 8, 8),         # feature attention: Conv2d(32 -> 8, kernel 8x8, groups = 8)
    ("full", 12, 3, 16, 1, 0, 3, 16, 16),      # 4 channels per group, L = 1024
    ("full", 6, 3, 5, 1, 0, 2, 5, 5),          # L = 50: not a multiple of 4
]


def _grouped_layer(kind, cin, mult, k, s, p):
    nn = torch.nn
    if kind == "conv":
        return nn.Conv2d(cin, cin * mult, k, stride=s, padding=p, bias=False, groups=cin), cin
    if kind == "convT":
        return nn.ConvTranspose2d(cin, cin * mult, k, stride=s, padding=p, bias=False, groups=cin), cin
    return nn.Conv2d(cin, mult, k, bias=False, groups=mult), mult        # "full": mult = number of groups = outputs


def _grouped_apply(kind, x, w, s, p, groups):
    if kind == "convT":
        return F.conv_transpose2d(x, w, None, s, p, groups=groups)
    return F.conv2d(x, w, None, s, p, groups=groups)


@pytest.mark.parametrize("kind,cin,mult,k,s,p,B,H,W", GROUPED_CASES)
def test_grouped_convs_vs_cpu(kind, cin, mult, k, s, p, B, H, W):
    """Depthwise (regular / transposed, with channel multiplier) and full-size grouped convs: forward, data gradient,
    weight gradient and the spectral-norm state / u, v gradients against ATen CPU grouped convs through the oracle's
    spectral norm."""
    from locate_amd import SpectralNorm
    from oracle import locate_oracle as O
    torch.manual_seed(cin * 100 + mult + k)
    inner, groups = _grouped_layer(kind, cin, mult, k, s, p)
    mod = SpectralNorm(inner)
    sd = {kk: v.clone() for kk, v in mod.state_dict().items()}
    x = torch.randn(B, cin, H, W)
    P = O.make_params(sd, trainable_uv=True)
    xr = x.clone().requires_grad_(True)
    yr = _grouped_apply(kind, xr, O.sn_weight(P, "module."), s, p, groups)
    g = torch.randn_like(yr)
    yr.backward(g)
    mod = mod.to(dev())
    mod.requires_grad_(True)
    xg = x.to(dev()).requires_grad_(True)
    yg = mod(xg)
    yg.backward(g.to(dev()))
    assert_close(yg.cpu(), yr, 2e-5, "y")
    assert_close(xg.grad.cpu(), xr.grad, 2e-5, "dx")
    assert_close(mod.module.weight_bar.grad.cpu(), P["module.weight_bar"].grad, 5e-5, "dw")
    assert_close(mod.module.weight_u.cpu(), P["module.weight_u"], 1e-5, "u")
    assert_close(mod.module.weight_u.grad.cpu(), P["module.weight_u"].grad, 5e-4, "du")
    assert_close(mod.module.weight_v.grad.cpu(), P["module.weight_v"].grad, 5e-4, "dv")


@pytest.mark.parametrize("kind,cin,mult,k,s,p,B,H,W", [GROUPED_CASES[2], GROUPED_CASES[7], GROUPED_CASES[10]])
def test_grouped_convs_stacked_calls(kind, cin, mult, k, s, p, B, H, W):
    """One pass over three stacked calls (own sigma per call) == the three calls in order, for the grouped kernels."""
    from locate_amd import SpectralNorm, ops
    torch.manual_seed(5)
    inner, _ = _grouped_layer(kind, cin, mult, k, s, p)
    m1 = SpectralNorm(inner).to(dev())
    import copy
    m2 = copy.deepcopy(m1)
    m1.requires_grad_(True)
    m2.requires_grad_(True)
    xs = [torch.randn(B, cin, H, W, device=dev()) for _ in range(3)]
    ys = [m1(x) for x in xs]
    gs = [torch.randn_like(y) for y in ys]
    torch.autograd.backward(ys, gs)
    mm = m2.module
    runs = [ops.sn_power_iteration(mm.weight_bar, mm.weight_u, mm.weight_v) for _ in range(3)]
    m2._pre = (torch.stack([r[0] for r in runs]), torch.stack([r[1] for r in runs]))
    with ops.stacked_calls(3):
        y_all = m2(torch.cat(xs))
    y_all.backward(torch.cat(gs))
    assert_close(y_all.detach().cpu(), torch.cat(ys).detach().cpu(), 1e-5, "y")
    for name in ("weight_bar", "weight_u", "weight_v"):
        a, b = getattr(m1.module, name), getattr(m2.module, name)
        assert_close(b.detach().cpu(), a.detach().cpu(), 1e-6, name)
        assert_close(b.grad.cpu(), a.grad.cpu(), 1e-3 if name != "weight_bar" else 1e-4, "grad " + name)


def test_conv_on_channel_slice_view():
    """Batch-strided inputs (a channel slice of a bigger NCHW tensor) are consumed in place."""
    from locate_amd import SpectralNorm
    torch.manual_seed(8)
    mod = SpectralNorm(torch.nn.Conv2d(6, 10, 1))
    mod2 = SpectralNorm(torch.nn.Conv2d(6, 10, 1))
    mod2.load_state_dict(mod.state_dict())
    mod, mod2 = mod.to(dev()), mod2.to(dev())
    big = torch.randn(3, 16, 5, 5, device=dev())
    view = big[:, 4:10]
    assert not view.is_contiguous()
    y1 = mod(view)
    y2 = mod2(view.contiguous())
    assert_close(y1.cpu(), y2.cpu(), 1e-6)


# ------------------------------------------------------------------------------------------------ G5
def test_indexing_bit_exact_and_resampling():
    from locate_amd import ops
    z = load_golden("g5_indexing")
    for r in (4, 2):
        x = G_(z["fpool%d_x" % r]).requires_grad_(True)
        y = ops.feature_pool(x, r)
        y.backward(G_(z["fpool%d_g" % r]))
        if r == 4:      # C_in / C_out = 2: the only ratio in the architecture -> bit exact
            assert torch.equal(y.cpu(), T(z["fpool%d_y" % r]))
            assert torch.equal(x.grad.cpu(), T(z["fpool%d_dx" % r]))
        else:
            assert_close(y.cpu(), z["fpool%d_y" % r], 2e-7)
            assert torch.equal(x.grad.cpu(), T(z["fpool%d_dx" % r]))
    x = G_(z["up_x"]).requires_grad_(True)
    y = ops.upsample2x(x)
    y.backward(G_(z["up_g"]))
    assert_close(y.cpu(), z["up_y"], 1e-6)
    assert_close(x.grad.cpu(), z["up_dx"], 1e-6)
    x = G_(z["pool_x"]).requires_grad_(True)
    y = ops.avgpool2(x)
    y.backward(G_(z["pool_g"]))
    assert_close(y.cpu(), z["pool_y"], 1e-6)
    assert_close(x.grad.cpu(), z["pool_dx"], 1e-6)
    from locate_amd import Expand, ResModule
    e = Expand(-1, 5, 4, 4)
    t = G_(z["expand_x"]).requires_grad_(True)
    out = e(t)
    assert torch.equal(out.cpu(), T(z["expand_y"]))
    out.backward(G_(z["expand_g"]))
    assert_close(t.grad.cpu(), z["expand_dx"], 1e-6)


@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (2, 3, 4, 6), (3, 5, 8, 8), (1, 2, 2, 2), (2, 4, 6, 12), (2, 2, 1, 1), (64, 6, 32, 32),
                                   (2, 3, 3, 16), (1, 1, 1, 8), (5, 7, 16, 16)])
def test_resampling_shapes_vs_aten(shape):
    """Bilinear x2 and 2x2 average pooling (forward and backward) against ATen on odd / even / vectorisable widths: the
    scalar, the 16-byte-store and the 2 x 8 patch kernels (W a multiple of 4, from 8 up: odd heights, one-row maps) must agree
    with the same reference."""
    from locate_amd import ops
    torch.manual_seed(sum(shape))
    x = torch.randn(shape)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)
    g = torch.randn_like(yr)
    yr.backward(g)
    xg = x.to(dev()).requires_grad_(True)
    yg = ops.upsample2x(xg)
    yg.backward(g.to(dev()))
    assert_close(yg.cpu(), yr, 1e-6, "upsample")
    assert_close(xg.grad.cpu(), xr.grad, 1e-6, "upsample dx")
    if shape[2] >= 2 and shape[3] >= 2:
        xr = x.clone().requires_grad_(True)
        yr = F.avg_pool2d(xr, 2, 2)
        g = torch.randn_like(yr)
        yr.backward(g)
        xg = x.to(dev()).requires_grad_(True)
        yg = ops.avgpool2(xg)
        yg.backward(g.to(dev()))
        assert_close(yg.cpu(), yr, 1e-6, "avgpool")
        assert torch.equal(xg.grad.cpu(), xr.grad), "avgpool dx"


def test_cat_channels():
    from locate_amd import ops
    torch.manual_seed(2)
    a = torch.randn(3, 4, 5, 5, device=dev(), requires_grad=True)
    b = torch.randn(3, 7, 5, 5, device=dev(), requires_grad=True)
    out = ops.cat_channels(a, b)
    assert torch.equal(out, torch.cat([a, b], 1))
    g = torch.randn_like(out)
    out.backward(g)
    assert torch.equal(a.grad, g[:, :4]) and torch.equal(b.grad, g[:, 4:])


@pytest.mark.parametrize("name,args", [("scale_up_pool", (8, 4, 2, True)), ("scale_up_cat", (4, 12, 2, True)),
                                       ("scale_down_cat", (4, 8, 2, False)), ("scale_down_same", (6, 6, 2, False))])
def test_scale_compositions_golden(name, args):
    from locate_amd import Scale
    z = load_golden("g5_indexing")
    layer = Scale(*args)
    if isinstance(layer, torch.nn.Module):
        layer.load_state_dict(sub(z, name + "/sd0/"))
        layer = layer.to(dev())
    x = G_(z[name + "/x"]).requires_grad_(True)
    y = layer(x)
    y.backward(G_(z[name + "/g"]))
    assert_close(y.cpu(), z[name + "/y"], 1e-5)
    assert_close(x.grad.cpu(), z[name + "/dx"], 2e-5)
    if isinstance(layer, torch.nn.Module):
        for k, v in sub(z, name + "/grad/").items():
            assert_close(dict(layer.named_parameters())[k].grad.cpu(), v, 5e-5, k)
        for k, v in sub(z, name + "/sd1/").items():
            assert_close(layer.state_dict()[k].cpu(), v, 1e-5, k)


# ------------------------------------------------------------------------------------------------ softmax
@pytest.mark.parametrize("rows,n", [(7, 3), (64 * 16, 64), (96, 256), (33, 1000), (64 * 48, 4096), (5, 65536), (64, 192),
                                    (9, 1028), (11, 2048), (6, 2500), (5, 3000), (3, 4094), (7, 5000)])
def test_softmax_rows(rows, n):
    from locate_amd import ops
    torch.manual_seed(rows + n)
    x = torch.randn(rows, n) * 3
    g = torch.randn(rows, n)
    xr = x.clone().requires_grad_(True)
    yr = torch.softmax(xr, -1)
    yr.backward(g)
    xg = x.to(dev()).requires_grad_(True)
    yg = ops.softmax_lastdim(xg)
    yg.backward(g.to(dev()))
    assert_close(yg.cpu(), yr, 2e-6)
    assert_close(xg.grad.cpu(), xr.grad, 2e-5)


# ------------------------------------------------------------------------------------------------ G6
def test_attention_layers_and_linear_golden():
    from locate_amd import LinearModule, NetConfig, SelfAttention, feature_attention
    z = load_golden("g6_attention")
    cfg = NetConfig()
    for name, mod in (("fa", feature_attention(8, 16, cfg=cfg)), ("sa", SelfAttention(16))):
        mod.load_state_dict(sub(z, name + "/sd0/"))
        mod = mod.to(dev())
        x = G_(z[name + "/x"]).requires_grad_(True)
        y = mod(x)
        y.backward(G_(z[name + "/g"]))
        assert_close(y.cpu(), z[name + "/y"], 1e-5, name)
        assert_close(x.grad.cpu(), z[name + "/dx"], 5e-5, name + " dx")
        params = dict(mod.named_parameters())
        for k, v in sub(z, name + "/grad/").items():
            assert_close(params[k].grad.cpu(), v, 2e-4, name + k)
        for k, v in sub(z, name + "/sd1/").items():
            assert_close(mod.state_dict()[k].cpu(), v, 1e-5, k)
    lm = LinearModule(12, 7)
    lm.load_state_dict(sub(z, "lin/sd0/"))
    lm = lm.to(dev())
    x = G_(z["lin/x"]).requires_grad_(True)
    act, pre = lm(x)
    ((act * G_(z["lin/g_act"])).sum() + (pre * G_(z["lin/g_pre"])).sum()).backward()
    assert_close(act.cpu(), z["lin/act"], 1e-5)
    assert_close(pre.cpu(), z["lin/pre"], 1e-5)
    assert_close(x.grad.cpu(), z["lin/dx"], 3e-5)
    params = dict(lm.named_parameters())
    for k, v in sub(z, "lin/grad/").items():
        assert_close(params[k].grad.cpu(), v, 1e-4, k)


# ------------------------------------------------------------------------------------------------ G7
@pytest.mark.parametrize("name,size,cin,cout,idx,transposed", [
    ("up", 8, 16, 8, 0, True), ("up_na", 4, 8, 8, 1, True), ("down", 8, 8, 16, 0, False), ("down_na", 4, 16, 16, 1, False)])
def test_blocks_golden(name, size, cin, cout, idx, transposed):
    from locate_amd import Block, NetConfig
    z = load_golden("g7_blocks")
    blk = Block(size, cin, cout, 2, transposed, idx, cfg=NetConfig())
    blk.load_state_dict(sub(z, name + "/sd0/"))
    blk = blk.to(dev())
    x = G_(z[name + "/x"]).requires_grad_(True)
    scales = None
    if transposed:
        scales = [G_(z[name + "/scale%d" % i]).requires_grad_(True) for i in range(3) if name + "/scale%d" % i in z.files]
    y = blk(x, scales)
    y.backward(G_(z[name + "/g"]))
    assert_close(y.cpu(), z[name + "/y"], 3e-5, "y")
    assert_close(x.grad.cpu(), z[name + "/dx"], 2e-4, "dx")
    if scales:
        for i, s in enumerate(scales):
            assert_close(s.grad.cpu(), z[name + "/dscale%d" % i], 2e-4, "dscale%d" % i)
    want = sub(z, name + "/grad/")
    got = {k: p.grad.cpu() for k, p in blk.named_parameters() if p.grad is not None}
    assert set(got) == set(want), set(got) ^ set(want)
    for k, v in want.items():
        assert_close(got[k], v, 3e-4, k)
    for k, v in sub(z, name + "/sd1/").items():
        assert_close(blk.state_dict()[k].cpu(), v, 1e-5, k)


# ------------------------------------------------------------------------------------------------ G9
def test_nadam_golden():
    from locate_amd import Nadam
    z = load_golden("g9_nadam")
    ps = [torch.nn.Parameter(G_(z["p%d_0" % i]).clone()) for i in range(3)]
    opt = Nadam(ps, lr=float(z["lr"]), betas=tuple(float(b) for b in z["betas"]))
    for step in range(1, 4):
        for i, p in enumerate(ps):
            key = "g%d_%d" % (i, step)
            p.grad = G_(z[key]).clone() if key in z.files else None
        opt.step()
        for i, p in enumerate(ps):
            assert_close(p.detach().cpu(), z["p%d_%d" % (i, step)], 2e-6, "p%d step %d" % (i, step))


def test_loss_kernels():
    from locate_amd.train import d_loss, g_loss
    from oracle import locate_oracle as O
    torch.manual_seed(1)
    for B in (8, 64, 300):
        t, f, a = (torch.randn(B) * 2 for _ in range(3))
        t[0], f[0] = 1.0, -1.0          # hinge arguments exactly 0: clamp passes the gradient there
        tr, fr, ar = (v.clone().requires_grad_(True) for v in (t, f, a))
        d_err = (O.hinge(tr) + O.hinge(-fr)).mean()       # main.py:150-155: d_gen = -D(generated)
        pen = O.consistency_penalty(tr, ar)
        (d_err + pen).backward()
        losses, gt, gf, ga = d_loss(t.to(dev()), f.to(dev()), a.to(dev()))
        assert_close(losses.cpu(), torch.stack([d_err, pen, d_err + pen]).detach(), 2e-6)
        assert_close(gt.cpu(), tr.grad, 1e-5)
        assert_close(gf.cpu(), fr.grad, 1e-6)
        assert_close(ga.cpu(), ar.grad, 1e-5)
        fr2 = f.clone().requires_grad_(True)
        ge = O.hinge(fr2).mean()
        ge.backward()
        loss, g = g_loss(f.to(dev()))
        assert_close(loss.cpu(), ge.detach().reshape(1), 2e-6)
        assert_close(g.cpu(), fr2.grad, 1e-6)


# ---------------------------------------------------------------------- bf16 x 6 vs the fp32-input MFMA kernels
_XCHECK = r"""
import sys, torch
sys.path.insert(0, %r)
from locate_amd import ops
dev = torch.device("cuda:0")
out = {}
for idx, (kind, cin, cout, k, s, p, B, H, W) in enumerate([("conv", 48, 64, 5, 2, 2, 8, 32, 32), ("convT", 192, 192, 4, 2, 1, 8, 8, 8),
                                                          ("conv", 96, 96, 3, 1, 1, 4, 16, 16)]):
    torch.manual_seed(100 + idx)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = (torch.randn(wshape) * 0.1).to(dev).requires_grad_(True)
    x = torch.randn(B, cin, H, W).to(dev).requires_grad_(True)
    h = wshape[0]
    u, v = torch.randn(h, device=dev), torch.randn(w.numel() // h, device=dev)
    sigma, wv = torch.tensor([1.0, 1.0], device=dev), torch.zeros(h, device=dev)
    y = ops.SNConvFn.apply(x, w, u, v, None, sigma, wv, ops.ConvSpec(kind, k, k, s, p, p))
    torch.manual_seed(200 + idx)
    y.backward(torch.randn(y.shape).to(dev))
    out["y%%d" %% idx], out["dx%%d" %% idx], out["dw%%d" %% idx] = y.detach().cpu(), x.grad.cpu(), w.grad.cpu()
torch.save(out, sys.argv[1])
"""


def test_bf16x6_kernels_match_fp32_mfma_kernels(tmp_path):
    """The default contractions (exact three-way bf16 splits, six bf16 MFMAs) against the fp32-input MFMA kernels kept
    in the DEBUG variant of the library as the cross-check (liblocate_hip_dbg.so, LOCATE_DISABLE=bx6,wbx6 read once per
    process - hence the two child processes; the product library has no such switch)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("bx6", {}), ("f32", {"LOCATE_HIP_DEBUG_LIBRARY": "1", "LOCATE_DISABLE": "bx6,wbx6"})):
        path = str(tmp_path / (tag + ".pt"))
        subprocess.check_call([sys.executable, "-c", _XCHECK % root, path], env=dict(os.environ, **env))
        res[tag] = torch.load(path, weights_only=True)
    for k in res["f32"]:
        assert_close(res["bx6"][k], res["f32"][k], 3e-6 if k.startswith("y") or k.startswith("dx") else 3e-5, k)


@pytest.mark.parametrize("C,size", [(768, 4), (96, 32)])
def test_conv_stage_linearity_at_benchmark_size(C, size):
    """Size-independent property at BASELINE's full sizes (batch 64): the ConvTranspose 4x4 s2 stage is linear in its
    input and in its weight - f(x1 + x2) = f(x1) + f(x2), f(a x) = a f(x), and the weight gradient of a sum of output
    gradients is the sum of the weight gradients."""
    from locate_amd import ops
    torch.manual_seed(C)
    B = 64
    w = (torch.randn(C, C, 4, 4, device=dev()) * 0.02).requires_grad_(True)
    u, v = torch.randn(C, device=dev()), torch.randn(C * 16, device=dev())
    sigma, wv = torch.tensor([2.0, 0.5], device=dev()), torch.zeros(C, device=dev())
    spec = ops.ConvSpec("convT", 4, 4, 2, 1, 1)
    x1, x2 = torch.randn(B, C, size, size, device=dev()), torch.randn(B, C, size, size, device=dev())

    def f(x):
        return ops.SNConvFn.apply(x, w, u, v, None, sigma, wv, spec)
    with torch.no_grad():
        y1, y2, y12, y3 = f(x1), f(x2), f(x1 + x2), f(3.0 * x1)
    assert_close(y12.cpu(), (y1 + y2).cpu(), 3e-6, "additivity")
    assert_close(y3.cpu(), (3.0 * y1).cpu(), 3e-6, "homogeneity")
    g1, g2 = torch.randn_like(y1), torch.randn_like(y1)
    grads = []
    for g in (g1, g2, g1 + g2):
        w.grad = None
        f(x1).backward(g)
        # (the rank-1 spectral-norm term dsigma u v^T is linear in g as well)
        grads.append(w.grad.clone())
    assert_close(grads[2].cpu(), (grads[0] + grads[1]).cpu(), 2e-5, "weight-gradient additivity")


@pytest.mark.parametrize("C,size,B", [(768, 4, 64), (384, 8, 64), (96, 8, 2)])
def test_split_k_combined_in_launch_is_reproducible_and_matches_the_reduction_kernel(C, size, B, monkeypatch):
    """Split-K launches that combine their partial tiles inside the launch (arrival counters, last block sums in z order):
    (1) bit-identical results over repeated launches while another stream keeps the chip unevenly busy and the consumer's
    caches are warm (the hand-off must not depend on timing or placement), (2) the counters are left zero, (3) the same
    values as the two-kernel path (partial tiles + reduction kernel) up to the fused multiply-add of the epilogue."""
    from locate_amd import ops
    torch.manual_seed(C + size)
    w = (torch.randn(C, C, 4, 4, device=dev()) * 0.02)
    u, v = torch.randn(C, device=dev()), torch.randn(C * 16, device=dev())
    sigma, wv = torch.tensor([2.0, 0.5], device=dev()), torch.zeros(C, device=dev())
    spec = ops.ConvSpec("convT", 4, 4, 2, 1, 1)
    x = torch.randn(B, C, size, size, device=dev())
    bias = torch.randn(C, device=dev())

    def run():
        with torch.no_grad():
            return ops.SNConvFn.apply(x, w, u, v, bias, sigma, wv, spec)
    first = run().clone()
    side = torch.cuda.Stream()
    junk = torch.randn(1 << 24, device=dev())
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(1 + it % 3):
                junk.mul_(1.0001)                 # uneven background load on the other stream
        again = run()
        assert torch.equal(again, first), "launch %d differs" % it
    torch.cuda.synchronize()
    counters = w.__dict__["_locate_counters"]
    assert counters and all(int(c.view(torch.int32).abs().sum()) == 0 for c in counters.values())
    monkeypatch.setattr(ops, "_counters", lambda owner, adjoint: None)      # NULL counters: the reduction-kernel path
    legacy = run()
    assert_close(first.cpu(), legacy.cpu(), 1e-6, "in-launch combine vs reduction kernel")



@pytest.mark.parametrize("first,second", [("gate", "norm"), ("gate", "roottanh"), ("feature_pool", "norm"), ("avgpool", "norm"),
                                          ("norm", "gate")])
def test_forked_tensor_second_backward_kernel_accumulates(first, second):
    """ops.fork: the backward kernels of a tensor's two consumers share one gradient buffer (the second to run adds into it,
    `accumulate` of the C ABI) - against the same graph with autograd forming the sum."""
    from locate_amd import ops
    torch.manual_seed(11)
    B, C, H, W = 6, 16, 12, 12
    x0 = torch.randn(B, C, H, W, device=dev())
    gamma0 = torch.full((1, 1), 1.7, device=dev())
    scale0, bias0 = torch.randn(1, C, 1, 1, device=dev()), torch.randn(1, C, 1, 1, device=dev())
    branch0 = torch.randn(B, C, H, W, device=dev())

    def consume(kind, t, leaves):
        if kind == "gate":
            return ops.residual_gate(t, leaves["branch"], leaves["gamma"]).square().sum()
        if kind == "norm":
            return (ops.inplace_norm(t, leaves["scale"], leaves["bias"], False) * leaves["branch"]).sum()
        if kind == "roottanh":
            return (ops.root_tanh(t) * leaves["branch"]).sum()
        if kind == "feature_pool":
            return ops.feature_pool(t, C // 2).square().sum()
        return ops.avgpool2(t).square().sum()

    grads = []
    for forked in (True, False):
        leaves = {"gamma": gamma0.clone().requires_grad_(True), "scale": scale0.clone().requires_grad_(True),
                  "bias": bias0.clone().requires_grad_(True), "branch": branch0.clone().requires_grad_(True)}
        x = x0.clone().requires_grad_(True)
        h = x * 1.0                     # a non-leaf, as inside the networks
        a, b = ops.fork(h) if forked else (h, h)
        assert (getattr(a, "_locate_slot", None) is not None) == forked
        (consume(first, a, leaves) + consume(second, b, leaves)).backward()
        grads.append([x.grad] + [leaves[k].grad for k in ("gamma", "scale", "bias", "branch")])
    for name, got, want in zip(("dx", "dgamma", "dscale", "dbias", "dbranch"), *grads):
        if got is None and want is None:
            continue
        assert_close(got.cpu(), want.cpu(), 2e-6, name)


@pytest.mark.gpu
def test_upsample_kernel_variants_agree_bit_for_bit():
    """The three forward kernels of the bilinear x2 upsample (scalar, four-wide, 2 x 8 patch) are picked by width and alignment;
    they evaluate the same stencil with the same roundings: a tensor wide enough for the patch kernel, and the same values seen
    through a view whose rows start 4 bytes off 16-byte alignment (scalar kernel), must give identical bits."""
    from locate_amd import ops
    torch.manual_seed(5)
    x = torch.randn(3, 4, 8, 16, device=dev())
    y_tile = ops.upsample2x(x)                                   # W = 16: the 2 x 8 patch kernel
    wide = torch.randn(3, 4, 8, 18, device=dev())
    wide[..., 1:17] = x
    y_wide = ops.upsample2x(wide)                                # W = 18: the four-wide kernel; interior columns see the same stencil
    assert torch.equal(y_tile[..., 2:30], y_wide[..., 4:32])
    odd = torch.randn(3, 4, 8, 17, device=dev())
    odd[..., :16] = x
    y_odd = ops.upsample2x(odd)                                  # W = 17: the scalar kernel
    assert torch.equal(y_tile[..., :30], y_odd[..., :30])


@pytest.mark.gpu
def test_branches_the_training_loop_never_takes():
    """Nadam with weight_decay and SpectralNorm(power_iterations > 1) on the MI355X against the reference's records (g21)."""
    from locate_amd import Nadam, SpectralNorm
    z = load_golden("g21_branches")
    ps = [torch.nn.Parameter(G_(z["wd/p%d_0" % i]).clone()) for i in range(2)]
    opt = Nadam(ps, lr=float(z["wd/lr"]), betas=tuple(float(b) for b in z["wd/betas"]), weight_decay=float(z["wd/weight_decay"]))
    for step in range(1, 4):
        for i, p in enumerate(ps):
            p.grad = G_(z["wd/g%d_%d" % (i, step)]).clone()
        opt.step()
        for i, p in enumerate(ps):
            assert_close(p.detach().cpu(), z["wd/p%d_%d" % (i, step)], 2e-6, "p%d step %d" % (i, step))
    nn = torch.nn
    for name, inner in (("conv3", nn.Conv2d(5, 3, 3, stride=1, padding=1, bias=False)),
                        ("convT4s2", nn.ConvTranspose2d(6, 6, 4, stride=2, padding=1, bias=False))):
        tag = "pi/" + name
        mod = SpectralNorm(inner, power_iterations=int(z[tag + "/iters"]))
        mod.load_state_dict(sub(z, tag + "/sd0/"))
        mod = mod.to(dev())
        mod.requires_grad_(True)
        x = G_(z[tag + "/x"]).requires_grad_(True)
        outs = []
        for k in range(2):
            y = mod(x)
            outs.append(y)
            assert_close(y.cpu(), z[tag + "/y%d" % k], 2e-5, name + " y%d" % k)
            assert_close(mod.module.weight_u.cpu(), z[tag + "/u%d" % k], 1e-5, name + " u%d" % k)
            assert_close(mod.module.weight_v.cpu(), z[tag + "/v%d" % k], 1e-5, name + " v%d" % k)
        sum((o * G_(z[tag + "/g%d" % k])).sum() for k, o in enumerate(outs)).backward()
        assert_close(x.grad.cpu(), z[tag + "/dx"], 3e-5, name + " dx")
        want = sub(z, tag + "/grad/")
        got = {k: p.grad.cpu() for k, p in mod.named_parameters() if p.grad is not None}
        assert set(got) == set(want)
        for k in want:
            assert_close(got[k], want[k], 5e-4 if k.endswith(("weight_u", "weight_v")) else 1e-4, name + " " + k)


@pytest.mark.gpu
def test_style_link_fused_equals_separate_launches():
    """A style-chain link as ONE launch (linear + bias, its RootTanh and the latent columns in front: the next link's input,
    nn.LinearModule.pre_and_next_input) against the same link as linear -> act_cat (two launches, ops.ActCatFn): identical
    values forward, and the same gradients for the weight, the bias and the input when both outputs carry a gradient."""
    from locate_amd import ops
    from locate_amd.nn import LinearModule
    torch.manual_seed(3)
    lm = LinearModule(24, 40).to(dev())
    lm2 = LinearModule(24, 40).to(dev())
    lm2.load_state_dict(lm.state_dict())
    latent = torch.randn(6, 8, device=dev())
    x1 = torch.randn(6, 24, device=dev(), requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    pre1, nxt1 = lm.pre_and_next_input(x1, latent)
    pre2 = lm2.pre_activation(x2)
    nxt2 = ops.act_cat(latent, pre2)
    assert nxt1.shape == (6, 8 + 40) and torch.equal(nxt1[:, :8], latent)
    assert torch.equal(pre1, pre2)
    assert_close(nxt1.cpu(), nxt2.detach().cpu(), 1e-6, "next input")
    g_pre, g_nxt = torch.randn_like(pre1), torch.randn_like(nxt1)
    torch.autograd.backward([pre1, nxt1], [g_pre, g_nxt])
    torch.autograd.backward([pre2, nxt2], [g_pre, g_nxt])
    assert_close(x1.grad.cpu(), x2.grad.cpu(), 1e-5, "dx")
    for (k, a), (_, b) in zip(lm.named_parameters(), lm2.named_parameters()):
        if a.grad is not None or b.grad is not None:
            assert_close(a.grad.cpu(), b.grad.cpu(), 1e-5, k)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,extra", [((6, 8, 4, 4), 4), ((3, 5, 1, 1), 2), ((2, 3, 3, 5), 7), ((192, 32, 32, 32), 32)])
def test_add3_strided_operands(shape, extra):
    """locate_add3 (ops.Fork3Fn): (a + b) + c with a a channel slice of a wider tensor, against the same two adds in torch -
    bit for bit (both round each add once, in the same order)."""
    from locate_amd._lib import check, lib
    torch.manual_seed(5)
    B, C = shape[0], shape[1]
    wide = torch.randn((B, C + extra) + shape[2:], device=dev())
    a = wide[:, :C]
    b, c = torch.randn(shape, device=dev()), torch.randn(shape, device=dev())
    out = torch.empty(shape, device=dev())
    per = b[0].numel()
    check(lib().locate_add3(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0), out.data_ptr(), B, per,
                            torch.cuda.current_stream().cuda_stream), "locate_add3")
    assert torch.equal(out, (a + b) + c)


@pytest.mark.gpu
def test_fork3_sums_three_gradients_in_one_launch():
    """A discriminator block's input has three consumers (norm of the conv branch, identity half of the concatenation, the skip
    branch's 1x1 conv): with ops.fork3 the three gradients are summed by one kernel; the result equals autograd's own sum of the
    same three gradients up to the order of two roundings."""
    from locate_amd import ops
    from locate_amd.nn import Block
    torch.manual_seed(11)
    blk = Block(16, 8, 16, 2, False, 1).to(dev())          # down-sampling stage, more channels out than in: CatModule + AvgPool2
    import copy
    blk2 = copy.deepcopy(blk)              # every forward advances the power iteration: the second run needs its own u, v
    x1 = torch.randn(4, 8, 16, 16, device=dev(), requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    y1 = blk(x1)
    g = torch.randn_like(y1)
    y1.backward(g)
    real = ops.fork3
    ops.fork3 = lambda t: (t, t, t)          # the same block with autograd's own accumulation
    try:
        y2 = blk2(x2)
        y2.backward(g)
    finally:
        ops.fork3 = real
    assert torch.equal(y1, y2)
    assert_close(x1.grad.cpu(), x2.grad.cpu(), 2e-6, "block input gradient")
    n = 0
    for (k, p), (_, q) in zip(blk.named_parameters(), blk2.named_parameters()):
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is not None:
            n += 1
            assert_close(p.grad.cpu(), q.grad.cpu(), 1e-6, k)
    assert n > 4


@pytest.mark.gpu
def test_small_weight_gradients_batched_equal_single_launches():
    """locate_wgrad_batch: the pass's 1x1-map / one-output-pixel weight gradients in one launch are bit for bit the single
    launches' (same kernel body per layer), including the <G, W_bar> partials and the zeroed taps of the one-pixel layers."""
    import ctypes
    from locate_amd import ops
    from locate_amd._lib import check, lib
    L = lib()
    torch.manual_seed(2)
    st = torch.cuda.current_stream().cuda_stream
    cases = [("conv", 1, 1, 1, 0, 24, 40, 1, 64), ("conv", 1, 1, 1, 0, 832, 384, 1, 64), ("conv", 5, 5, 2, 2, 16, 16, 2, 12),
             ("conv", 3, 3, 1, 1, 32, 8, 1, 6), ("conv", 1, 1, 1, 0, 7, 3, 1, 5), ("conv", 5, 5, 2, 2, 32, 160, 2, 12),
             ("conv", 3, 3, 1, 1, 64, 136, 1, 6), ("conv", 4, 4, 2, 1, 16, 130, 3, 4)]
    recs, expect, outs, keep = [], [], [], []
    for kind, kh, kw, s, p, cin, cout, H, B in cases:
        spec = ops.ConvSpec(kind, kh, kw, s, p, p)
        x = torch.randn(B, cin, H, H, device=dev())
        w = torch.randn(cout, cin, kh, kw, device=dev())
        geom, out_shape = spec.geometry(tuple(x.shape), tuple(w.shape))
        garr = (ctypes.c_int * 12)(*geom)
        gy = torch.randn(out_shape, device=dev())
        inv = torch.full((1,), 0.37, device=dev())
        npart = L.locate_conv_wgrad_partials(garr)
        gw1, gw2 = torch.full_like(w, 7.0), torch.full_like(w, -3.0)
        p1, p2 = torch.zeros(npart, dtype=torch.float64, device=dev()), torch.zeros(npart, dtype=torch.float64, device=dev())
        check(L.locate_conv_wgrad(garr, x.data_ptr(), x.stride(0), gy.data_ptr(), gy.stride(0), gw1.data_ptr(), w.data_ptr(), inv.data_ptr(),
                                  0, 0, p1.data_ptr(), None, 0, None, None, None, st), "locate_conv_wgrad")
        rec = ctypes.create_string_buffer(L.locate_wgrad_batch_record_bytes())
        blocks = L.locate_wgrad_batch_record(garr, x.data_ptr(), x.stride(0), gy.data_ptr(), gy.stride(0), gw2.data_ptr(), w.data_ptr(),
                                             inv.data_ptr(), 0, 0, p2.data_ptr(), rec)
        assert blocks > 0, (kind, kh, cin, cout, H)
        recs.append(rec.raw)
        expect.append((gw1, p1))
        outs.append((gw2, p2))
        keep.append((x, w, gy, inv))
    check(L.locate_wgrad_batch(b"".join(recs), len(recs), st), "locate_wgrad_batch")
    torch.cuda.synchronize()
    for (g1, q1), (g2, q2) in zip(expect, outs):
        assert torch.equal(g1, g2) and torch.equal(q1, q2)
    # a layer the batch does not take: the caller launches it on its own
    spec = ops.ConvSpec("conv", 3, 3, 1, 1, 1, 1)
    geom, _ = spec.geometry((2, 4, 8, 8), (4, 4, 3, 3))
    rec = ctypes.create_string_buffer(L.locate_wgrad_batch_record_bytes())
    t = torch.zeros(4096, device=dev())
    assert L.locate_wgrad_batch_record((ctypes.c_int * 12)(*geom), t.data_ptr(), 256, t.data_ptr(), 256, t.data_ptr(), None, None, 0, 0,
                                       None, rec) == 0


@pytest.mark.gpu
def test_deferred_split_reductions_equal_immediate_ones():
    """locate_conv_wgrad(deferred_reduce) + locate_slab_reduce_batch: the split reductions of several weight gradients in one
    launch at the end are bit for bit the reductions each launch would have run itself (gw and the <G, W_bar> partials)."""
    import ctypes
    from locate_amd import ops
    from locate_amd._lib import check, lib
    L = lib()
    torch.manual_seed(4)
    st = torch.cuda.current_stream().cuda_stream
    cases = [("conv", 5, 2, 2, 32, 32, 16, 24), ("conv", 1, 1, 0, 48, 48, 32, 16), ("conv", 3, 1, 1, 48, 48, 32, 8),
             ("conv", 1, 1, 0, 256, 512, 2, 48), ("conv", 4, 2, 1, 96, 96, 16, 8), ("conv", 3, 1, 1, 8, 8, 4, 2)]
    recs, pending, keep = [], [], []
    for kind, k, s, p, cin, cout, H, B in cases:
        spec = ops.ConvSpec(kind, k, k, s, p, p)
        x = torch.randn(B, cin, H, H, device=dev())
        w = torch.randn(cout, cin, k, k, device=dev())
        geom, out_shape = spec.geometry(tuple(x.shape), tuple(w.shape))
        garr = (ctypes.c_int * 12)(*geom)
        gy = torch.randn(out_shape, device=dev())
        inv = torch.full((1,), 1.7, device=dev())
        npart = L.locate_conv_wgrad_partials(garr)
        nws = max(L.locate_conv_wgrad_workspace_bytes(garr), 16)
        res = []
        for deferred in (False, True):
            gw = torch.full_like(w, 5.0)
            part = torch.zeros(npart, dtype=torch.float64, device=dev())
            ws = torch.empty(nws, dtype=torch.uint8, device=dev())
            rec = ctypes.create_string_buffer(L.locate_slab_reduce_record_bytes()) if deferred else None
            check(L.locate_conv_wgrad(garr, x.data_ptr(), x.stride(0), gy.data_ptr(), gy.stride(0), gw.data_ptr(), w.data_ptr(),
                                      inv.data_ptr(), 0, 0, part.data_ptr(), ws.data_ptr(), 0, None, None, rec, st), "locate_conv_wgrad")
            if deferred and L.locate_slab_reduce_record_blocks(rec) > 0:
                recs.append(rec.raw)
            res.append((gw, part))
            keep.append((ws, x, gy, inv, w))          # the deferred reduction reads the slab and w at the END
        pending.append(res)
    assert len(recs) >= 4              # the last case is a single-block launch without a split: nothing pending for it
    check(L.locate_slab_reduce_batch(b"".join(recs), len(recs), st), "locate_slab_reduce_batch")
    torch.cuda.synchronize()
    for (g1, p1), (g2, p2) in pending:
        assert torch.equal(g1, g2) and torch.equal(p1, p2)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,k,s,p,cin,cout,H,Bg,groups", [("conv", 5, 2, 2, 16, 16, 16, 8, 3), ("conv", 1, 1, 0, 3, 29, 32, 8, 3),
                                                              ("conv", 1, 1, 0, 64, 64, 8, 16, 3), ("conv", 3, 1, 1, 12, 20, 6, 4, 2),
                                                              ("conv", 5, 2, 2, 8, 8, 7, 3, 3), ("convT", 4, 2, 1, 8, 8, 4, 4, 4)])
def test_weight_side_group_dots(kind, k, s, p, cin, cout, H, Bg, groups):
    """Stacked calls: the split reduction of the weight gradient, laid out call by call, emits <G_k / sigma_k, W_bar> per call -
    against float64: the per-call dots equal <gy_k, y_k> of y = conv(x, W_bar) / sigma_k (what locate_fin_sn_dots reads off the
    activations), and gw = sum_k G_k / sigma_k as without them."""
    import ctypes
    from locate_amd import ops
    from locate_amd._lib import check, lib
    L = lib()
    torch.manual_seed(groups * 100 + cin)
    st = torch.cuda.current_stream().cuda_stream
    B = Bg * groups
    spec = ops.ConvSpec(kind, k, k, s, p, p)
    x = torch.randn(B, cin, H, H, device=dev())
    w = torch.randn((cout, cin, k, k) if kind == "conv" else (cin, cout, k, k), device=dev())
    geom, out_shape = spec.geometry(tuple(x.shape), tuple(w.shape))
    garr = (ctypes.c_int * 12)(*geom)
    gy = torch.randn(out_shape, device=dev())
    sig = torch.tensor([1.3, 0.7, 2.1, 0.9][:groups], device=dev())
    inv = (1.0 / sig).contiguous()
    npg = L.locate_conv_wgrad_group_partials(garr, groups)
    assert npg > 0
    part = torch.zeros(groups * npg, dtype=torch.float64, device=dev())
    ws = torch.empty(max(L.locate_conv_wgrad_group_workspace_bytes(garr, groups), 16), dtype=torch.uint8, device=dev())
    gw = torch.empty_like(w)
    xin, gout = (x, gy) if kind == "conv" else (gy, x)
    check(L.locate_conv_wgrad(garr, xin.data_ptr(), xin.stride(0), gout.data_ptr(), gout.stride(0), gw.data_ptr(), w.data_ptr(), inv.data_ptr(),
                              Bg, 1, part.data_ptr(), ws.data_ptr(), 0, None, None, None, st), "locate_conv_wgrad")
    torch.cuda.synchronize()
    xd, wd, gd = x.double().cpu(), w.double().cpu(), gy.double().cpu()
    gw_ref = torch.zeros_like(wd)
    for kk in range(groups):
        sl = slice(kk * Bg, (kk + 1) * Bg)
        wr = wd.clone().requires_grad_(True)
        y = (F.conv2d(xd[sl], wr, None, s, p) if kind == "conv" else F.conv_transpose2d(xd[sl], wr, None, s, p)) / float(sig[kk])
        y.backward(gd[sl])
        gw_ref += wr.grad
        dot_ref = float((gd[sl] * y.detach()).sum())
        got = float(part[kk * npg:(kk + 1) * npg].sum())
        assert abs(got - dot_ref) <= 2e-5 * float((gd[sl] * y.detach()).abs().sum()), (kk, got, dot_ref)
    assert_close(gw.cpu().double(), gw_ref, 2e-5, "gw")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,H", [("conv", 48, 40, 3, 1, 1, 2, 8), ("convT", 40, 72, 4, 2, 1, 2, 4), ("conv", 20, 33, 5, 2, 2, 2, 8),
                                                       ("convT", 3, 5, 4, 2, 1, 2, 4), ("conv", 64, 64, 1, 1, 0, 2, 4), ("conv", 12, 9, 5, 1, 2, 2, 6),
                                                       ("conv", 8, 1, 3, 1, 1, 3, 1)])
@pytest.mark.parametrize("fmt", [2, 0])
def test_direct_repack_equals_two_pass_repack(kind, cin, cout, k, s, p, B, H, fmt):
    """A fp16-piece panel re-packed in its direct form (weights' largest magnitude given: the two planes in one pass, fp32 rows
    only for single-tap panels) serves forward and input gradient exactly like a freshly two-pass-packed panel of the same
    weights - for both directions, full-tap, sub-pixel-phase and single-tap panels, row counts that are no multiple of 8."""
    import ctypes
    from locate_amd import ops
    from locate_amd._lib import check, lib
    L = lib()
    torch.manual_seed(cin * 7 + cout)
    S = lambda: torch.cuda.current_stream().cuda_stream
    spec = ops.ConvSpec(kind, k, k, s, p, p)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w_old = torch.randn(wshape, device=dev())
    w_new = (w_old * 1.7 + 0.3 * torch.randn(wshape, device=dev())).contiguous()
    x = torch.randn(B, cin, H, H, device=dev())
    geom, out_shape = spec.geometry(tuple(x.shape), wshape)
    garr = (ctypes.c_int * 12)(*geom)
    gy = torch.randn(out_shape, device=dev())
    nw = L.locate_absmax_words()
    amax = torch.zeros(3 * nw, dtype=torch.int32, device=dev())
    check(L.locate_absmax(x.data_ptr(), x.numel(), amax[0:].data_ptr(), S()))
    check(L.locate_absmax(gy.data_ptr(), gy.numel(), amax[nw:].data_ptr(), S()))
    check(L.locate_absmax(w_new.data_ptr(), w_new.numel(), amax[2 * nw:].data_ptr(), S()))
    one = torch.ones(1, device=dev())
    res = {}
    seen_direct = []
    for adjoint in (0, 1):
        nbytes = max(L.locate_conv_panel_bytes(garr, adjoint | fmt), 16)
        fresh = torch.zeros(nbytes, dtype=torch.uint8, device=dev())
        check(L.locate_conv_pack_panel(garr, adjoint | fmt, w_new.data_ptr(), fresh.data_ptr(), S()))
        reused = torch.zeros(nbytes, dtype=torch.uint8, device=dev())
        check(L.locate_conv_pack_panel(garr, adjoint | fmt, w_old.data_ptr(), reused.data_ptr(), S()))          # its first, full packing
        job = ctypes.create_string_buffer(L.locate_conv_pack_job_bytes())
        nb = ctypes.c_int(0)
        check(L.locate_conv_pack_job(garr, adjoint | fmt, w_new.data_ptr(), reused.data_ptr(), 0, job, ctypes.byref(nb), 1, amax[2 * nw:].data_ptr() if fmt else None))
        table = torch.frombuffer(job, dtype=torch.uint8).clone().to(dev())
        is_direct = L.locate_conv_pack_job_is_direct(job)          # 0: a geometry the direct bodies do not cover (tap sub-rectangle)
        seen_direct.append(is_direct)
        check(L.locate_conv_pack_panels(table.data_ptr(), 1, nb.value, 0 if is_direct else int(fmt != 0), 0 if is_direct else 1, S()))
        res[adjoint] = (fresh, reused)
    ws = torch.empty(max(L.locate_conv_fwd_workspace_bytes(garr), L.locate_conv_dgrad_workspace_bytes(garr), 16), dtype=torch.uint8, device=dev())
    outs = []
    for which in (0, 1):
        inp, out, pan, ax = (x, torch.empty(out_shape, device=dev()), res[0][which], amax[0:]) if kind == "conv" else \
                            (gy, torch.empty_like(x), res[0][which], amax[nw:])
        check(L.locate_conv_fwd(garr, inp.data_ptr(), inp.stride(0), pan.data_ptr(), one.data_ptr(), 0, 0, None, out.data_ptr(), out.stride(0),
                                ws.data_ptr(), None, 2 if fmt else 0, ax.data_ptr() if fmt else None, None, S()), "locate_conv_fwd")
        inp2, out2, pan2, ax2 = (gy, torch.empty_like(x), res[1][which], amax[nw:]) if kind == "conv" else \
                                (x, torch.empty(out_shape, device=dev()), res[1][which], amax[0:])
        check(L.locate_conv_dgrad(garr, inp2.data_ptr(), inp2.stride(0), pan2.data_ptr(), one.data_ptr(), 0, 0, None, out2.data_ptr(),
                                  out2.stride(0), ws.data_ptr(), None, 2 if fmt else 0, ax2.data_ptr() if fmt else None, None, S()), "locate_conv_dgrad")
        outs.append((out, out2))
    torch.cuda.synchronize()
    assert any(seen_direct)
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(a).all() and a.abs().max() > 0
        assert_close(b.cpu(), a.cpu(), 1e-6, "direct vs two-pass")


@pytest.mark.gpu
def test_end_of_pass_batches_split_into_chunks():
    """More deferred weight gradients in one pass than one batched launch takes (locate_wgrad_batch_max / locate_slab_reduce_max
    records): the runtime launches them in chunks; every gradient equals the un-deferred one bit for bit."""
    import ctypes
    from locate_amd import ops
    from locate_amd._lib import lib
    L = lib()
    torch.manual_seed(9)
    n_small, n_split = L.locate_wgrad_batch_max() + 7, L.locate_slab_reduce_max() + 5
    rt = ops.Runtime()
    rt._end_scheduled = True          # no autograd pass here: the end-of-pass work is run by hand below
    plain = ops.Runtime()
    plain.defer_finalisers = False
    jobs = []
    for i in range(n_small + n_split):
        small = i < n_small
        cin, cout = (5 + i % 7, 3 + i % 5) if small else (8 + i % 4, 8)
        spec = ops.ConvSpec("conv", 1, 1, 1, 0, 0) if small else ops.ConvSpec("conv", 3, 3, 1, 1, 1)
        x = torch.randn(6, cin, 1, 1, device=dev()) if small else torch.randn(4, cin, 16, 16, device=dev())
        w = torch.randn((cout, cin) + ((1, 1) if small else (3, 3)), device=dev())
        geom, out_shape = spec.geometry(tuple(x.shape), tuple(w.shape))
        garr = (ctypes.c_int * 12)(*geom)
        gy = torch.randn(out_shape, device=dev())
        inv = torch.full((1,), 0.5 + 0.01 * i, device=dev())
        npart = ops._weight_grad_partials(spec, geom, garr)
        outs = []
        for r in (rt, plain):
            gw = torch.full_like(w, 3.0)
            part = torch.zeros(npart, dtype=torch.float64, device=dev())
            ops._raw_weight_grad(spec, geom, garr, x, gy, gw, w, inv, 0, 0, part, 0, None, None, r)
            outs.append((gw, part))
        jobs.append(outs)
    assert len(rt._fin_wgrad) == n_small and len(rt._fin_slab) > L.locate_slab_reduce_max()
    rt._end_of_backward()
    torch.cuda.synchronize()
    for (g1, p1), (g2, p2) in jobs:
        assert torch.equal(g1, g2) and torch.equal(p1, p2)


@pytest.mark.gpu
@pytest.mark.parametrize("f16", [True, False])
@pytest.mark.parametrize("kind,cin,cout,k,s,p,H,B", [("convT", 768, 768, 4, 2, 1, 4, 64), ("convT", 96, 96, 4, 2, 1, 32, 64),
                                                      ("conv", 48, 48, 3, 1, 1, 64, 64), ("conv", 128, 128, 5, 2, 2, 8, 192)])
def test_conv_adjointness_at_benchmark_size(kind, cin, cout, k, s, p, H, B, f16):
    """Size-independent property at BASELINE's full sizes, tying the three kernels of a layer together: with y = f(x; W),
    <g, f(x)> = <f^T(g), x> (forward against input gradient) = <dW(x, g), W> (against the weight gradient) - f is bilinear in
    (x, W).  In both fp32-faithful forms (fp16 pieces when the operands carry their largest magnitude, bf16 pieces otherwise);
    sums in float64, bound 2e-6 of sum |g f(x)| (the forms' own error is a few 1e-7 per element, uncorrelated)."""
    from locate_amd import ops
    torch.manual_seed(cin + k + H)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = (torch.randn(wshape, device=dev()) * 0.05).requires_grad_(True)
    rows = wshape[0]
    u, v = torch.randn(rows, device=dev()), torch.randn(w.numel() // rows, device=dev())
    sigma, wv = torch.tensor([1.0, 1.0], device=dev()), torch.zeros(rows, device=dev())      # sigma = 1: f is the bare layer
    spec = ops.ConvSpec(kind, k, k, s, p, p)
    x = torch.randn(B, cin, H, H, device=dev())
    if f16:
        ops.tag_amax(x)
    x.requires_grad_(True)
    before = dict(ops.F16_CALLS)
    y = ops.SNConvFn.apply(x, w, u, v, None, sigma, wv, spec)
    g = torch.randn_like(y)
    if f16:
        ops.tag_amax(g)
    y.backward(g)
    took = {kk: ops.F16_CALLS[kk] - before[kk] for kk in before}
    assert took == ({"fwd": 1, "dgrad": 1, "wgrad": 1} if f16 else {"fwd": 0, "dgrad": 0, "wgrad": 0}), took
    lhs = float((g.double() * y.detach().double()).sum())
    scale = float((g.double() * y.detach().double()).abs().sum())
    via_x = float((x.grad.double() * x.detach().double()).sum())
    # the weight gradient carries the spectral-norm correction d(sigma) u v^T with d(sigma) = -<G, W> / sigma^2; with sigma = 1:
    # <dW_total, W> = <G, W> - <G, W> <u v^T, W>, so <G, W> = <dW_total, W> / (1 - <u v^T, W>)
    uvw = float((torch.outer(u, v).double() * w.detach().double().view(rows, -1)).sum())
    via_w = float((w.grad.double() * w.detach().double()).sum()) / (1.0 - uvw)
    assert abs(via_x - lhs) <= 2e-6 * scale, (lhs, via_x, scale)
    assert abs(via_w - lhs) <= 2e-6 * scale * max(1.0, abs(1.0 / (1.0 - uvw))), (lhs, via_w, scale, uvw)


@pytest.mark.gpu
def test_linear_stencils_are_adjoint_pairs_at_benchmark_size():
    """Size-independent property at config 2's own tensor sizes (batch 64): every linear resampling operator and its backward
    kernel are adjoint - <op(x), y> = <x, op^T(y)> - for the bilinear x2 upsample (all three kernel variants' sizes), the 2x2
    mean and the channel pooling of the skip branch; sums in float64."""
    from locate_amd import ops
    torch.manual_seed(21)
    cases = [("upsample 96@32", lambda t: ops.upsample2x(t), (64, 96, 32, 32)), ("upsample 768@4", lambda t: ops.upsample2x(t), (64, 768, 4, 4)),
             ("upsample 384@8", lambda t: ops.upsample2x(t), (64, 384, 8, 8)), ("avgpool 32@64 [3B]", lambda t: ops.avgpool2(t), (192, 32, 64, 64)),
             ("avgpool 512@2", lambda t: ops.avgpool2(t), (192, 512, 2, 2)), ("feature pool 96->48@64", lambda t: ops.feature_pool(t, 48), (64, 96, 64, 64)),
             ("feature pool 768->384@8", lambda t: ops.feature_pool(t, 384), (64, 768, 8, 8))]
    for name, op, shape in cases:
        x = torch.randn(shape, device=dev(), requires_grad=True)
        y = op(x)
        g = torch.randn_like(y)
        y.backward(g)
        lhs = float((y.detach().double() * g.double()).sum())
        rhs = float((x.grad.double() * x.detach().double()).sum())
        scale = float((y.detach().double() * g.double()).abs().sum())
        assert abs(lhs - rhs) <= 1e-6 * scale, (name, lhs, rhs, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("with_act", [False, True])
def test_inplace_norm_gradient_annihilates_shift_and_scale_at_benchmark_size(with_act):
    """Size-independent property at config 2's largest norm input (64 x 96 x 64 x 64): out = norm(x) is invariant under x -> a x + b
    (a > 0) for its GLOBAL statistics, so the input gradient is orthogonal to the all-ones tensor and to x itself - whatever the
    scale, the bias, the following RootTanh and the incoming gradient are.  Float64 sums; bound 2e-6 of sum |gx x|."""
    from locate_amd import ops
    torch.manual_seed(33)
    x = (torch.randn(64, 96, 64, 64, device=dev()) * 1.7 + 0.4).requires_grad_(True)
    scale = torch.randn(1, 96, 1, 1, device=dev()) * 0.5 + 1.0
    bias = torch.randn(1, 96, 1, 1, device=dev()) * 0.3
    y = ops.inplace_norm(x, scale, bias, with_act=with_act)
    g = torch.randn_like(y)
    y.backward(g)
    gx, xd = x.grad.double(), x.detach().double()
    ref = float((gx * xd).abs().sum())
    assert abs(float(gx.sum())) <= 2e-6 * float(gx.abs().sum())
    assert abs(float((gx * xd).sum())) <= 2e-6 * ref


@pytest.mark.gpu
def test_gate_and_softmax_properties_at_benchmark_size():
    """Size-independent properties at config 2's own sizes.  The residual gate out = (gamma a + 1) x is linear in x and affine in
    a: <out, g> = <x, dx> and <out - x, g> = <a, da> (its a-dependent part is linear in a).  The position softmax (N = 4096,
    64 x 48 rows): every row sums to 1 and its input gradient sums to 0 over each row."""
    from locate_amd import ops
    torch.manual_seed(41)
    x = torch.randn(64, 48, 64, 64, device=dev(), requires_grad=True)
    a = torch.randn(64, 48, 64, 64, device=dev(), requires_grad=True)
    gamma = torch.full((1, 1), 2.5, device=dev(), requires_grad=True)
    out = ops.residual_gate(x, a, gamma, with_stats=False)
    g = torch.randn_like(out)
    out.backward(g)
    od, gd, xd, ad = out.detach().double(), g.double(), x.detach().double(), a.detach().double()
    scale = float((od * gd).abs().sum())
    assert abs(float((od * gd).sum()) - float((x.grad.double() * xd).sum())) <= 1e-6 * scale
    assert abs(float(((od - xd) * gd).sum()) - float((a.grad.double() * ad).sum())) <= 1e-6 * scale
    rows = torch.randn(64, 48, 4096, device=dev(), requires_grad=True)
    p = ops.softmax_lastdim(rows)
    gp = torch.randn_like(p)
    p.backward(gp)
    assert float((p.detach().double().sum(-1) - 1.0).abs().max()) <= 1e-6
    assert float(rows.grad.double().sum(-1).abs().max()) <= 1e-6 * float(rows.grad.double().abs().sum(-1).max())
