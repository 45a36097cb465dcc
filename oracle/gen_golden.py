"""Golden-fixture generator: runs the REAL reference (imported from /root/reference, container only)
on small seeded inputs and writes input/expected-output vectors to tests/golden/*.npz.

TEST INFRASTRUCTURE.  Run here, once, by hand:   python oracle/gen_golden.py all
The fixtures are data (inputs + expected outputs); no reference source text is stored.
Groups follow SURVEY.md section 8(c):  G1 roottanh, G2 inplace_norm, G3 residual, G4 spectral_norm,
G5 indexing/resample, G6 attention, G7 blocks, G8 tiny end-to-end (2 full steps), G9 nadam,
G10 seeded-construction checksums, G11 config-1 (32x32 bs 8, full width) losses and norms.

Every group runs in its own subprocess because the reference binds its configuration constants at
import time (one process = one configuration).
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, d):
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote %s  (%d arrays, %.1f KiB)" % (path, len(d), os.path.getsize(path) / 1024))


def _sd(prefix, module, out):
    for k, v in module.state_dict().items():
        out[prefix + k] = _np(v)


def _grads(prefix, module, out):
    for k, p in module.named_parameters():
        if p.grad is not None:
            out[prefix + k] = _np(p.grad)


# --------------------------------------------------------------------------------------------
def gen_ops():
    """G1-G6 + G9: operator-level vectors (config-independent; tiny config loaded)."""
    import torch
    from ref_loader import load_reference
    ns = load_reference(32, 1)
    torch.manual_seed(1234)

    # ---- G1 RootTanh (libs/activation.py:7-36)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        special = torch.tensor([0.0, 1e-8, -1e-8, 0.5, -0.5, 1.0, -1.0, 3.0, -3.0, 10.0, -10.0, 44.0, 46.0, 50.0,
                                -50.0, 100.0, -100.0, 1e4, -1e4], dtype=dt)
        x = torch.cat([special, torch.randn(237, dtype=dt) * 3]).requires_grad_(True)
        y = ns.activation.nonlinear_function(x)
        g = torch.randn_like(y)
        y.backward(g)
        d = dict(x=_np(x), y=_np(y), g=_np(g), dx=_np(x.grad))
        _save("g1_roottanh_" + tag, d)

    # ---- G2 InPlaceNorm (libs/inplace_norm.py:4-45)
    d = {}
    for case, (B, C, H, W) in (("a", (3, 5, 4, 6)), ("b", (2, 4, 1, 1)), ("c", (4, 8, 8, 8))):
        norm = ns.inplace_norm.InPlaceNorm(C)
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5)
            norm.bias.normal_()
        x = (torch.randn(B, C, H, W) * 2 + 0.7).requires_grad_(True)
        g = torch.randn(B, C, H, W)
        out = norm(x)
        out.backward(g)
        d.update({case + "_w_x": _np(x), case + "_w_weight": _np(norm.weight), case + "_w_bias": _np(norm.bias),
                  case + "_w_g": _np(g), case + "_w_out": _np(out), case + "_w_dx": _np(x.grad),
                  case + "_w_dweight": _np(norm.weight.grad), case + "_w_dbias": _np(norm.bias.grad),
                  case + "_w_mean": _np(x.mean()), case + "_w_std": _np(x.std())})
        # style-scale variant: scale [B,C,1,1] replaces the weight (block.py:125, inplace_norm.py:44-45)
        norm.zero_grad()
        x2 = (torch.randn(B, C, H, W) - 0.3).requires_grad_(True)
        scale = (torch.randn(B, C, 1, 1)).requires_grad_(True)
        out = norm(x2, scale)
        out.backward(g)
        assert norm.weight.grad is None
        d.update({case + "_s_x": _np(x2), case + "_s_scale": _np(scale), case + "_s_bias": _np(norm.bias),
                  case + "_s_g": _np(g), case + "_s_out": _np(out), case + "_s_dx": _np(x2.grad),
                  case + "_s_dscale": _np(scale.grad), case + "_s_dbias": _np(norm.bias.grad)})
    _save("g2_inplace_norm", d)

    # ---- G3 ResidualFunction / ResModule (libs/merge.py:19-62)
    d = {}
    x = torch.randn(2, 3, 4, 5, requires_grad=True)
    a = torch.randn(2, 3, 4, 5, requires_grad=True)
    gamma = torch.tensor([[2.5]], requires_grad=True)
    gexp = gamma.view(1, 1, 1, 1).expand_as(a)
    out = ns.merge.residual_function(x, a, gexp)
    g = torch.randn_like(out)
    out.backward(g)
    d.update(full_x=_np(x), full_a=_np(a), full_gamma=_np(gamma), full_g=_np(g), full_out=_np(out),
             full_dx=_np(x.grad), full_da=_np(a.grad), full_dgamma=_np(gamma.grad))
    # attention given as a stride-0 expand of [B,C,1,1] (feature attention path)
    x = torch.randn(2, 3, 4, 4, requires_grad=True)
    a0 = torch.randn(2, 3, 1, 1, requires_grad=True)
    gamma = torch.tensor([[-0.75]], requires_grad=True)
    out = ns.merge.residual_function(x, a0.expand(2, 3, 4, 4), gamma.view(1, 1, 1, 1).expand(2, 3, 4, 4))
    g = torch.randn_like(out)
    out.backward(g)
    d.update(bc_x=_np(x), bc_a=_np(a0), bc_gamma=_np(gamma), bc_g=_np(g), bc_out=_np(out),
             bc_dx=_np(x.grad), bc_da=_np(a0.grad), bc_dgamma=_np(gamma.grad))
    # gamma initial values of ResModule(m) for a fixed seed (merge.py:51-53)
    torch.manual_seed(7)
    d["resmodule_gammas_m0_m1_m3"] = np.array(
        [ns.merge.ResModule(lambda t: t, lambda t: t, m=m).gamma.item() for m in (0, 1, 3, 0, 3, 1)], np.float32)
    _save("g3_residual", d)

    # ---- G4 SpectralNorm on Conv2d / ConvTranspose2d / Conv1d / Linear (libs/spectral_norm.py)
    torch.manual_seed(4321)
    d = {}
    SN = ns.spectral_norm.SpectralNorm
    cases = {
        "conv5s2": (torch.nn.Conv2d(4, 6, 5, stride=2, padding=2, bias=False), (3, 4, 8, 8)),
        "conv3": (torch.nn.Conv2d(5, 3, 3, stride=1, padding=1, bias=False), (2, 5, 6, 6)),
        "conv1x1b": (torch.nn.Conv2d(4, 7, 1), (2, 4, 5, 5)),
        "convT4s2": (torch.nn.ConvTranspose2d(6, 6, 4, stride=2, padding=1, bias=False), (2, 6, 4, 4)),
        "convT1x1": (torch.nn.ConvTranspose2d(6, 3, 1, bias=False), (2, 6, 8, 8)),
        "conv1d": (torch.nn.Conv1d(8, 8, 1, bias=False), (2, 8, 16)),
        "convS1": (torch.nn.Conv2d(8, 2, (4, 1), bias=False), (3, 8, 4, 4)),
        "conv1S": (torch.nn.Conv2d(2, 2, (1, 4), bias=False), (3, 2, 1, 4)),
        "linear": (torch.nn.Linear(10, 6), (5, 10)),
    }
    for name, (inner, xshape) in cases.items():
        for uv_grad in (False, True):
            mod = SN(inner) if not uv_grad else mod  # reuse the same wrapped module for the 2nd phase
            tag = name + ("_uvg" if uv_grad else "")
            if uv_grad:
                mod.zero_grad()
                mod.requires_grad_(True)  # what main.py:172 does to every D parameter, u/v included
            _sd(tag + "/sd0/", mod, d)
            x = torch.randn(*xshape, requires_grad=True)
            d[tag + "/x"] = _np(x)
            outs, gs = [], []
            for k in range(3):  # three forwards before one backward, like the D-step (main.py:149-156)
                y = mod(x)
                outs.append(y)
                d[tag + "/y%d" % k] = _np(y)
                d[tag + "/u%d" % k] = _np(mod.module.weight_u)
                d[tag + "/v%d" % k] = _np(mod.module.weight_v)
                gk = torch.randn_like(y)
                gs.append(gk)
                d[tag + "/g%d" % k] = _np(gk)
            sum((o * gg).sum() for o, gg in zip(outs, gs)).backward()
            d[tag + "/dx"] = _np(x.grad)
            _grads(tag + "/grad/", mod, d)
    _save("g4_spectral_norm", d)

    # ---- G5 indexing / resampling (libs/scale.py, libs/util_modules.py, libs/merge.py:4-16)
    d = {}
    x = torch.randn(2, 8, 3, 4, requires_grad=True)
    for r_out in (4, 2):
        fp = ns.scale.FeaturePooling(r_out)
        y = fp(x)
        g = torch.randn_like(y)
        x.grad = None
        y.backward(g)
        d.update({"fpool%d_x" % r_out: _np(x), "fpool%d_y" % r_out: _np(y), "fpool%d_g" % r_out: _np(g),
                  "fpool%d_dx" % r_out: _np(x.grad)})
    t = torch.randn(3, 5, 1, 1, requires_grad=True)
    e = ns.util_modules.Expand(-1, 5, 4, 4)(t)
    g = torch.randn(3, 5, 4, 4)
    e.backward(g)
    d.update(expand_x=_np(t), expand_y=_np(e), expand_g=_np(g), expand_dx=_np(t.grad))
    x = torch.randn(2, 3, 5, 6, requires_grad=True)
    up = torch.nn.Upsample(mode="bilinear", scale_factor=2, align_corners=False)
    y = up(x)
    g = torch.randn_like(y)
    y.backward(g)
    d.update(up_x=_np(x), up_y=_np(y), up_g=_np(g), up_dx=_np(x.grad))
    x = torch.randn(2, 3, 6, 8, requires_grad=True)
    y = torch.nn.AvgPool2d(2, 2)(x)
    g = torch.randn_like(y)
    y.backward(g)
    d.update(pool_x=_np(x), pool_y=_np(y), pool_g=_np(g), pool_dx=_np(x.grad))
    # Scale() compositions as built by Block (scale.py:19-45)
    torch.manual_seed(99)
    for name, args, xshape in (("scale_up_pool", (8, 4, 2, True), (2, 8, 3, 3)),
                               ("scale_up_cat", (4, 12, 2, True), (2, 4, 2, 2)),
                               ("scale_down_cat", (4, 8, 2, False), (2, 4, 6, 6)),
                               ("scale_down_same", (6, 6, 2, False), (2, 6, 4, 4))):
        layer = ns.scale.Scale(*args)
        x = torch.randn(*xshape, requires_grad=True)
        if isinstance(layer, torch.nn.Module):
            _sd(name + "/sd0/", layer, d)
        y = layer(x)
        g = torch.randn_like(y)
        y.backward(g)
        d.update({name + "/x": _np(x), name + "/y": _np(y), name + "/g": _np(g), name + "/dx": _np(x.grad)})
        if isinstance(layer, torch.nn.Module):
            _sd(name + "/sd1/", layer, d)
            _grads(name + "/grad/", layer, d)
    _save("g5_indexing", d)

    # ---- G6 attention layers (libs/attention.py)
    torch.manual_seed(606)
    d = {}
    fa = ns.attention.feature_attention(8, 16)
    sa = ns.attention.SelfAttention(16)
    for name, mod, xshape in (("fa", fa, (3, 16, 8, 8)), ("sa", sa, (3, 16, 8, 8))):
        _sd(name + "/sd0/", mod, d)
        x = torch.randn(*xshape, requires_grad=True)
        y = mod(x)
        g = torch.randn(*y.shape)
        y.backward(g)
        d.update({name + "/x": _np(x), name + "/y": _np(y), name + "/g": _np(g), name + "/dx": _np(x.grad)})
        _sd(name + "/sd1/", mod, d)
        _grads(name + "/grad/", mod, d)
    # LinearModule (libs/linear.py)
    lm = ns.linear.LinearModule(12, 7)
    _sd("lin/sd0/", lm, d)
    x = torch.randn(4, 12, requires_grad=True)
    act, pre = lm(x)
    g1, g2 = torch.randn_like(act), torch.randn_like(pre)
    ((act * g1).sum() + (pre * g2).sum()).backward()
    d.update({"lin/x": _np(x), "lin/act": _np(act), "lin/pre": _np(pre), "lin/g_act": _np(g1), "lin/g_pre": _np(g2),
              "lin/dx": _np(x.grad)})
    _sd("lin/sd1/", lm, d)
    _grads("lin/grad/", lm, d)
    _save("g6_attention", d)

    # ---- G9 Nadam, 3 steps (libs/nadam.py:31-89, hyper-parameters config.py:70-73)
    import warnings
    torch.manual_seed(909)
    d = {}
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)),
          torch.nn.Parameter(torch.randn(1, 1))]
    opt = ns.nadam.Nadam(ps, lr=ns.config.DLR, betas=(ns.config.BETA_1, ns.config.BETA_2))
    for i, p in enumerate(ps):
        d["p%d_0" % i] = _np(p)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(1, 4):
            for i, p in enumerate(ps):
                # parameter 2 has no gradient on the first step (like D's u/v)
                p.grad = None if (i == 2 and step == 1) else torch.randn_like(p)
                if p.grad is not None:
                    d["g%d_%d" % (i, step)] = _np(p.grad)
            opt.step()
            for i, p in enumerate(ps):
                d["p%d_%d" % (i, step)] = _np(p)
    d["lr"] = np.float64(ns.config.DLR)
    d["betas"] = np.array([ns.config.BETA_1, ns.config.BETA_2], np.float64)
    _save("g9_nadam", d)


# --------------------------------------------------------------------------------------------
def _build_models(ns, seed=999):
    import torch
    from ref_loader import quiet_stdout
    torch.manual_seed(seed)
    with quiet_stdout():
        G, GO = ns.utils.get_model(ns.models.Generator(), ns.config.GLR, ns.config.DEVICE)
        D, DO = ns.utils.get_model(ns.models.Discriminator(), ns.config.DLR, ns.config.DEVICE)
    return G, GO, D, DO


def _train_step(ns, G, GO, D, DO, latent, real, aug):
    """Loop body of main.py:142-172 with miniter=1, MINIBATCHES=1, DITERS=1, on given tensors."""
    hinge, penalty = ns.utils.hinge, ns.grad_penalty.penalty
    rec = {}
    generated = G(latent).detach()
    rec["generated"] = generated
    D.zero_grad()
    d_true = D(real).view(-1)
    d_gen = -D(generated).view(-1)
    d_error = (hinge(d_true) + hinge(d_gen)).mean()
    pen = penalty(d_true, aug, D, ns.config.DEVICE)
    (d_error + pen).backward()
    rec.update(d_true=d_true.detach(), d_gen=d_gen.detach(), d_error=d_error.detach(), penalty=pen.detach())
    rec["d_grads"] = {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None}
    rec["d_sd_pre_step"] = {k: v.clone() for k, v in D.state_dict().items()}
    DO.step()
    rec["d_sd_post_step"] = {k: v.clone() for k, v in D.state_dict().items()}
    D.requires_grad_(False)
    G.zero_grad()
    fake = G(latent)
    g_error = hinge(D(fake).view(-1)).mean()
    g_error.backward()
    rec["g_error"] = g_error.detach()
    rec["fake"] = fake.detach()
    rec["g_grads"] = {k: p.grad.clone() for k, p in G.named_parameters() if p.grad is not None}
    rec["g_none_grads"] = [k for k, p in G.named_parameters() if p.grad is None and p.requires_grad]
    GO.step()
    D.requires_grad_(True)
    rec["g_sd_post_step"] = {k: v.clone() for k, v in G.state_dict().items()}
    rec["d_sd_end"] = {k: v.clone() for k, v in D.state_dict().items()}
    return rec


def gen_tiny():
    """G7 + G8 + G10(tiny): IMAGE_SIZE=32, BASE_FEATURE_FACTOR=1 (attention in both nets)."""
    import warnings
    import torch
    from ref_loader import load_reference, quiet_stdout
    ns = load_reference(32, 1)
    warnings.simplefilter("ignore")

    # ---- G7 single blocks (libs/block.py:15-52)
    torch.manual_seed(707)
    d = {}
    with quiet_stdout():
        up = ns.block.Block(8, 16, 8, 2, True, 0)      # transposed, attention (in_size>=8, idx%2==0)
        up_na = ns.block.Block(4, 8, 8, 2, True, 1)    # transposed, no attention, FeaturePooling absent (8->8)
        down = ns.block.Block(8, 8, 16, 2, False, 0)   # strided, attention, cat skip
        down_na = ns.block.Block(4, 16, 16, 2, False, 1)
    B = 3
    for name, blk, xshape, feats in (("up", up, (B, 16, 4, 4), (16, 8)), ("up_na", up_na, (B, 8, 2, 2), (8, 8)),
                                     ("down", down, (B, 8, 16, 16), (8, 16)),
                                     ("down_na", down_na, (B, 16, 8, 8), (16, 16))):
        blk.apply(ns.utils.init)
        _sd(name + "/sd0/", blk, d)
        x = torch.randn(*xshape, requires_grad=True)
        if name.startswith("up"):
            n_scales = 3 if blk.attention else 1
            chans = [feats[0]] + [feats[1]] * (n_scales - 1)
            scales = [torch.randn(B, c, 1, 1, requires_grad=True) for c in chans]
        else:
            scales = None
        y = blk(x, scales)
        g = torch.randn_like(y)
        y.backward(g)
        d.update({name + "/x": _np(x), name + "/y": _np(y), name + "/g": _np(g), name + "/dx": _np(x.grad)})
        if scales is not None:
            for i, s in enumerate(scales):
                d[name + "/scale%d" % i] = _np(s)
                d[name + "/dscale%d" % i] = _np(s.grad)
        _sd(name + "/sd1/", blk, d)
        _grads(name + "/grad/", blk, d)
    _save("g7_blocks", d)

    # ---- G8 tiny end-to-end, two full steps
    G, GO, D, DO = _build_models(ns, 999)
    d = {}
    _sd("G/sd0/", G, d)
    _sd("D/sd0/", D, d)
    d["G/noise"] = _np(G.noise)
    d["meta/g_in"] = np.int64(G.g_in)
    d["meta/g_param_count"] = np.int64(ns.utils.parameter_count(G))
    d["meta/d_param_count"] = np.int64(ns.utils.parameter_count(D))
    B, S = 8, 32
    for step in (1, 2):
        latent = torch.randn(B, S)
        real = torch.randn(B, 3, S, S).clamp(-1, 1)
        aug = torch.randn(B, 3, S, S).clamp(-1, 1)
        rec = _train_step(ns, G, GO, D, DO, latent, real, aug)
        p = "step%d/" % step
        d[p + "latent"], d[p + "real"], d[p + "aug"] = _np(latent), _np(real), _np(aug)
        for k in ("generated", "fake", "d_true", "d_gen", "d_error", "penalty", "g_error"):
            d[p + k] = _np(rec[k])
        for k, v in rec["d_grads"].items():
            d[p + "D/grad/" + k] = _np(v)
        for k, v in rec["g_grads"].items():
            d[p + "G/grad/" + k] = _np(v)
        d[p + "G/none_grads"] = np.array(rec["g_none_grads"])
        for k, v in rec["d_sd_pre_step"].items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                d[p + "D/sd_pre_step/" + k] = _np(v)       # u/v after the three D forwards
        for k, v in rec["d_sd_post_step"].items():
            d[p + "D/sd_post_step/" + k] = _np(v)
        for k, v in rec["g_sd_post_step"].items():
            d[p + "G/sd_post_step/" + k] = _np(v)
        for k, v in rec["d_sd_end"].items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                d[p + "D/sd_end/" + k] = _np(v)             # u/v after the G-step's D forward
    d["meta/d_param_count_after"] = np.int64(ns.utils.parameter_count(D))
    # eval-mode sampling still advances u/v (main.py:195-202)
    G.eval()
    with torch.no_grad():
        fixed = torch.randn(4, S)
        d["sample/latent"] = _np(fixed)
        d["sample/image"] = _np(G(fixed))
    _save("g8_tiny_e2e", d)


def gen_init():
    """G10: seeded-construction checksums at 32 (tiny), 64 and 128 full width (state_dict per-tensor stats)."""
    import torch
    cfgs = {"tiny32": (32, 1), "full32": (32, 8), "full64": (64, 8)}
    which = os.environ["LOCATE_GOLDEN_INIT_CFG"]
    size, bff = cfgs[which]
    from ref_loader import load_reference
    ns = load_reference(size, bff)
    G, GO, D, DO = _build_models(ns, 999)
    d = {}
    for tag, net in (("G", G), ("D", D)):
        keys = list(net.state_dict().keys())
        d[tag + "/keys"] = np.array(keys)
        d[tag + "/shapes"] = np.array([str(tuple(v.shape)) for v in net.state_dict().values()])
        d[tag + "/sum"] = np.array([float(v.double().sum()) for v in net.state_dict().values()])
        d[tag + "/abssum"] = np.array([float(v.double().abs().sum()) for v in net.state_dict().values()])
        d[tag + "/first"] = np.array([float(v.flatten()[0]) for v in net.state_dict().values()])
        d[tag + "/requires_grad"] = np.array([bool(p.requires_grad) for _, p in net.named_parameters()])
    d["G/noise_sum"] = np.float64(G.noise.double().sum())
    d["G/noise_first"] = np.float64(G.noise.flatten()[0])
    d["G/g_in"] = np.int64(G.g_in)
    d["G/param_count"] = np.int64(ns.utils.parameter_count(G))
    d["D/param_count"] = np.int64(ns.utils.parameter_count(D))
    d["after_rng"] = _np(torch.randn(4))  # RNG position after construction
    _save("g10_init_" + which, d)


def gen_config1():
    """G11: config 1 (32x32 RGB bs 8, full width): losses and per-tensor norms of one step."""
    _gen_step_record("g11_config1", 32, 8, 8)


def gen_config3():
    """G12: the reference's default architecture (128x128, full width; BASELINE.json configs[3]) at batch 2."""
    _gen_step_record("g12_config3_128", 128, 2, 8)


def gen_config2():
    """G14: BASELINE.json configs[1], the benchmark workload itself (64x64, batch 64, full width)."""
    _gen_step_record("g14_config2_64", 64, 64, 8)


def gen_size256():
    """G13: the 256x256 architecture (BASELINE.json configs[4]: attention up to N = 65 536) at 1/8 width, batch 2."""
    _gen_step_record("g13_256_narrow", 256, 2, 1)


def gen_size256_full():
    """G20: BASELINE.json configs[4]'s architecture AS NAMED - 256x256 at FULL width (G 225.9 M / D 186.5 M parameters,
    ConvTranspose [3072, 3072, 4, 4], self-attention over N = 65 536 positions at C = 48) - at batch 2, fp32."""
    _gen_step_record("g20_256_full", 256, 2, 8)


def gen_variants():
    """G15-G19: the architecture switches of libs/config.py that the shipped defaults leave off:
    DEPTH > 1 (conv.py:61-67), FEATURE_MULTIPLIER > 1 (conv.py:16,21), SEPARABLE (conv.py:17, attention.py:15-21)."""
    which = os.environ["LOCATE_GOLDEN_VARIANT"]
    name, S, B, ff, consts = VARIANTS[which]
    _gen_step_record(name, S, B, ff, **consts)


VARIANTS = {
    "depth2": ("g15_depth2_32", 32, 4, 2, dict(DEPTH=2)),
    "depth3_fm2": ("g16_depth3_fm2_32", 32, 4, 2, dict(DEPTH=3, FEATURE_MULTIPLIER=2)),
    "separable": ("g17_separable_32", 32, 4, 4, dict(SEPARABLE=True)),
    "separable_all": ("g18_separable_depth2_fm2_64", 64, 2, 2, dict(SEPARABLE=True, DEPTH=2, FEATURE_MULTIPLIER=2)),
    "separable128": ("g19_separable_128", 128, 2, 8, dict(SEPARABLE=True)),
}


def _gen_step_record(name, S, B, ff, **consts):
    import warnings
    import torch
    from ref_loader import load_reference
    ns = load_reference(S, ff, **consts)
    warnings.simplefilter("ignore")
    G, GO, D, DO = _build_models(ns, 999)
    latent = torch.randn(B, S)
    real = torch.randn(B, 3, S, S).clamp(-1, 1)
    aug = torch.randn(B, 3, S, S).clamp(-1, 1)
    rec = _train_step(ns, G, GO, D, DO, latent, real, aug)
    d = {"after_build_rng_check": _np(latent[0, :4])}
    for k in ("d_true", "d_gen", "d_error", "penalty", "g_error"):
        d[k] = _np(rec[k])
    d["fake_norm"] = np.float64(rec["fake"].double().norm())
    d["fake_first"] = _np(rec["fake"].flatten()[:16])
    for tag, gr in (("D", rec["d_grads"]), ("G", rec["g_grads"])):
        d[tag + "/grad_keys"] = np.array(list(gr.keys()))
        d[tag + "/grad_norms"] = np.array([float(v.double().norm()) for v in gr.values()])
    for tag, sd in (("D", rec["d_sd_post_step"]), ("G", rec["g_sd_post_step"])):
        d[tag + "/post_keys"] = np.array(list(sd.keys()))
        d[tag + "/post_norms"] = np.array([float(v.double().norm()) for v in sd.values()])
    _save(name, d)


def gen_branches():
    """G21: the two branches of the reference the shipped training loop never takes ("small refusals"):
    Nadam with weight_decay != 0 (libs/nadam.py:65-66) and SpectralNorm(power_iterations > 1) (libs/spectral_norm.py:26-29)."""
    import warnings
    import torch
    from ref_loader import load_reference
    ns = load_reference(32, 1)
    warnings.simplefilter("ignore")
    torch.manual_seed(2121)
    d = {}
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    opt = ns.nadam.Nadam(ps, lr=ns.config.DLR, betas=(ns.config.BETA_1, ns.config.BETA_2), weight_decay=0.05)
    for i, p in enumerate(ps):
        d["wd/p%d_0" % i] = _np(p)
    for step in range(1, 4):
        for i, p in enumerate(ps):
            p.grad = torch.randn_like(p)
            d["wd/g%d_%d" % (i, step)] = _np(p.grad)
        opt.step()
        for i, p in enumerate(ps):
            d["wd/p%d_%d" % (i, step)] = _np(p)
    d["wd/lr"] = np.float64(ns.config.DLR)
    d["wd/betas"] = np.array([ns.config.BETA_1, ns.config.BETA_2], np.float64)
    d["wd/weight_decay"] = np.float64(0.05)
    SN = ns.spectral_norm.SpectralNorm
    for name, inner, xshape, iters in (("conv3", torch.nn.Conv2d(5, 3, 3, stride=1, padding=1, bias=False), (2, 5, 6, 6), 3),
                                        ("convT4s2", torch.nn.ConvTranspose2d(6, 6, 4, stride=2, padding=1, bias=False), (2, 6, 4, 4), 2)):
        mod = SN(inner, power_iterations=iters)
        mod.requires_grad_(True)
        tag = "pi/" + name
        d[tag + "/iters"] = np.int64(iters)
        _sd(tag + "/sd0/", mod, d)
        x = torch.randn(*xshape, requires_grad=True)
        d[tag + "/x"] = _np(x)
        outs, gs = [], []
        for k in range(2):
            y = mod(x)
            outs.append(y)
            d[tag + "/y%d" % k] = _np(y)
            d[tag + "/u%d" % k] = _np(mod.module.weight_u)
            d[tag + "/v%d" % k] = _np(mod.module.weight_v)
            gk = torch.randn_like(y)
            gs.append(gk)
            d[tag + "/g%d" % k] = _np(gk)
        sum((o * gg).sum() for o, gg in zip(outs, gs)).backward()
        d[tag + "/dx"] = _np(x.grad)
        _grads(tag + "/grad/", mod, d)
    _save("g21_branches", d)


def gen_f64():
    """The reference's own fp32 rounding noise on the ill-conditioned scalars of a step, per configuration: the same seeded
    build and the same inputs run twice through the REFERENCE, once as shipped (fp32) and once with both networks converted
    to float64 (`module.double()`, `G.noise.double()`, double inputs).  Stored: every gate's d(gamma) (merge.py:33-38, a sum
    with heavy cancellation) from both runs, the losses and the consistency penalty (grad_penalty.py:1-2) from both runs,
    the gradient norms and the post-optimizer-step parameter norms of every tensor from both runs.  The GPU tests then hold the kernels to a multiple of
    |ref32 - ref64| - the deviation the reference shows against itself - instead of to a bound derived from the build."""
    import warnings
    import torch
    from ref_loader import load_reference
    which = os.environ["LOCATE_GOLDEN_F64"]
    name, S, B, ff = F64_RECORDS[which]
    ns = load_reference(S, ff)
    warnings.simplefilter("ignore")
    runs = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        G, GO, D, DO = _build_models(ns, 999)
        latent = torch.randn(B, S)
        real = torch.randn(B, 3, S, S).clamp(-1, 1)
        aug = torch.randn(B, 3, S, S).clamp(-1, 1)
        if dt is torch.float64:
            G, D = G.double(), D.double()
            G.noise = G.noise.double()
            latent, real, aug = latent.double(), real.double(), aug.double()
        runs[tag] = (_train_step(ns, G, GO, D, DO, latent, real, aug), latent)
    d = {"after_build_rng_check": _np(runs["f32"][1][0, :4])}
    for tag, (rec, _) in runs.items():
        for k in ("d_error", "penalty", "g_error"):
            d["%s/%s" % (tag, k)] = np.float64(rec[k].double())
        for net, gr in (("D", rec["d_grads"]), ("G", rec["g_grads"])):
            gam = [k for k in gr if k.endswith("gamma")]
            d["%s/%s/gamma_keys" % (tag, net)] = np.array(gam)
            d["%s/%s/gamma_grads" % (tag, net)] = np.array([float(gr[k].double().sum()) for k in gam])
            d["%s/%s/grad_keys" % (tag, net)] = np.array(list(gr.keys()))
            d["%s/%s/grad_norms" % (tag, net)] = np.array([float(v.double().norm()) for v in gr.values()])
        # post-optimizer-step norms: the first Nadam steps move an element by lr * g / (|g| + 1e-8) - a sign-like function of
        # the gradient, so elements whose gradient is at rounding-noise level land differently in the two precisions
        for net, sd in (("D", rec["d_sd_post_step"]), ("G", rec["g_sd_post_step"])):
            d["%s/%s/post_keys" % (tag, net)] = np.array(list(sd.keys()))
            d["%s/%s/post_norms" % (tag, net)] = np.array([float(v.double().norm()) for v in sd.values()])
    _save(name + "_f64", d)


F64_RECORDS = {
    "config1": ("g11_config1", 32, 8, 8),
    "config2": ("g14_config2_64", 64, 64, 8),
    "config3": ("g12_config3_128", 128, 2, 8),
    "size256": ("g13_256_narrow", 256, 2, 1),
    "size256_full": ("g20_256_full", 256, 2, 8),
}


GROUPS = {"f64": gen_f64, "branches": gen_branches, "ops": gen_ops, "tiny": gen_tiny, "init": gen_init, "config1": gen_config1, "config3": gen_config3,
          "config2": gen_config2, "size256": gen_size256, "size256_full": gen_size256_full, "variants": gen_variants}


def main(argv):
    if len(argv) >= 2 and argv[1] in GROUPS:
        GROUPS[argv[1]]()
        return
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    runs = [("ops", {}), ("branches", {}), ("tiny", {}), ("config1", {}), ("config3", {}), ("config2", {}), ("size256", {}), ("size256_full", {})]
    runs += [("init", {"LOCATE_GOLDEN_INIT_CFG": c}) for c in ("tiny32", "full32", "full64")]
    runs += [("variants", {"LOCATE_GOLDEN_VARIANT": v}) for v in VARIANTS]
    runs += [("f64", {"LOCATE_GOLDEN_F64": v}) for v in F64_RECORDS]
    for name, extra in runs:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), name], env=dict(env, **extra),
                              cwd="/tmp")


if __name__ == "__main__":
    main(sys.argv)
