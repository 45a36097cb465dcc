"""The CPU oracle (oracle/locate_oracle.py) against the fixtures generated from the real reference
(tests/golden/*.npz, oracle/gen_golden.py).  CPU only.  Tolerances: fp32 noise floor of the reference
itself is <=7e-7 on outputs and <=1.3e-5 on gradients; per-op 1e-5, end-to-end
gradients 1e-4, normalised max error."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, load_golden
from oracle import locate_oracle as O

T = torch.as_tensor


def sub(z, prefix):
    return {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}


@pytest.mark.parametrize("tag,tol", [("f32", 2e-6), ("f64", 1e-12)])
def test_g1_roottanh(tag, tol):
    z = load_golden("g1_roottanh_" + tag)
    x = T(z["x"]).requires_grad_(True)
    y = O.root_tanh(x)
    y.backward(T(z["g"]))
    assert torch.isfinite(x.grad).all()
    assert_close(y, z["y"], tol, "y")
    assert_close(x.grad, z["dx"], tol, "dx")


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_g2_inplace_norm(case):
    z = load_golden("g2_inplace_norm")
    for mode, yname in (("w", "weight"), ("s", "scale")):
        p = "%s_%s_" % (case, mode)
        x = T(z[p + "x"]).requires_grad_(True)
        y = T(z[p + yname]).requires_grad_(True)
        b = T(z[p + "bias"]).requires_grad_(True)
        out = O.inplace_norm(x, y, b)
        out.backward(T(z[p + "g"]))
        assert_close(out, z[p + "out"], 1e-5, p + "out")
        assert_close(x.grad, z[p + "dx"], 2e-5, p + "dx")
        assert_close(y.grad, z[p + "d" + yname], 2e-5, p + "dy")
        assert_close(b.grad, z[p + "dbias"], 1e-5, p + "db")


def test_g3_residual():
    z = load_golden("g3_residual")
    for p in ("full_", "bc_"):
        x = T(z[p + "x"]).requires_grad_(True)
        a = T(z[p + "a"]).requires_grad_(True)
        gamma = T(z[p + "gamma"]).requires_grad_(True)
        out = O.residual_gate(x, a.expand_as(x), gamma)
        out.backward(T(z[p + "g"]))
        assert_close(out, z[p + "out"], 1e-6)
        assert_close(x.grad, z[p + "dx"], 1e-6)
        assert_close(a.grad, z[p + "da"], 1e-5)
        assert_close(gamma.grad, z[p + "dgamma"], 1e-5)     # the x^2 quirk


SN_CASES = {
    "conv5s2": lambda x, w, b: torch.nn.functional.conv2d(x, w, b, 2, 2),
    "conv3": lambda x, w, b: torch.nn.functional.conv2d(x, w, b, 1, 1),
    "conv1x1b": lambda x, w, b: torch.nn.functional.conv2d(x, w, b),
    "convT4s2": lambda x, w, b: torch.nn.functional.conv_transpose2d(x, w, b, 2, 1),
    "convT1x1": lambda x, w, b: torch.nn.functional.conv_transpose2d(x, w, b),
    "conv1d": lambda x, w, b: torch.nn.functional.conv1d(x, w, b),
    "convS1": lambda x, w, b: torch.nn.functional.conv2d(x, w, b),
    "conv1S": lambda x, w, b: torch.nn.functional.conv2d(x, w, b),
    "linear": lambda x, w, b: torch.nn.functional.linear(x, w, b),
}


@pytest.mark.parametrize("name", sorted(SN_CASES))
@pytest.mark.parametrize("uvg", [False, True])
def test_g4_spectral_norm(name, uvg):
    z = load_golden("g4_spectral_norm")
    tag = name + ("_uvg" if uvg else "")
    P = O.make_params(sub(z, tag + "/sd0/"), trainable_uv=uvg)
    x = T(z[tag + "/x"]).requires_grad_(True)
    outs = []
    for k in range(3):
        y = SN_CASES[name](x, O.sn_weight(P, "module."), P.get("module.bias"))
        outs.append(y)
        assert_close(y, z[tag + "/y%d" % k], 1e-5, "y%d" % k)
        assert_close(P["module.weight_u"], z[tag + "/u%d" % k], 1e-5)
        assert_close(P["module.weight_v"], z[tag + "/v%d" % k], 1e-5)
    sum((o * T(z[tag + "/g%d" % k])).sum() for k, o in enumerate(outs)).backward()
    assert_close(x.grad, z[tag + "/dx"], 2e-5, "dx")
    want = sub(z, tag + "/grad/")
    got = {k: p.grad for k, p in P.items() if p.grad is not None}
    assert set(got) == set(want)
    for k in want:
        assert_close(got[k], want[k], 5e-5, k)


def test_g5_indexing_bit_exact():
    z = load_golden("g5_indexing")
    for r in (4, 2):
        x = T(z["fpool%d_x" % r]).requires_grad_(True)
        y = O.feature_pooling(x, r)
        y.backward(T(z["fpool%d_g" % r]))
        assert torch.equal(y, T(z["fpool%d_y" % r]))
        assert torch.equal(x.grad, T(z["fpool%d_dx" % r]))


@pytest.mark.parametrize("name,args", [("scale_up_pool", (8, 4, 2, True)), ("scale_up_cat", (4, 12, 2, True)),
                                       ("scale_down_cat", (4, 8, 2, False)), ("scale_down_same", (6, 6, 2, False))])
def test_g5_scale(name, args):
    z = load_golden("g5_indexing")
    P = O.make_params(sub(z, name + "/sd0/"))
    x = T(z[name + "/x"]).requires_grad_(True)
    y = O.scale_layer(P, "", x, *args)
    y.backward(T(z[name + "/g"]))
    assert_close(y, z[name + "/y"], 1e-5)
    assert_close(x.grad, z[name + "/dx"], 1e-5)
    for k, v in sub(z, name + "/grad/").items():
        assert_close(P[k].grad, v, 5e-5, k)
    for k, v in sub(z, name + "/sd1/").items():
        assert_close(P[k], v, 1e-5, k)


def test_g6_attention_and_linear():
    z = load_golden("g6_attention")
    cfg = O.NetConfig()
    for name in ("fa", "sa"):
        P = O.make_params(sub(z, name + "/sd0/"))
        x = T(z[name + "/x"]).requires_grad_(True)
        y = O.feature_attention(P, "", x, 16, 8, cfg.bottleneck) if name == "fa" else O.self_attention(P, "", x)
        y.backward(T(z[name + "/g"]))
        assert_close(y, z[name + "/y"], 1e-5)
        assert_close(x.grad, z[name + "/dx"], 5e-5)
        for k, v in sub(z, name + "/grad/").items():
            assert_close(P[k].grad, v, 1e-4, name + k)
        for k, v in sub(z, name + "/sd1/").items():
            assert_close(P[k], v, 1e-5, k)
    P = O.make_params(sub(z, "lin/sd0/"))
    x = T(z["lin/x"]).requires_grad_(True)
    act, pre = O.linear_module(P, "", x)
    ((act * T(z["lin/g_act"])).sum() + (pre * T(z["lin/g_pre"])).sum()).backward()
    assert_close(act, z["lin/act"], 1e-5)
    assert_close(pre, z["lin/pre"], 1e-5)
    assert_close(x.grad, z["lin/dx"], 2e-5)
    for k, v in sub(z, "lin/grad/").items():
        assert_close(P[k].grad, v, 5e-5, k)


@pytest.mark.parametrize("name,cin,cout,size,idx,transposed", [
    ("up", 16, 8, 8, 0, True), ("up_na", 8, 8, 4, 1, True), ("down", 8, 16, 8, 0, False),
    ("down_na", 16, 16, 4, 1, False)])
def test_g7_blocks(name, cin, cout, size, idx, transposed):
    z = load_golden("g7_blocks")
    cfg = O.NetConfig()
    P = O.make_params(sub(z, name + "/sd0/"))
    x = T(z[name + "/x"]).requires_grad_(True)
    scales = None
    if transposed:
        scales = [T(z[name + "/scale%d" % i]).requires_grad_(True) for i in range(3) if name + "/scale%d" % i in z.files]
    y = O.block_forward(P, "", x, cin, cout, size, idx, transposed, cfg, scales)
    y.backward(T(z[name + "/g"]))
    assert_close(y, z[name + "/y"], 2e-5)
    assert_close(x.grad, z[name + "/dx"], 1e-4)
    if scales:
        for i, s in enumerate(scales):
            assert_close(s.grad, z[name + "/dscale%d" % i], 1e-4)
    want = sub(z, name + "/grad/")
    got = {k: p.grad for k, p in P.items() if p.grad is not None}
    assert set(got) == set(want)
    for k, v in want.items():
        assert_close(got[k], v, 2e-4, k)


def test_g9_nadam():
    z = load_golden("g9_nadam")
    P = {"p%d" % i: T(z["p%d_0" % i]).clone() for i in range(3)}
    opt = O.Nadam(float(z["lr"]), tuple(z["betas"]))
    for step in range(1, 4):
        grads = {"p%d" % i: T(z["g%d_%d" % (i, step)]) for i in range(3) if "g%d_%d" % (i, step) in z.files}
        opt.step(P, grads)
        for i in range(3):
            assert_close(P["p%d" % i], z["p%d_%d" % (i, step)], 2e-6, "p%d step %d" % (i, step))


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


def test_g8_tiny_end_to_end_two_steps():
    z = load_golden("g8_tiny_e2e")
    cfg = O.NetConfig(image_size=32, base_feature_factor=1)
    PG = O.make_params(sub(z, "G/sd0/"))
    PD = O.make_params(sub(z, "D/sd0/"))
    noise = T(z["G/noise"])
    og, od = O.Nadam(cfg.glr, (cfg.beta1, cfg.beta2)), O.Nadam(cfg.dlr, (cfg.beta1, cfg.beta2))
    assert sum(p.numel() for p in PG.values() if p.requires_grad) == int(z["meta/g_param_count"])
    assert sum(p.numel() for p in PD.values() if p.requires_grad) == int(z["meta/d_param_count"])
    for step in (1, 2):
        p = "step%d/" % step
        rec = O.train_step(PG, PD, noise, og, od, T(z[p + "latent"]), T(z[p + "real"]), T(z[p + "aug"]), cfg)
        otol, gtol = (1e-5, 1e-4) if step == 1 else (2e-4, 2e-3)   # step 2 compounds step-1 rounding
        for k in ("generated", "fake", "d_true", "d_gen", "d_error", "penalty", "g_error"):
            assert_close(rec[k], z[p + k], otol, p + k)
        for net, grads in (("D", rec["d_grads"]), ("G", rec["g_grads"])):
            want = sub(z, p + net + "/grad/")
            assert set(grads) == set(want), set(grads) ^ set(want)
            for k, v in want.items():
                assert_close(grads[k], v, gtol, p + net + k)
        assert sorted(z[p + "G/none_grads"].tolist()) == sorted(
            k for k, q in PG.items() if q.grad is None and q.requires_grad)
        for k, v in sub(z, p + "D/sd_pre_step/").items():
            assert_close(rec["d_uv_pre_step"][k], v, otol, k)
        for k, v in sub(z, p + "D/sd_post_step/").items():
            assert_step_close(rec["d_post_step"][k], v, cfg.dlr, step, p + "D post " + k)
        for k, v in sub(z, p + "G/sd_post_step/").items():
            assert_step_close(rec["g_post_step"][k], v, cfg.glr, step, p + "G post " + k)
        for k, v in sub(z, p + "D/sd_end/").items():
            assert_close(rec["d_uv_end"][k], v, 10 * otol, k)
    assert sum(q.numel() for q in PD.values() if q.requires_grad) == int(z["meta/d_param_count_after"])
    with torch.no_grad():
        img = O.generator_forward(PG, noise, T(z["sample/latent"]), cfg)
    assert_close(img, z["sample/image"], 5e-4)


def test_config_feature_lists():
    c = O.NetConfig(64)
    assert c.g_features() == [64, 768, 384, 192, 96, 48]
    assert c.d_features() == [32, 64, 128, 256, 512, 512]
    assert [c.has_attention(s, i) for i, s in enumerate(c.g_block_sizes())] == [False, False, True, False, True]
    assert [c.has_attention(s, i) for i, s in enumerate(c.d_block_sizes())] == [True, False, False, False, False]
    c = O.NetConfig(256)
    assert c.g_features() == [256, 3072, 1536, 768, 384, 192, 96, 48]
    assert c.d_features() == [32, 64, 128, 256, 512, 1024, 2048, 2048]


VARIANT_RECORDS = [          # the libs/config.py switches the shipped defaults leave off (oracle/gen_golden.py VARIANTS)
    ("g15_depth2_32", 32, 4, 2, dict(depth=2)),
    ("g16_depth3_fm2_32", 32, 4, 2, dict(depth=3, feature_multiplier=2)),
    ("g17_separable_32", 32, 4, 4, dict(separable=True)),
    ("g18_separable_depth2_fm2_64", 64, 2, 2, dict(separable=True, depth=2, feature_multiplier=2)),
]


@pytest.mark.parametrize("name,S,B,ff,switches", [("g11_config1", 32, 8, 8, {}), ("g13_256_narrow", 256, 2, 1, {})] + VARIANT_RECORDS)
def test_oracle_full_architectures_vs_reference_record(name, S, B, ff, switches):
    """The CPU oracle on the full architectures (seeded construction through the host mirror, which reproduces the
    reference's RNG draw order): losses, output and per-tensor gradient norms of one step as recorded from the
    reference (oracle/gen_golden.py g11 / g13, and g15 - g18 for DEPTH / FEATURE_MULTIPLIER / SEPARABLE)."""
    from locate_amd import Discriminator, Generator, NetConfig, init
    z = load_golden(name)
    cfg = NetConfig(image_size=S, base_feature_factor=ff, **switches)
    torch.manual_seed(cfg.seed)
    G = Generator(cfg)
    G.apply(init)
    D = Discriminator(cfg)
    D.apply(init)
    latent = torch.randn(B, S)
    real = torch.randn(B, 3, S, S).clamp(-1, 1)
    aug = torch.randn(B, 3, S, S).clamp(-1, 1)
    np.testing.assert_array_equal(latent[0, :4].numpy(), z["after_build_rng_check"])
    ocfg = O.NetConfig(image_size=S, base_feature_factor=ff, **switches)
    PG = O.make_params({k: v.clone() for k, v in G.state_dict().items()})
    PD = O.make_params({k: v.clone() for k, v in D.state_dict().items()})
    og, od = O.Nadam(ocfg.glr, (ocfg.beta1, ocfg.beta2)), O.Nadam(ocfg.dlr, (ocfg.beta1, ocfg.beta2))
    rec = O.train_step(PG, PD, G.noise.clone(), og, od, latent, real, aug, ocfg)
    for k in ("d_true", "d_gen", "d_error", "penalty", "g_error"):
        assert_close(rec[k].reshape(z[k].shape), z[k], 2e-5, k)
    assert_close(rec["fake"].flatten()[:16], z["fake_first"], 2e-5)
    for tag, got in (("D", rec["d_grads"]), ("G", rec["g_grads"])):
        keys = z[tag + "/grad_keys"].tolist()
        assert sorted(keys) == sorted(got)
        want = dict(zip(keys, z[tag + "/grad_norms"]))
        scale = max(want.values())
        for k in keys:
            g = float(got[k].double().norm())
            assert abs(g - want[k]) <= 2e-4 * max(want[k], 1e-3 * scale), (tag, k, g, want[k])


@pytest.mark.parametrize("name", ["g11_config1", "g12_config3_128", "g14_config2_64", "g13_256_narrow", "g20_256_full"])
def test_float64_reference_records_are_the_same_run(name):
    """The `_f64` records (oracle/gen_golden.py f64: the reference run in fp32 AND with both networks converted to float64) are
    taken on the same seeded build and inputs as the base record: their fp32 half reproduces the base record's numbers, and the
    float64 half sits within the fp32 noise the survey measured (gradients <= 1.3e-5 of the largest,
    ill-conditioned scalars aside)."""
    import os
    from conftest import GOLDEN_DIR
    if not os.path.exists(os.path.join(GOLDEN_DIR, name + "_f64.npz")):
        pytest.skip("no float64 record for " + name)
    z, z64 = load_golden(name), load_golden(name + "_f64")
    np.testing.assert_array_equal(z["after_build_rng_check"], z64["after_build_rng_check"])
    for k in ("d_error", "penalty", "g_error"):
        assert abs(float(z64["f32/" + k]) - float(z[k])) <= 1e-12 + 1e-6 * abs(float(z[k])), k
        assert abs(float(z64["f64/" + k]) - float(z[k])) <= 1e-4 * abs(float(z[k])), k
    for tag in ("D", "G"):
        assert z64["f32/%s/grad_keys" % tag].tolist() == z[tag + "/grad_keys"].tolist()
        np.testing.assert_allclose(z64["f32/%s/grad_norms" % tag], z[tag + "/grad_norms"], rtol=1e-12)
        np.testing.assert_allclose(z64["f32/%s/post_norms" % tag], z[tag + "/post_norms"], rtol=1e-12)
        a, b = z64["f32/%s/grad_norms" % tag], z64["f64/%s/grad_norms" % tag]
        assert float(np.max(np.abs(a - b)) / np.max(b)) <= 1e-4, tag


def test_g21_branches_the_training_loop_never_takes():
    """Nadam with weight_decay (nadam.py:65-66) and SpectralNorm(power_iterations > 1) (spectral_norm.py:26-29) against records
    taken from the reference (oracle/gen_golden.py branches)."""
    z = load_golden("g21_branches")
    P = {"p%d" % i: T(z["wd/p%d_0" % i]).clone() for i in range(2)}
    opt = O.Nadam(float(z["wd/lr"]), tuple(z["wd/betas"]), weight_decay=float(z["wd/weight_decay"]))
    for step in range(1, 4):
        opt.step(P, {"p%d" % i: T(z["wd/g%d_%d" % (i, step)]) for i in range(2)})
        for i in range(2):
            assert_close(P["p%d" % i], z["wd/p%d_%d" % (i, step)], 2e-6, "p%d step %d" % (i, step))
    for name, kind, stride, pad in (("conv3", "conv", 1, 1), ("convT4s2", "convT", 2, 1)):
        tag = "pi/" + name
        iters = int(z[tag + "/iters"])
        P = O.make_params({k[len(tag + "/sd0/"):]: T(z[k]) for k in z.files if k.startswith(tag + "/sd0/")}, trainable_uv=True)
        x = T(z[tag + "/x"]).clone().requires_grad_(True)
        outs = []
        for k in range(2):
            w = O.sn_weight(P, "module.", power_iterations=iters)
            y = F.conv2d(x, w, None, stride, pad) if kind == "conv" else F.conv_transpose2d(x, w, None, stride, pad)
            outs.append(y)
            assert_close(y, z[tag + "/y%d" % k], 1e-5, name + " y%d" % k)
            assert_close(P["module.weight_u"], z[tag + "/u%d" % k], 1e-6, name + " u%d" % k)
        sum((o * T(z[tag + "/g%d" % k])).sum() for k, o in enumerate(outs)).backward()
        assert_close(x.grad, z[tag + "/dx"], 2e-5, name + " dx")
        for k in z.files:
            if k.startswith(tag + "/grad/"):
                assert_close(P[k[len(tag + "/grad/"):]].grad, z[k], 5e-4 if k.endswith(("weight_u", "weight_v")) else 5e-5, k)
