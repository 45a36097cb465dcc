"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared symbol, the module
tree has the reference's state_dict layout, and seeded construction reproduces the reference's RNG draw order
(tests/golden/g10_init_*.npz, generated from the reference by oracle/gen_golden.py)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
import locate_amd
from locate_amd import Discriminator, Generator, NetConfig, get_model, parameter_count
from locate_amd._lib import PROTOTYPES, LocateError, lib
from locate_amd.models import discriminator_features, generator_features


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    header = open(os.path.join(ROOT, "include", "locate_hip.h")).read()
    declared = set(re.findall(r"\b(locate_[a-z0-9_]+)\s*\(", header))
    assert declared == set(PROTOTYPES), declared ^ set(PROTOTYPES)
    handle = lib()
    for name in declared:
        assert hasattr(handle, name), name
    from locate_amd._lib import EXPECTED_ABI
    assert handle.locate_abi_version() == EXPECTED_ABI
    assert handle.locate_sn_table_record_bytes() == 80
    assert handle.locate_nadam_tensor_record_bytes() == 56


def test_no_cpu_fallback():
    """The product path must fail loudly without an MI355X: CPU tensors are rejected, not silently computed."""
    from locate_amd import ops
    with pytest.raises(TypeError):
        ops.root_tanh(torch.randn(8))
    if not torch.cuda.is_available():
        from locate_amd._lib import require_gpu
        with pytest.raises(LocateError):
            require_gpu()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "locate_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the CPU oracle", ""), f


def test_feature_lists():
    assert generator_features(NetConfig(64)) == [64, 768, 384, 192, 96, 48]
    assert discriminator_features(NetConfig(64)) == [32, 64, 128, 256, 512, 512]
    assert generator_features(NetConfig(128)) == [128, 1536, 768, 384, 192, 96, 48]
    assert discriminator_features(NetConfig(256)) == [32, 64, 128, 256, 512, 1024, 2048, 2048]
    assert generator_features(NetConfig(32, 1)) == [32, 48, 24, 12, 4]


@pytest.mark.parametrize("which,size,bff", [("tiny32", 32, 1), ("full32", 32, 8), ("full64", 64, 8)])
def test_seeded_construction_matches_reference(which, size, bff):
    z = load_golden("g10_init_" + which)
    cfg = NetConfig(image_size=size, base_feature_factor=bff)
    torch.manual_seed(cfg.seed)
    G, _ = get_model(Generator(cfg), cfg.glr, "cpu")
    D, _ = get_model(Discriminator(cfg), cfg.dlr, "cpu")
    assert G.g_in == int(z["G/g_in"])
    for tag, net in (("G", G), ("D", D)):
        sd = net.state_dict()
        assert list(sd.keys()) == list(z[tag + "/keys"])
        assert [str(tuple(v.shape)) for v in sd.values()] == list(z[tag + "/shapes"])
        np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], z[tag + "/sum"], rtol=0, atol=1e-9)
        np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], z[tag + "/abssum"], rtol=0, atol=1e-9)
        assert [bool(p.requires_grad) for _, p in net.named_parameters()] == list(z[tag + "/requires_grad"])
        assert parameter_count(net) == int(z[tag + "/param_count"])
    assert float(G.noise.double().sum()) == pytest.approx(float(z["G/noise_sum"]), abs=1e-9)
    assert "noise" not in G.state_dict()
    np.testing.assert_array_equal(torch.randn(4).numpy(), z["after_rng"])   # same RNG position afterwards


def test_requires_grad_toggle_makes_uv_trainable():
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    D = Discriminator(cfg)
    before = parameter_count(D)
    D.requires_grad_(False)
    D.requires_grad_(True)          # main.py:161,172
    z = load_golden("g8_tiny_e2e")
    assert before == int(z["meta/d_param_count"])
    assert parameter_count(D) == int(z["meta/d_param_count_after"])


def test_noise_follows_module_moves():
    G = Generator(NetConfig(image_size=32, base_feature_factor=1))
    G = G.double()
    assert G.noise.dtype == torch.float64


@pytest.mark.parametrize("miniter,diters", [(1, 1), (2, 1), (3, 2), (8, 1)])
def test_train_loop_schedule_matches_reference_loop(miniter, diters):
    """TrainLoop against a literal restatement of the reference's loop body (main.py:146-172): which phases run on
    which batch, in which order."""
    from locate_amd import TrainLoop

    class Recorder:
        def __init__(self):
            self.calls = []

        def d_forward_backward(self, latent, real, aug):
            self.calls.append(("d_fb", latent))
            return {"d": latent}

        def d_optimizer(self):
            self.calls.append(("d_opt",))

        def g_forward_backward(self, latent):
            self.calls.append(("g_fb", latent))
            return {"g": latent}

        def g_optimizer(self):
            self.calls.append(("g_opt",))

    rec = Recorder()
    loop = TrainLoop(rec, miniter=miniter, diters=diters)
    want = []
    for i in range(1, 20):
        out = loop.iteration(i, None, None)
        want.append(("d_fb", i))                       # gen(noise).detach(), dis.zero_grad(), three D passes, backward
        ran_g = False
        if i % miniter == 0:
            want.append(("d_opt",))
            if (i // miniter) % diters == 0:
                want += [("g_fb", i), ("g_opt",)]
                ran_g = True
        assert ("g" in out) == ran_g and out["d"] == i
    assert rec.calls == want


def test_conv_spec_geometry_for_grouped_layers():
    """Host-side geometry of the SEPARABLE layers (no GPU): depthwise conv / transposed conv with a channel multiplier and
    the full-size grouped conv, as handed to locate_dwconv_* / locate_groupdot_*; mismatched shapes are rejected."""
    from locate_amd.ops import ConvSpec
    # Conv2d(6, 12, 5, stride 2, pad 2, groups=6): weight [12, 1, 5, 5]
    geom, out = ConvSpec("conv", 5, 5, 2, 2, 2, mode="depthwise").geometry((4, 6, 16, 16), (12, 1, 5, 5))
    assert geom == [4, 6, 16, 16, 12, 5, 5, 2, 2, 2, 8, 8] and out == (4, 12, 8, 8)
    # ConvTranspose2d(6, 12, 4, stride 2, pad 1, groups=6): weight [6, 2, 4, 4]; R's input side is the layer's output
    geom, out = ConvSpec("convT", 4, 4, 2, 1, 1, mode="depthwise").geometry((4, 6, 8, 8), (6, 2, 4, 4))
    assert geom == [4, 12, 16, 16, 6, 4, 4, 2, 1, 1, 8, 8] and out == (4, 12, 16, 16)
    # Conv2d(32, 8, kernel = the 8 x 8 map, groups=8): 8 dot products of 4 * 64 contiguous values per sample
    geom, out = ConvSpec("conv", 8, 8, 1, 0, 0, mode="groupdot").geometry((5, 32, 8, 8), (8, 4, 8, 8))
    assert geom[:3] == [5, 8, 256] and out == (5, 8, 1, 1)
    with pytest.raises(ValueError):
        ConvSpec("conv", 5, 5, 2, 2, 2, mode="depthwise").geometry((4, 5, 16, 16), (12, 1, 5, 5))     # 12 % 5 != 0
    with pytest.raises(ValueError):
        ConvSpec("convT", 4, 4, 2, 1, 1, mode="depthwise").geometry((4, 7, 8, 8), (6, 2, 4, 4))
    with pytest.raises(ValueError):
        ConvSpec("conv", 8, 8, 1, 0, 0, mode="groupdot").geometry((5, 32, 8, 8), (8, 4, 4, 4))        # kernel != map


@pytest.mark.parametrize("switches", [dict(depth=2), dict(depth=3, feature_multiplier=2), dict(separable=True),
                                      dict(separable=True, depth=2, feature_multiplier=2)])
def test_architecture_switches_build_the_reference_key_layout(switches):
    """state_dict keys and shapes of the mirror under the non-default switches == the reference's (the g15 - g18 records keep
    the reference's gradient / post-step key lists; construction needs no GPU)."""
    from conftest import load_golden
    from locate_amd import Discriminator, Generator, NetConfig
    name = {(2, 1, False): "g15_depth2_32", (3, 2, False): "g16_depth3_fm2_32", (1, 1, True): "g17_separable_32",
            (2, 2, True): "g18_separable_depth2_fm2_64"}[(switches.get("depth", 1), switches.get("feature_multiplier", 1),
                                                          switches.get("separable", False))]
    size, ff = {"g15_depth2_32": (32, 2), "g16_depth3_fm2_32": (32, 2), "g17_separable_32": (32, 4),
                "g18_separable_depth2_fm2_64": (64, 2)}[name]
    z = load_golden(name)
    cfg = NetConfig(image_size=size, base_feature_factor=ff, **switches)
    torch.manual_seed(cfg.seed)
    G, D = Generator(cfg), Discriminator(cfg)
    assert list(G.state_dict().keys()) == z["G/post_keys"].tolist()
    assert list(D.state_dict().keys()) == z["D/post_keys"].tolist()


def test_checkpoint_roundtrip_reference_layout(tmp_path):
    """save_checkpoint writes the reference's netG.torch / netD.torch (plain state_dicts, main.py:235-236) plus the noise
    map the reference forgets (models.py:59); load_checkpoint restores all of it with weights_only loads, and also
    accepts a folder written by the reference (no side file)."""
    from locate_amd import load_checkpoint, save_checkpoint
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    torch.manual_seed(1)
    G, D = Generator(cfg), Discriminator(cfg)
    files = save_checkpoint(str(tmp_path), G, D)
    assert sorted(os.path.basename(f) for f in files) == ["netD.torch", "netG.extra.torch", "netG.torch"]
    raw = torch.load(os.path.join(tmp_path, "netG.torch"), weights_only=True)
    assert list(raw.keys()) == list(G.state_dict().keys()) and "noise" not in raw      # exactly the reference's file
    torch.manual_seed(2)
    G2, D2 = Generator(cfg), Discriminator(cfg)
    assert not torch.equal(G2.noise, G.noise)
    noise_obj = G2.noise
    rec = load_checkpoint(str(tmp_path), G2, D2)
    assert rec["noise_restored"] and G2.noise is noise_obj and torch.equal(G2.noise, G.noise)
    for a, b in ((G, G2), (D, D2)):
        for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
            assert k == k2 and torch.equal(v, v2), k
    # a reference-written folder: the g8 record's initial state_dicts saved the way main.py:235-236 does
    z = load_golden("g8_tiny_e2e")
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    for tag, name in (("G", "netG.torch"), ("D", "netD.torch")):
        torch.save({k[len(tag + "/sd0/"):]: torch.as_tensor(z[k]) for k in z.files if k.startswith(tag + "/sd0/")}, ref_dir / name)
    rec = load_checkpoint(str(ref_dir), G2, D2)
    assert not rec["noise_restored"]
    key = "conv_block.block_0.res_module_i.gamma"
    assert torch.equal(G2.state_dict()[key], torch.as_tensor(z["G/sd0/" + key]))


def test_nadam_state_dict_roundtrip_keeps_the_schedule_in_float64(tmp_path):
    """torch's Optimizer.load_state_dict casts float state to the parameter dtype; the (step, m_schedule) pair must come
    back as float64 (the schedule kernel reads two doubles) and the cached device tables must be dropped."""
    from locate_amd import Nadam
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    opt = Nadam(ps, lr=1e-3, betas=(0.5, 0.9))
    for p in ps:
        st = opt._state_for(p)
        st["sched"][0], st["sched"][1] = 3.0, 0.123456789012345678
        st["exp_avg"].normal_()
    opt._tables["stale"] = object()
    path = tmp_path / "opt.torch"
    torch.save(opt.state_dict(), path)
    opt2 = Nadam([torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))], lr=1e-3, betas=(0.5, 0.9))
    opt2._tables["stale"] = object()
    opt2.load_state_dict(torch.load(path, weights_only=True))
    assert opt2._tables == {}
    for p, q in zip(ps, opt2.param_groups[0]["params"]):
        a, b = opt.state[p], opt2.state[q]
        assert b["sched"].dtype == torch.float64 and torch.equal(a["sched"], b["sched"])
        assert torch.equal(a["exp_avg"], b["exp_avg"])


def test_architecture_descriptions_match_the_survey_tables():
    """locate_amd.arch against SURVEY.md Appendix A / section 8(a) a11 (measured on the reference): attention stages, style
    chain layout (`depths [1,1,3,1,3]`, `sums [0,1,2,5,6,9]` at 64x64; ten linears at 128x128) and its widths."""
    from locate_amd import arch
    cfg = NetConfig(64)
    feats = arch.generator_widths(cfg)
    plan = arch.stack_plan(5, 2, feats, [2] * 5, True, True, cfg)
    assert [st.side for st in plan] == [4, 8, 16, 32, 64]
    assert [st.attention for st in plan] == [False, False, True, False, True]
    assert [len(st.style) for st in plan] == [1, 1, 3, 1, 3]
    flat = [w for st in plan for w in st.style]
    assert flat[:4] == [(64, 64), (128, 768), (832, 384), (448, 192)]
    assert flat[4:] == [(256, 192), (256, 192), (256, 96), (160, 48), (112, 48)]
    G = Generator(cfg)
    assert G.conv_block.depths == [1, 1, 3, 1, 3] and G.conv_block.sums == [0, 1, 2, 5, 6, 9]
    assert sum(len(st.style) for st in arch.stack_plan(6, 2, arch.generator_widths(NetConfig(128)), [2] * 6, True, True,
                                                        NetConfig(128))) == 10
    dplan = arch.stack_plan(5, 32, arch.discriminator_widths(cfg), [2] * 5, False, False, cfg)
    assert [st.side for st in dplan] == [16, 8, 4, 2, 1] and [st.attention for st in dplan] == [True, False, False, False, False]
    assert all(st.style == () for st in dplan)
    # conv chains: kernels / pads of the three stage kinds, and the DEPTH > 1 layout
    assert arch.conv_chain(64, 64, True, 2, True, 1, cfg) == [arch.ConvLink(64, 64, 4, 2, 1, True, False, False)]
    assert arch.conv_chain(32, 64, False, 2, True, 1, cfg)[0][2:5] == (5, 2, 2)
    assert arch.conv_chain(48, 3, False, 1, False, 1, cfg)[0][2:5] == (3, 1, 1)
    chain = arch.conv_chain(32, 32, False, 2, True, 3, cfg)
    assert [(l.cin, l.cout, l.normalized, l.residual) for l in chain] == [(32, 8, False, False), (8, 8, False, True), (8, 32, True, False)]
    assert [c[:4] for c in arch.squeeze_plan(16, 192, cfg)] == [(192, 48, (16, 1), 1), (48, 48, (1, 16), 1), (48, 192, (1, 1), 1)]


def test_networks_do_not_share_runtime_state():
    """Every network owns its ops.Runtime; stacking calls in one discriminator is invisible to another model's layers
    (the class-level state of round 1 is gone), and a deep copy gets a fresh runtime shared by all of ITS layers."""
    import copy
    from locate_amd import ops
    from locate_amd.nn import InPlaceNorm, SpectralNorm
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    D1, D2, G = Discriminator(cfg), Discriminator(cfg), Generator(cfg)
    rts = {id(D1.runtime), id(D2.runtime), id(G.runtime), id(ops.DEFAULT_RUNTIME)}
    assert len(rts) == 4
    for net in (D1, D2, G):
        bound = [m for m in net.modules() if isinstance(m, (InPlaceNorm, SpectralNorm))]
        assert bound and all(m.runtime is net.runtime for m in bound)
    with D1.runtime.stacked_calls(3):
        assert D1.runtime.stacked == 3 and D2.runtime.stacked == 1 and ops.DEFAULT_RUNTIME.stacked == 1
    assert D1.runtime.stacked == 1
    D3 = copy.deepcopy(D1)
    assert D3.runtime is not D1.runtime
    assert all(m.runtime is D3.runtime for m in D3.modules() if isinstance(m, (InPlaceNorm, SpectralNorm)))
    assert SpectralNorm(torch.nn.Conv2d(3, 4, 1)).runtime is None            # stand-alone layer: the default runtime at call time



def test_fork_gradient_protocol_on_plain_tensors():
    """ops.fork is pure autograd plumbing (no kernel): two consumers that know nothing about the slot get the ordinary sum;
    two that claim the slot's buffer and return it (the second adding into it) are not added a second time; under no_grad
    the fork is the identity."""
    from locate_amd import ops
    x = torch.randn(5, 3, requires_grad=True)
    a, b = ops.fork(x * 1.0)
    (a.sin().sum() + (2.0 * b).sum()).backward()
    torch.testing.assert_close(x.grad, x.detach().cos() + 2.0)

    class Claims(torch.autograd.Function):           # stands for a kernel with an `accumulate` flag
        @staticmethod
        def forward(ctx, t, factor, slot):
            ctx.factor, ctx.slot = factor, slot
            return t * factor

        @staticmethod
        def backward(ctx, g):
            buf, acc = ctx.slot.claim(g)
            if acc:
                buf.add_(g * ctx.factor)
            else:
                buf.copy_(g * ctx.factor)
            return buf, None, None

    y = torch.randn(4, 2, requires_grad=True)
    a, b = ops.fork(y * 1.0)
    assert a._locate_slot is b._locate_slot
    (Claims.apply(a, 3.0, a._locate_slot).sum() + Claims.apply(b, 5.0, b._locate_slot).sum()).backward()
    torch.testing.assert_close(y.grad, torch.full_like(y, 8.0))
    with torch.no_grad():
        u, v = ops.fork(y)
    assert u is y and v is y
