"""BASELINE.json configs[3] and configs[4] AT THEIR NAMED SIZES (128x128 batch 32; 256x256 batch 16, full width), where the
CPU oracle cannot follow: size-independent properties instead of element-wise records (the records at batch 2 of the same
architectures are g12_config3_128 / g20_256_full in test_gpu_e2e.py).
  * one full G+D step: finite losses and gradients, bit-reproducible from the same seed (every reduction in the library has
    a fixed order), parameter counts as SURVEY.md Appendix A;
  * the widest ConvTranspose 4x4 stage (C = 1536 / 3072) is linear in its input at the named batch;
  * one full-width discriminator block: the stacked [3B] pass == three calls;
  * softmax over N = 65 536 positions (self-attention at 256x256, C = 48): against torch on the CPU."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _one_step(S, B):
    from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, parameter_count
    cfg = NetConfig(image_size=S)
    torch.manual_seed(cfg.seed)
    G, GO = get_model(Generator(cfg), cfg.glr, DEV)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, DEV)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    counts = (parameter_count(G), parameter_count(D))
    gen = torch.Generator().manual_seed(4321)
    latent = torch.randn(B, S, generator=gen).to(DEV)
    real = torch.randn(B, 3, S, S, generator=gen).clamp(-1, 1).to(DEV)
    aug = torch.randn(B, 3, S, S, generator=gen).clamp(-1, 1).to(DEV)
    rec = {}
    step = TrainStep(G, D, GO, DO)
    d_orig, g_orig = DO.step, GO.step

    def d_hook():
        rec["d"] = torch.stack([p.grad.double().norm() for p in D.parameters() if p.grad is not None]).cpu()
        return d_orig()

    def g_hook():
        rec["g"] = torch.stack([p.grad.double().norm() for p in G.parameters() if p.grad is not None]).cpu()
        return g_orig()
    DO.step, GO.step = d_hook, g_hook
    out = step(latent, real, aug)
    torch.cuda.synchronize()
    res = {k: out[k].detach().cpu().clone() for k in ("d_error", "penalty", "g_error")}
    res["fake_sum"] = out["fake"].double().sum().cpu()
    res["d"], res["g"] = rec["d"], rec["g"]
    res["w"] = torch.stack([p.detach().double().sum() for net in (G, D) for p in net.parameters()]).cpu()
    del step, G, D, GO, DO
    torch.cuda.empty_cache()
    return counts, res


@pytest.mark.parametrize("S,B,g_params,d_params", [(128, 32, 56398266, 46846560), (256, 16, 225919837, 186518113)])
def test_named_size_step_is_finite_and_reproducible(S, B, g_params, d_params):
    counts, a = _one_step(S, B)
    assert counts == (g_params, d_params)                     # SURVEY.md Appendix A, measured on the reference
    for k, v in a.items():
        assert torch.isfinite(v).all(), k
    # (whole tensors legitimately get an exactly-zero gradient at the first step: an attention gate whose gamma was drawn as
    # -1 + 0 + 1 = 0 (libs/merge.py:51-53) passes nothing back into its branch - but a dead backward pass would zero most of them)
    assert float((a["g"] > 0).double().mean()) > 0.5 and float((a["d"] > 0).double().mean()) > 0.5
    _, b = _one_step(S, B)
    for k in a:
        assert torch.equal(a[k], b[k]), "not bit-reproducible: " + k


@pytest.mark.parametrize("C,size,B", [(1536, 4, 32), (3072, 4, 16)])
def test_widest_transposed_stage_is_linear_at_named_batch(C, size, B):
    from locate_amd import ops
    torch.manual_seed(C)
    w = torch.randn(C, C, 4, 4, device=DEV) * 0.01
    u, v = torch.randn(C, device=DEV), torch.randn(C * 16, device=DEV)
    sigma, wv = torch.tensor([2.0, 0.5], device=DEV), torch.zeros(C, device=DEV)
    spec = ops.ConvSpec("convT", 4, 4, 2, 1, 1)
    x1, x2 = torch.randn(B, C, size, size, device=DEV), torch.randn(B, C, size, size, device=DEV)

    def f(x):
        with torch.no_grad():
            return ops.SNConvFn.apply(x, w, u, v, None, sigma, wv, spec)
    y1, y2, y12 = f(x1), f(x2), f(x1 + x2)
    assert_close(y12.cpu(), (y1 + y2).cpu(), 3e-6, "additivity at C = %d" % C)
    # one output element against an fp64 dot product taken straight from the definition
    b, co, oy, ox = 1, 5, 3, 6
    acc = 0.0
    for ky in range(4):
        for kx in range(4):
            iy, ix = (oy + 1 - ky), (ox + 1 - kx)
            if iy % 2 == 0 and ix % 2 == 0 and 0 <= iy // 2 < size and 0 <= ix // 2 < size:
                acc += float((x1[b, :, iy // 2, ix // 2].double() * w[:, co, ky, kx].double()).sum())
    assert abs(float(y1[b, co, oy, ox]) - 0.5 * acc) <= 2e-5 * max(abs(0.5 * acc), float(y1.abs().max()) * 1e-2)


def test_full_width_discriminator_block_stacked_equals_three_calls():
    import copy
    from locate_amd import NetConfig
    from locate_amd.nn import Block
    cfg = NetConfig(image_size=256)
    torch.manual_seed(3)
    blk = Block(2, 2048, 2048, 2, False, 6, cfg=cfg).to(DEV)      # the 256x256 discriminator's last block: 2048 -> 2048 at 2x2
    blk2 = copy.deepcopy(blk)
    from locate_amd import ops
    xs = [torch.randn(16, 2048, 2, 2, device=DEV) for _ in range(3)]
    ys = torch.cat([blk(x) for x in xs])
    sns = [m for m in blk2.modules() if type(m).__name__ == "SpectralNorm"]
    for sn in sns:
        mm = sn.module
        runs = [ops.sn_power_iteration(mm.weight_bar, mm.weight_u, mm.weight_v) for _ in range(3)]
        sn._pre = (torch.stack([r[0] for r in runs]), torch.stack([r[1] for r in runs]))
    with ops.stacked_calls(3):
        y_all = blk2(torch.cat(xs))
    assert_close(y_all.detach().cpu(), ys.detach().cpu(), 1e-5, "stacked block")


def test_softmax_over_65536_positions():
    from locate_amd import ops
    torch.manual_seed(11)
    x = torch.randn(2, 48, 65536) * 3.0
    y = ops.softmax_lastdim(x.to(DEV).requires_grad_(True))
    ref = torch.softmax(x.double(), dim=-1)
    assert_close(y.detach().cpu(), ref, 2e-6, "softmax N = 65536")
    assert float((y.detach().double().sum(-1) - 1).abs().max()) < 1e-5
