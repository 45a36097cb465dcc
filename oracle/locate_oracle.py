"""CPU oracle: a functional restatement of the LocAtE generator/discriminator training step.

TEST INFRASTRUCTURE ONLY.  Nothing in `locate_amd/` imports this file; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may use it, and only as the checker /
reported CPU baseline - never as the thing measured or shipped.

Parity status: PINNED by fixtures generated from the reference itself (tests/golden/*.npz, made by
oracle/gen_golden.py which imports /root/reference in the build container).  The reference ships no
tests or golden vectors of its own, and its arithmetic kernels are PyTorch ATen
CPU ops; this oracle therefore composes the same ATen CPU ops (conv2d, conv_transpose2d, linear,
softmax, interpolate, avg_pool2d) in the same order, and restates by hand every piece the reference
hand-writes: the three custom autograd Functions and the spectral-norm state machine.

The oracle is *functional*: a network is a flat dict  name -> tensor  with exactly the reference's
`state_dict()` key layout (e.g. `conv_block.block_0.res_module_i.layer_module.module.conv_0.conv_0.
module.weight_bar`), plus `NetConfig` which replaces the reference's import-time constants
(libs/config.py).  All citations are file:line in /root/reference.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# configuration (replaces libs/config.py constants; values are inputs to the hot path)
# ------------------------------------------------------------------------------------------------
@dataclass
class NetConfig:
    image_size: int = 128          # config.py:37
    base_feature_factor: int = 8   # config.py:58
    factor: int = 2                # config.py:44
    bottleneck: int = 4            # config.py:62
    min_attention_size: int = 8    # config.py:63
    attention_every: int = 2       # config.py:64
    depth: int = 1                 # config.py:68  DEPTH
    feature_multiplier: int = 1    # config.py:55  FEATURE_MULTIPLIER
    separable: bool = False        # config.py:53  SEPARABLE
    glr: float = 5e-4              # config.py:70
    dlr: float = 2e-3              # config.py:71
    beta1: float = 0.5             # config.py:72
    beta2: float = 0.9             # config.py:73

    @property
    def z(self):                   # config.py:65  INPUT_VECTOR_Z = IMAGE_SIZE
        return self.image_size

    @property
    def n_blocks(self):            # models.py:37 (LAYERS - 1, config.py:50)
        return int(math.log(self.image_size, 2)) - 1

    def g_features(self):
        """models.py:16-22,43-52 + utils.py:28-31: [Z, 48*2^(n-1) ... 48] at base factor 8."""
        n = self.n_blocks
        gen = self.factor ** int(math.log(self.image_size, 2)) * self.base_feature_factor * 3  # config.py:60
        widths = [int(gen * self.factor ** (idx - n)) // 4 * 4 for idx in range(n - 1, -1, -1)]
        return [self.z] + widths

    def d_features(self):
        """models.py:25-31,76-78: [32, 64, ..., last, last] at base factor 8."""
        n = self.n_blocks
        dis = self.factor ** int(math.log(self.image_size, 2)) * self.base_feature_factor      # config.py:61
        widths = [int(dis * self.factor ** ((idx + 1) - n)) // 4 * 4 for idx in range(n)]
        return widths + [widths[-1]]

    def g_block_sizes(self):       # block.py:61-70 with in_size=2 (models.py:53): output side of block i
        return [2 * 2 ** (i + 1) for i in range(self.n_blocks)]

    def d_block_sizes(self):       # in_size = IMAGE_SIZE // 2 (models.py:82)
        return [int(self.image_size // 2 / 2 ** (i + 1) + 1 - 1e-12) for i in range(self.n_blocks)]

    def has_attention(self, size, idx):  # block.py:28-29
        return size >= self.min_attention_size and idx % self.attention_every == 0


# ------------------------------------------------------------------------------------------------
# the three hand-written autograd Functions of the reference, restated
# ------------------------------------------------------------------------------------------------
class RootTanhFn(torch.autograd.Function):
    """activation.py:7-36.  y = (x^2+1)^(1/4) tanh x ;  dx = g [2(x^2+1) sech^2 x + x tanh x] / (2 (x^2+1)^(3/4)).
    cosh(x)^2 overflows to inf for large |x| in fp32 -> 1/inf = 0 -> the sech term vanishes (finite result)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return (x * x + 1).pow(0.25) * torch.tanh(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        q = x * x + 1
        sech2 = (torch.cosh(x) ** 2).reciprocal()
        num = sech2 * q * 2 + torch.tanh(x) * x
        return num / (q.pow(0.75) * 2) * g


root_tanh = RootTanhFn.apply


class NormFn(torch.autograd.Function):
    """inplace_norm.py:4-27,40-45 fused: out = (x - mean(x)) * y / std(x) + b with GLOBAL scalar mean and unbiased
    std over the whole tensor.  The reference computes std outside its Function (so autograd adds the std path);
    here the whole thing is one closed form:
        dx = y g / s - mean(y g / s) + dz (x - mu) / ((N-1) s),   dz = -sum((x-mu) g y) / s^2
        dy = sum_bcast (x-mu) g / s,   db = sum_bcast g."""

    @staticmethod
    def forward(ctx, x, y, b):
        mu = x.mean()
        s = x.std()
        ctx.save_for_backward(x, y, mu, s)
        ctx.b_shape = b.shape
        return (x - mu) * y / s + b

    @staticmethod
    def backward(ctx, g):
        x, y, mu, s = ctx.saved_tensors
        n = x.numel()
        xc = x - mu
        xg = y * g / s
        dz = -(xc * g * y).sum() / (s * s)
        dx = xg - xg.mean() + dz * xc / ((n - 1) * s)
        dy = (xc * g / s).sum_to_size(y.shape)
        db = g.sum_to_size(ctx.b_shape)
        return dx, dy, db


def inplace_norm(x, y, b):
    return NormFn.apply(x, y, b)


class GateFn(torch.autograd.Function):
    """merge.py:19-39.  out = (gamma a + 1) x.  Backward as coded: dx = (gamma a + 1) g, da = gamma x g and
    (reference bug, reproduced) dgamma = sum x^2 g  instead of sum x a g."""

    @staticmethod
    def forward(ctx, x, a, gamma):
        ctx.save_for_backward(x, a, gamma)
        return (a * gamma + 1) * x

    @staticmethod
    def backward(ctx, g):
        x, a, gamma = ctx.saved_tensors
        xg = x * g
        return (a * gamma + 1) * g, xg * gamma, xg * x


def residual_gate(x, a, gamma):
    """merge.py:55-62: gamma [1,1] broadcast over the layer output."""
    gam = gamma.view(*[1] * a.dim()).expand_as(a)
    return GateFn.apply(x, a, gam)


# ------------------------------------------------------------------------------------------------
# spectral norm (spectral_norm.py:8-32,57-59)
# ------------------------------------------------------------------------------------------------
def _l2n(v, eps=1e-12):
    return v / (v.norm() + eps)


class SigmaFn(torch.autograd.Function):
    """sigma = u . (W v) (spectral_norm.py:31) with the reference's autograd behaviour made explicit:
    u and v are overwritten through `.data` on every forward (spectral_norm.py:28-29) which autograd does
    not version-track, so the backward of EVERY earlier forward sees the u, v left by the LATEST forward
    (checked against the reference: tests/golden/g4_spectral_norm.npz, three forwards then one backward):
        dW = dsigma * u_latest v_latest^T,  du = dsigma * (W v_k)  [saved intermediate],  dv = dsigma * W^T u_latest."""

    @staticmethod
    def forward(ctx, w2d, u, v):
        wv = w2d.mv(v)
        ctx.save_for_backward(w2d, wv)
        ctx.u, ctx.v = u, v          # live references, read at backward time
        return u.dot(wv)

    @staticmethod
    def backward(ctx, ds):
        w2d, wv = ctx.saved_tensors
        u, v = ctx.u.detach(), ctx.v.detach()
        dw = ds * torch.outer(u, v) if ctx.needs_input_grad[0] else None
        du = ds * wv if ctx.needs_input_grad[1] else None
        dv = ds * w2d.t().mv(u) if ctx.needs_input_grad[2] else None
        return dw, du, dv


def sn_weight(P, prefix, power_iterations=1):
    """`power_iterations` power iterations (spectral_norm.py:26-29; state update through .data, no graph; the networks use 1) +
    normalised weight W_bar / sigma."""
    w, u, v = P[prefix + "weight_bar"], P[prefix + "weight_u"], P[prefix + "weight_v"]
    h = w.shape[0]
    with torch.no_grad():
        w2 = w.detach().reshape(h, -1)
        for _ in range(power_iterations):
            v.data.copy_(_l2n(w2.t().mv(u.detach())))
            u.data.copy_(_l2n(w2.mv(v.detach())))
    sigma = SigmaFn.apply(w.reshape(h, -1), u, v)
    return w / sigma


# ------------------------------------------------------------------------------------------------
# layers, addressed by state_dict prefix
# ------------------------------------------------------------------------------------------------
def activated_base_conv(P, prefix, x, kernel, stride, pad, transposed, separable=False):
    """conv.py:11-24: SN 1x1 ( RootTanh ( SN kxk ( RootTanh x ) ) ), no biases; both convs transposed when
    `transposed` (conv.py:49-52) so weights are [C_in, C_out, k, k].  SEPARABLE: the k x k conv is depthwise
    (groups = C_in, conv.py:17) - the channel multiplier is read off the weight."""
    w0 = sn_weight(P, prefix + "conv_0.module.")
    groups = x.shape[1] if separable else 1
    h = root_tanh(x)
    if transposed:
        h = F.conv_transpose2d(h, w0, None, stride, pad, groups=groups)
    else:
        h = F.conv2d(h, w0, None, stride, pad, groups=groups)
    w1 = sn_weight(P, prefix + "conv_1.module.")
    h = root_tanh(h)
    return F.conv_transpose2d(h, w1) if transposed else F.conv2d(h, w1)


def deep_residual_conv(P, prefix, x, transposed, stride, cfg=None, cin=None, cout=None, use_bottleneck=True, depth=1):
    """conv.py:27-72.  Stage 0: kernel = 2*stride + (0 if transposed else 1) (conv.py:36), pads utils.py:34-39, to the
    bottleneck width when depth > 1 (conv.py:31-34,61-62).  Stages 1 .. depth-1 (conv.py:63-67): 5x5 stride-1 regular
    convs; `residual` receives the conv class (truthy), so each is a ResModule(m=1) when its widths agree (conv.py:55-56),
    around a Norm from the second of them on (`normalize=bool(i)` / `bool(depth - 2)`)."""
    sep = bool(cfg.separable) if cfg is not None else False
    kernel = stride * 2 + int(not transposed)
    pad = max(kernel // 2 - stride // 2, 0) if transposed else kernel // 2
    x = activated_base_conv(P, prefix + "conv_0.", x, kernel, stride, pad, transposed, sep)
    if depth <= 1:
        return x
    mid = min(cin, cout)
    if use_bottleneck and max(cin, cout) // mid < cfg.bottleneck:
        mid //= cfg.bottleneck
    stages = [(mid, mid, bool(i)) for i in range(depth - 2)] + [(mid, cout, bool(depth - 2))]
    for k, (a, b, normalize) in enumerate(stages, start=1):
        p = prefix + "conv_%d." % k
        residual = a == b
        q = p + "layer_module." if residual else p
        h = x
        if normalize:
            h = inplace_norm(h, P[q + "i_norm.weight"], P[q + "i_norm.bias"])
            q += "module."
        h = activated_base_conv(P, q, h, 5, 1, 2, False, sep)
        x = residual_gate(x, h, P[p + "gamma"]) if residual else h
    return x


def feature_pooling(x, out_features):
    """scale.py:7-16: raw memory view [B, C_out, H, W, r] then mean over r: averages r adjacent flat elements."""
    size = list(x.shape)
    size[1] = out_features
    return x.contiguous().view(*size, -1).mean(dim=-1)


def scale_layer(P, prefix, x, cin, cout, stride, transposed):
    """scale.py:19-45 (skip branch): channel change then resample."""
    n_layers = int(cin != cout) + int(stride > 1)
    sub = (prefix + "0.") if n_layers > 1 else prefix     # nn.Sequential only when more than one layer
    if cin > cout:
        if cin % cout == 0:
            x = feature_pooling(x, cout)
        else:
            w = sn_weight(P, sub + "module.")
            x = F.conv2d(x, w, P[sub + "module.bias"])
    elif cout > cin:                                     # CatModule(identity, SN conv1x1 with bias), merge.py:4-16
        w = sn_weight(P, sub + "layer_module.module.")
        x = torch.cat([x, F.conv2d(x, w, P[sub + "layer_module.module.bias"])], dim=1)
    if stride > 1:
        if transposed:
            x = F.interpolate(x, scale_factor=stride, mode="bilinear", align_corners=False)
        else:
            x = F.avg_pool2d(x, stride, stride)
    return x


def feature_attention(P, prefix, x, features, size, bottleneck, separable=False):
    """attention.py:9-37: conv (S x 1) -> RootTanh -> conv (1 x S) -> RootTanh -> conv 1x1 -> softmax over channels ->
    expand to [B, C, S, S] (util_modules.py:6-12).  SEPARABLE (attention.py:15-21): ONE full-size (S x S) conv with
    groups = C // BOTTLENECK takes the place of the pair - and no RootTanh separates it from the 1x1 conv."""
    bf = features // bottleneck
    if separable and features % min(features, bf) == 0:
        h = F.conv2d(x, sn_weight(P, prefix + "0.module."), groups=bf)
        h = F.conv2d(h, sn_weight(P, prefix + "1.module."))
    else:
        h = F.conv2d(x, sn_weight(P, prefix + "0.module."))
        h = root_tanh(h)
        h = F.conv2d(h, sn_weight(P, prefix + "2.module."))
        h = root_tanh(h)
        h = F.conv2d(h, sn_weight(P, prefix + "4.module."))
    h = torch.softmax(h, dim=1)
    return h.view(h.size(0), -1, 1, 1).expand(-1, features, size, size)


def self_attention(P, prefix, x):
    """attention.py:40-54: softmax over the N = H*W positions of W2 RootTanh(W1 x_flat); no QK^T anywhere."""
    b, c = x.shape[:2]
    h = x.reshape(b, c, -1)
    h = F.conv1d(h, sn_weight(P, prefix + "conv_0.module."))
    h = root_tanh(h)
    h = F.conv1d(h, sn_weight(P, prefix + "conv_1.module."))
    return torch.softmax(h, dim=-1).view_as(x)


def _norm_scale(P, prefix, scale):
    return P[prefix + "i_norm.weight"] if scale is None else scale


def block_forward(P, prefix, x, cin, cout, size, idx, transposed, cfg, scales=None):
    """block.py:44-52."""
    if scales is None:
        scales = [None] * 4
    scaled = scale_layer(P, prefix + "scale_layer.", x, cin, cout, 2, transposed)
    p = prefix + "res_module_i."
    h = inplace_norm(x, _norm_scale(P, p + "layer_module.", scales[0]), P[p + "layer_module.i_norm.bias"])
    h = deep_residual_conv(P, p + "layer_module.module.", h, transposed, 2, cfg, cin, cout, True, cfg.depth)
    out = residual_gate(scaled, h, P[p + "gamma"])
    if cfg.has_attention(size, idx):
        p = prefix + "res_module_f."
        h = inplace_norm(out, _norm_scale(P, p + "layer_module.", scales[1]), P[p + "layer_module.i_norm.bias"])
        h = feature_attention(P, p + "layer_module.module.", h, cout, size, cfg.bottleneck, cfg.separable)
        out = residual_gate(out, h, P[p + "gamma"])
        p = prefix + "res_module_s."
        h = inplace_norm(out, _norm_scale(P, p + "layer_module.", scales[2]), P[p + "layer_module.i_norm.bias"])
        h = self_attention(P, p + "layer_module.module.", h)
        out = residual_gate(out, h, P[p + "gamma"])
    return out


def linear_module(P, prefix, x):
    """linear.py:7-15: SN(Linear with bias); returns (RootTanh(out), out)."""
    out = F.linear(x, sn_weight(P, prefix + "module.module."), P[prefix + "module.module.bias"])
    return root_tanh(out), out


def generator_forward(P, noise, latent, cfg):
    """models.py:61-66 + block.py:112-127.  `noise` is G.noise [1, Z, 2, 2] (a plain tensor, not in state_dict)."""
    feats, sizes = cfg.g_features(), cfg.g_block_sizes()
    x = noise.expand(latent.size(0), -1, -1, -1)
    chain = None
    mul_idx = 0
    for i in range(cfg.n_blocks):
        attn = cfg.has_attention(sizes[i], i)
        operand = []
        for _ in range(3 if attn else 1):
            chain = latent if chain is None else torch.cat([latent, chain], dim=1)
            chain, pre = linear_module(P, "conv_block.mul_block_%d." % mul_idx, chain)
            mul_idx += 1
            operand.append(pre.view(*pre.shape, 1, 1))
        x = block_forward(P, "conv_block.block_%d." % i, x, feats[i], feats[i + 1], sizes[i], i, True, cfg, operand)
    x = deep_residual_conv(P, "out_conv.", x, False, 1, cfg)
    return torch.tanh(x)


def discriminator_forward(P, x, cfg):
    """models.py:96-97: stem ResModule(Scale(3->w0), DRC(3->w0, 5x5 s2)) -> blocks -> 3x3 + 1x1 head."""
    feats, sizes = cfg.d_features(), cfg.d_block_sizes()
    scaled = scale_layer(P, "main.0.residual_module.", x, 3, feats[0], 2, False)
    h = deep_residual_conv(P, "main.0.layer_module.", x, False, 2, cfg)
    x = residual_gate(scaled, h, P["main.0.gamma"])
    for i in range(cfg.n_blocks):
        x = block_forward(P, "main.1.block_%d." % i, x, feats[i], feats[i + 1], sizes[i], i, False, cfg)
    return deep_residual_conv(P, "main.2.", x, False, 1, cfg)


# ------------------------------------------------------------------------------------------------
# loss glue + optimizer + step (utils.py:133-134, grad_penalty.py:1-2, nadam.py:31-89, main.py:142-172)
# ------------------------------------------------------------------------------------------------
def hinge(x):
    return (1 - x).clamp(min=0)


def consistency_penalty(d_true, d_aug, gamma=100):
    """grad_penalty.py:1-2 - not a gradient penalty: 100 (mean D(real) - mean D(aug))^2."""
    return gamma * (d_true.mean() - d_aug.view(-1).mean()) ** 2


class Nadam:
    """nadam.py:31-89 restated; per-tensor step / m_schedule (tensors whose grad was None skip the update)."""

    def __init__(self, lr, betas, eps=1e-8, schedule_decay=4e-3, weight_decay=0.0):
        self.lr, self.betas, self.eps, self.schedule_decay, self.weight_decay = lr, betas, eps, schedule_decay, weight_decay
        self.state = {}

    def step(self, P, grads):
        b1, b2 = self.betas
        for name, p in P.items():
            g = grads.get(name)
            if g is None:
                continue
            st = self.state.setdefault(name, dict(step=0, m_schedule=1.0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
            st["step"] += 1
            t = st["step"]
            if self.weight_decay != 0:                      # nadam.py:65-66
                g = g.add(p.detach(), alpha=self.weight_decay)
            mc_t = b1 * (1.0 - 0.5 * 0.96 ** (t * self.schedule_decay))
            mc_t1 = b1 * (1.0 - 0.5 * 0.96 ** ((t + 1) * self.schedule_decay))
            ms_new = st["m_schedule"] * mc_t
            ms_next = ms_new * mc_t1
            st["m_schedule"] = ms_new
            st["m"].mul_(b1).add_(g, alpha=1.0 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (st["v"] / (1.0 - b2 ** t)).sqrt_().add_(self.eps)
            with torch.no_grad():
                p.addcdiv_(g, denom, value=-self.lr * (1.0 - mc_t) / (1.0 - ms_new))
                p.addcdiv_(st["m"], denom, value=-self.lr * mc_t1 / (1.0 - ms_next))


def _grads_of(P):
    return {k: p.grad for k, p in P.items() if p.grad is not None}


def _zero_grad(P):
    for p in P.values():
        p.grad = None


def is_uv(name):
    return name.endswith("weight_u") or name.endswith("weight_v")


def make_params(sd, trainable_uv=False, dtype=None):
    """state_dict-like mapping -> leaf tensors with the reference's requires_grad flags (u, v frozen at
    construction, spectral_norm.py:45-46)."""
    P = {}
    for k, v in sd.items():
        t = torch.as_tensor(v).clone()
        if dtype is not None:
            t = t.to(dtype)
        t.requires_grad_(trainable_uv or not is_uv(k))
        P[k] = t
    return P


def train_step(PG, PD, g_noise, opt_g, opt_d, latent, real, aug, cfg):
    """One iteration of main.py:142-172 with miniter = MINIBATCHES = DITERS = 1.  Returns a record of the same
    quantities tests/golden/g8_tiny_e2e.npz holds."""
    rec = {}
    with torch.enable_grad():
        generated = generator_forward(PG, g_noise, latent, cfg).detach()      # main.py:146
        _zero_grad(PD)                                                        # :148
        d_true = discriminator_forward(PD, real, cfg).view(-1)                # :149
        d_gen = -discriminator_forward(PD, generated, cfg).view(-1)           # :150
        d_error = (hinge(d_true) + hinge(d_gen)).mean()                       # :151-155
        pen = consistency_penalty(d_true, discriminator_forward(PD, aug, cfg))
        (d_error + pen).backward()                                            # :156
    rec.update(generated=generated, d_true=d_true.detach(), d_gen=d_gen.detach(), d_error=d_error.detach(),
               penalty=pen.detach(), d_grads=_grads_of(PD),
               d_uv_pre_step={k: v.detach().clone() for k, v in PD.items() if is_uv(k)})
    opt_d.step(PD, rec["d_grads"])                                            # :159
    rec["d_post_step"] = {k: v.detach().clone() for k, v in PD.items()}
    flags = {k: p.requires_grad for k, p in PD.items()}
    for p in PD.values():
        p.requires_grad_(False)                                               # :161
    _zero_grad(PG)                                                            # :163
    with torch.enable_grad():
        fake = generator_forward(PG, g_noise, latent, cfg)
        g_error = hinge(discriminator_forward(PD, fake, cfg).view(-1)).mean() # :164-168
        g_error.backward()                                                    # :169
    rec.update(fake=fake.detach(), g_error=g_error.detach(), g_grads=_grads_of(PG))
    opt_g.step(PG, rec["g_grads"])                                            # :171
    for p in PD.values():
        p.requires_grad_(True)                                                # :172 - u and v become trainable
    del flags
    rec["g_post_step"] = {k: v.detach().clone() for k, v in PG.items()}
    rec["d_uv_end"] = {k: v.detach().clone() for k, v in PD.items() if is_uv(k)}
    return rec
