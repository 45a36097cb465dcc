"""Host-side mirror of the reference's layer interface (same class names, constructor arguments, forward
signatures, parameter names and therefore the same `state_dict()` key layout), computing on MI355X through
`locate_amd.ops`.  torch.nn.Conv2d / ConvTranspose2d / Conv1d / Linear objects are used only as parameter
containers (shape, default initialisation, hyper-parameters): their own forward is never called.

What is fixed by the drop-in contract (and pinned by tests/golden/g10_init_*.npz): attribute names that become
`state_dict()` keys, the order in which parameters are registered, and the order in which constructors draw from the
RNG.  Everything else - how a stack is described, how the style chain is laid out, how a forward walks its parts - is
this package's own: a network is first DESCRIBED (`locate_amd/arch.py`: a list of `Stage` records with the widths of the
style chain as data) and then built from the description.

Reference interface -> here
    libs/activation.py:42-51   RootTanhModule / NonLinear
    libs/inplace_norm.py:34-56 InPlaceNorm, Norm
    libs/spectral_norm.py:12-59 SpectralNorm
    libs/merge.py:4-16,46-62   CatModule, ResModule
    libs/scale.py:7-45         FeaturePooling, Scale
    libs/util_modules.py:6-12  Expand
    libs/attention.py:9-54     feature_attention, SelfAttention
    libs/conv.py:11-72         ActivatedBaseConv, DeepResidualConv
    libs/linear.py:7-15        LinearModule
    libs/block.py:15-127       Block, BlockBlock
"""
import torch
from torch import nn

from . import arch, ops
from .config import get_default


def _identity(x):
    return x


class _Bound(nn.Module):
    """A layer whose kernels need the owning network's runtime state (ops.Runtime: stacked calls, backward-pass deferrals).
    `runtime` is set by the network after construction (models._NetBase.adopt); stand-alone layers use the default one."""
    runtime = None


class RootTanhModule(nn.Module):
    def forward(self, function_input):
        return ops.root_tanh(function_input)


NonLinear = RootTanhModule


class InPlaceNorm(_Bound):
    """Global-statistics normalisation + per-channel affine (or per-sample style scale)."""

    def __init__(self, features=1, dim=2):
        super().__init__()
        shape = (1, features) + (1,) * dim
        self.weight = nn.Parameter(torch.ones(shape))
        self.bias = nn.Parameter(torch.zeros(shape))

    def forward(self, function_input, scale=None, with_act=False):
        gain = self.weight if scale is None else scale
        return ops.inplace_norm(function_input, gain, self.bias, with_act, self.runtime)


class Norm(nn.Module):
    def __init__(self, features, module, dim=2):
        super().__init__()
        self.i_norm = InPlaceNorm(features, dim=dim)
        self.module = module

    def forward(self, function_input, scale=None):
        fused = bool(getattr(self.module, "starts_with_activation", False))
        normed = self.i_norm(function_input, scale, with_act=fused)      # fused: norm + the stage's leading RootTanh, one kernel
        return self.module(normed, pre_activated=True) if fused else self.module(normed)


class SpectralNorm(_Bound):
    """Wraps a torch.nn conv / linear: every forward runs one power iteration on (weight_u, weight_v) in place
    and applies the layer with weight_bar / sigma.  Parameters live on the wrapped module under the reference's
    names (`module.weight_bar|weight_u|weight_v`)."""

    _SUFFIXES = ("_u", "_v", "_bar")

    def __init__(self, module, name="weight", power_iterations=1):
        super().__init__()
        if int(power_iterations) < 1:
            raise ValueError("power_iterations must be >= 1")          # (the reference's loop would leave sigma from stale u, v)
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        self._pre = None          # (sigma, wv) left by a batched update for the next forward
        if not all(hasattr(module, name + s) for s in self._SUFFIXES):
            self._adopt_weight()

    def _adopt_weight(self):
        """Moves `module.<name>` to `<name>_bar` and adds the power-iteration state: u (one entry per row of the weight
        seen as a matrix [shape[0], rest]) and v (one per column), each one N(0, 1) draw scaled to unit length, u drawn
        before v, registered in the order u, v, bar - after whatever the layer already holds (its bias)."""
        layer, stem = self.module, self.name
        weight = layer._parameters.pop(stem)
        rows = weight.shape[0]
        cols = weight.numel() // rows

        def unit_vector(n):
            t = weight.data.new_empty(n).normal_(0, 1)
            return nn.Parameter(t / (t.norm() + 1e-12), requires_grad=False)

        state = {"_u": unit_vector(rows), "_v": unit_vector(cols), "_bar": nn.Parameter(weight.data)}
        for suffix in self._SUFFIXES:
            layer.register_parameter(stem + suffix, state[suffix])

    # geometry of the wrapped layer in terms of ops.ConvSpec ------------------------------------------------
    def _plan(self, x):
        m = self.module
        w = m.weight_bar
        if isinstance(m, nn.Linear):
            lead = x.shape[:-1]
            x4 = x.reshape(-1, x.shape[-1], 1, 1)
            return x4, w.view(w.shape[0], w.shape[1], 1, 1), ops.ConvSpec("conv", 1, 1, 1, 0, 0), lambda y: y.view(*lead, -1)
        if isinstance(m, nn.Conv1d):
            if m.kernel_size != (1,) or m.stride != (1,) or m.padding != (0,) or m.groups != 1:
                raise NotImplementedError("only Conv1d(kernel_size=1) is on the hot path")
            b, c, n = x.shape
            return (x.view(b, c, 1, n), w.view(w.shape[0], w.shape[1], 1, 1), ops.ConvSpec("conv", 1, 1, 1, 0, 0),
                    lambda y: y.view(b, -1, n))
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if tuple(m.dilation) != (1, 1) or m.stride[0] != m.stride[1]:
                raise NotImplementedError("dilated / anisotropic-stride convs are not on the hot path")
            kh, kw = m.kernel_size
            ph, pw = m.padding
            s = m.stride[0]
            transposed = isinstance(m, nn.ConvTranspose2d)
            if transposed and tuple(m.output_padding) != (0, 0):
                raise NotImplementedError("output_padding")
            if m.groups != 1:
                # the SEPARABLE switch (libs/conv.py:17, libs/attention.py:15-21): depthwise k x k, or one grouped conv whose
                # kernel covers the whole map
                if m.groups == m.in_channels:
                    return x, w, ops.ConvSpec("convT" if transposed else "conv", kh, kw, s, ph, pw, mode="depthwise"), _identity
                if not transposed and (kh, kw) == tuple(x.shape[2:]) and (ph, pw) == (0, 0) and m.groups == m.out_channels:
                    return x, w, ops.ConvSpec("conv", kh, kw, 1, 0, 0, mode="groupdot"), _identity
                raise NotImplementedError("grouped convs other than depthwise / full-size one-output-per-group are not on the hot path")
            if transposed:
                return x, w, ops.ConvSpec("convT", kh, kw, s, ph, pw), _identity
            b, c, h, wd = x.shape
            if s == 1 and ph == 0 and pw == 0 and kw == 1 and kh == h and kh > 1 and x.is_contiguous():
                # full-height (S x 1) conv == 1x1 conv over the merged (channel, row) axis (same memory)
                return x.view(b, c * h, 1, wd), w.view(w.shape[0], c * kh, 1, 1), ops.ConvSpec("conv", 1, 1, 1, 0, 0), _identity
            if s == 1 and ph == 0 and pw == 0 and kh == 1 and h == 1 and kw == wd and kw > 1 and x.is_contiguous():
                # full-width (1 x S) conv on a one-row map == 1x1 conv over the merged (channel, column) axis
                return x.view(b, c * wd, 1, 1), w.view(w.shape[0], c * kw, 1, 1), ops.ConvSpec("conv", 1, 1, 1, 0, 0), _identity
            return x, w, ops.ConvSpec("conv", kh, kw, s, ph, pw), _identity
        raise NotImplementedError("SpectralNorm over %s" % type(m).__name__)

    def take_pre(self):
        """(sigma, wv, guard) of this forward's power iteration: left by a batched update (guard: its ring sets, checked at
        backward time), or run now (guard None)."""
        m = self.module
        pre, self._pre = self._pre, None
        if pre is None:
            for _ in range(self.power_iterations):       # spectral_norm.py:26-29: u, v advance every round, sigma from the last
                pre = ops.sn_power_iteration(m.weight_bar, m.weight_u, m.weight_v)
        return tuple(pre) + (None,) * (3 - len(pre))

    def forward(self, x, out=None, in_link=None):
        """out (Conv2d / ConvTranspose2d only): a channel slice of a concatenation buffer to write the result into.
        in_link: x is the linked activation of the stage's previous conv (forward_act_linked)."""
        m = self.module
        x4, w4, spec, restore = self._plan(x)
        ops.carry_amax(x, x4)          # a reshaped view has its source's largest magnitude
        sigma, wv, guard = self.take_pre()
        y = ops.sn_conv(x4, w4, m.weight_u, m.weight_v, getattr(m, "bias", None), spec, (sigma, wv), self.runtime, guard,
                        out if restore is _identity else None, None, in_link)
        return restore(y)

    def forward_act_linked(self, x, in_link=None):
        """(RootTanh(layer(x)), link) from ONE launch, for the single consumer that takes `link` as its in_link - the next conv of
        the stage (libs/conv.py:19-20, libs/attention.py:44-46): neither direction launches the activation (ops.ActLink).
        None where the layer has no dense contraction (the SEPARABLE switch's grouped convs) or runs on the CPU."""
        m = self.module
        x4, w4, spec, restore = self._plan(x)
        if spec.mode != "dense" or not x4.is_cuda or not ops.ACT_LINKS[0]:
            return None
        geom, out_shape = spec.geometry(tuple(x4.shape), tuple(w4.shape))
        if out_shape[0] * out_shape[1] * out_shape[2] * out_shape[3] > ops.ACT_LINK_MAX_NUMEL[0]:
            return None
        ops.carry_amax(x, x4)
        sigma, wv, guard = self.take_pre()
        link = ops.ActLink()
        _, second = ops.sn_conv(x4, w4, m.weight_u, m.weight_v, getattr(m, "bias", None), spec, (sigma, wv), self.runtime, guard,
                                None, {"link": link}, in_link)
        return restore(second), link

    def forward_activated(self, x, latent=None):
        """(layer(x), second) with second = RootTanh(layer(x)) - or, with `latent`, cat([latent, RootTanh(layer(x))], 1), the next
        style link's input - from ONE launch where the layer runs on a 1x1 map (ops.act_epilogue_ok); None if it does not."""
        m = self.module
        x4, w4, spec, restore = self._plan(x)
        if not ops.act_epilogue_ok(spec, x4.shape) or not x4.is_cuda:
            return None
        sigma, wv, guard = self.take_pre()
        y, second = ops.sn_conv(x4, w4, m.weight_u, m.weight_v, getattr(m, "bias", None), spec, (sigma, wv), self.runtime, guard,
                                None, {"latent": latent})
        return restore(y), (second if latent is not None else restore(second))


class _TwoBranch(nn.Module):
    """A skip branch and a layer branch over the same input (optionally a different input and a style scale for the layer
    branch); subclasses say how the two results merge.  Attribute names are the reference's (they are state_dict keys)."""

    def __init__(self, residual_module, layer_module):
        super().__init__()
        self.residual_module = residual_module
        self.layer_module = layer_module

    def _run(self, function_input, layer_input, scale):
        feed = function_input if layer_input is None else layer_input
        if layer_input is None and self.residual_module is _identity and isinstance(self.layer_module, (Norm, ActivatedBaseConv)):
            # one tensor, two consumers of this package (the merge itself and the branch's leading norm / RootTanh): their
            # backward kernels share one gradient buffer instead of leaving an add to autograd (ops.fork)
            function_input, feed = ops.fork(function_input)
        stages = list(self.residual_module) if isinstance(self.residual_module, nn.Sequential) else [self.residual_module]
        if layer_input is None and isinstance(stages[0], CatModule) and stages[0].in_place(function_input):
            # the discriminator's stem: its input (the generated image in the G-step) feeds the identity half of the skip branch's
            # concatenation, that branch's 1x1 conv and the conv branch - the three gradients are summed by one launch (ops.fork3)
            skip, conv_alias, feed = ops.fork3(function_input)
            for i, stage in enumerate(stages):
                skip = stage(skip, conv_alias=conv_alias) if (i == 0 and conv_alias is not skip) else stage(skip)
        else:
            skip = self.residual_module(function_input)
        branch = self.layer_module(feed) if scale is None else self.layer_module(feed, scale)
        return skip, branch


class CatModule(_TwoBranch):
    """Channel concatenation [skip, layer] (libs/merge.py:4-16).  Where the layer is a spectral-normalised conv on the same
    input (every use in the two networks: libs/scale.py:28-34) it writes its result straight into its slice of the
    concatenation, and only the skip part is copied."""

    def in_place(self, x):
        """whether forward(x) takes the write-in-place path (the conv's result lands in its slice of the concatenation)"""
        layer = self.layer_module
        if not (self.residual_module is _identity and isinstance(layer, SpectralNorm) and isinstance(layer.module, nn.Conv2d)
                and x.dim() == 4 and x.is_cuda):
            return False
        m = layer.module
        return (m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0) and m.groups == 1 and x.shape[1] == m.in_channels)

    def forward(self, function_input, layer_input=None, scale=None, conv_alias=None):
        """conv_alias: a second autograd alias of function_input for the conv (ops.fork3: the block sums the gradients of its
        input's three consumers in one launch); only with the in-place path."""
        if layer_input is None and scale is None and self.in_place(function_input):
            x = function_input
            m = self.layer_module.module
            buf = x.new_empty((x.shape[0], x.shape[1] + m.out_channels) + tuple(x.shape[2:]))
            branch = self.layer_module(x if conv_alias is None else conv_alias, out=buf[:, x.shape[1]:])
            return ops.cat_channels(x, branch, [buf])
        if conv_alias is not None:
            raise RuntimeError("CatModule: conv_alias needs the in-place path")
        return ops.cat_channels(*self._run(function_input, layer_input, scale))


class ResModule(_TwoBranch, _Bound):
    """out = (gamma * layer(x) + 1) * skip(x) with one scalar gamma, initialised to (+-1) + m + 1 (the sign is the
    orthogonal initialisation of a 1 x 1 matrix: one RNG draw; libs/merge.py:46-62)."""

    def __init__(self, residual_module, layer_module, m=0):
        super().__init__(residual_module, layer_module)
        gamma = torch.ones((1, 1))
        nn.init.orthogonal_(gamma)
        self.gamma = nn.Parameter(gamma + (m + 1))

    def forward(self, function_input, layer_input=None, scale=None):
        skip, branch = self._run(function_input, layer_input, scale)
        compact = getattr(branch, "_locate_compact", None)     # un-expanded [B, C, 1, 1] source (see Expand)
        return ops.residual_gate(skip, branch if compact is None else compact, self.gamma, self.runtime)


class FeaturePooling(nn.Module):
    def __init__(self, out_features):
        super().__init__()
        self.out_features = out_features

    def forward(self, function_input):
        return ops.feature_pool(function_input, self.out_features)


class Upsample2x(nn.Module):
    """nn.Upsample(mode='bilinear', scale_factor=2, align_corners=False)."""

    def forward(self, x):
        return ops.upsample2x(x)


class AvgPool2(nn.Module):
    """nn.AvgPool2d(2, 2)."""

    def forward(self, x):
        return ops.avgpool2(x)


def Scale(in_features, out_features, stride, transpose, dim=2):
    """Skip branch: channel change, then resample (libs/scale.py:19-45).  Same container structure as the
    reference (a bare layer, an nn.Sequential or the identity) so that parameter names match."""
    if dim != 2:
        raise NotImplementedError("dim = 2 only")
    layers = []
    if in_features > out_features:
        if in_features % out_features == 0:
            layers.append(FeaturePooling(out_features))
        else:
            layers.append(SpectralNorm(nn.Conv2d(in_features, out_features, 1)))
    elif out_features > in_features:
        layers.append(CatModule(_identity, SpectralNorm(nn.Conv2d(in_features, out_features - in_features, 1))))
    if stride > 1:
        if stride != 2:
            raise NotImplementedError("stride 2 only")
        layers.append(Upsample2x() if transpose else AvgPool2())
    if len(layers) > 1:
        return nn.Sequential(*layers)
    if not layers:
        return _identity
    return layers[0]


class Expand(nn.Module):
    """view(B, -1, 1, ..) + expand(target) (stride-0 broadcast).  The compact source rides along so that the
    residual gate can consume it without materialising the expansion."""

    def __init__(self, *target_size):
        super().__init__()
        self.target_size = target_size

    def forward(self, function_input):
        if function_input is None:
            return None
        compact = function_input.view(function_input.size(0), -1, *[1] * (function_input.dim() - 2))
        out = compact.expand(self.target_size)
        out._locate_compact = compact
        return out


class ChannelSoftmax(nn.Module):
    """nn.Softmax(dim=1) on a [B, C, 1, 1] map."""

    def forward(self, x):
        b, c = x.shape[:2]
        if x.numel() != b * c:
            raise NotImplementedError("channel softmax is only used on [B, C, 1, 1] maps")
        return ops.softmax_lastdim(x.reshape(b, c)).view_as(x)


def feature_attention(in_size, features, dim=2, cfg=None):
    """Channel gate of a block (libs/attention.py:9-37): squeeze the S x S map to one value per bottleneck channel, expand
    back to `features` channels, softmax over the channels, broadcast over the map.  Built from `arch.squeeze_plan` - a
    list of conv descriptions - as an nn.Sequential whose positions are the reference's (they are state_dict keys)."""
    cfg = cfg or get_default()
    if dim != 2:
        raise NotImplementedError("dim = 2 only")
    parts = []
    for conv in arch.squeeze_plan(in_size, features, cfg):
        parts.append(SpectralNorm(nn.Conv2d(conv.cin, conv.cout, kernel_size=conv.kernel, bias=False, groups=conv.groups)))
        if conv.activated:
            parts.append(NonLinear())
    parts += [ChannelSoftmax(), Expand(-1, features, *([in_size] * dim))]
    return _GateSequence(*parts)


class _GateSequence(nn.Sequential):
    """nn.Sequential (same positions = same state_dict keys) that runs a spectral-normalised conv on a 1x1 map together with
    the RootTanh behind it as one launch (SpectralNorm.forward_activated)."""

    def forward(self, x):
        parts = list(self)
        i = 0
        link = None          # x is the linked activation of the previous conv
        while i < len(parts):
            part = parts[i]
            if isinstance(part, SpectralNorm) and i + 1 < len(parts) and isinstance(parts[i + 1], RootTanhModule) and x.is_cuda:
                fused = part.forward_act_linked(x, in_link=link)
                if fused is None:
                    fused = part.forward_activated(x)
                    fused = None if fused is None else (fused[1], None)
                if fused is not None:
                    x, link = fused
                    i += 2
                    continue
            x = part(x, in_link=link) if (link is not None and isinstance(part, SpectralNorm)) else part(x)
            link = None
            i += 1
        return x


class SelfAttention(nn.Module):
    """Position gate: softmax over the N = H*W positions of conv1x1(RootTanh(conv1x1(x))), per (sample, channel) - not a
    QK^T attention (libs/attention.py:40-54)."""

    def __init__(self, features):
        super().__init__()
        self.conv_0 = SpectralNorm(nn.Conv1d(features, features, 1, bias=False))
        self.nlin_0 = NonLinear()
        self.conv_1 = SpectralNorm(nn.Conv1d(features, features, 1, bias=False))

    def forward(self, function_input):
        shape = function_input.shape
        rows = function_input.reshape(shape[0], shape[1], -1)          # [B, C, N]
        ops.carry_amax(function_input, rows)
        fused = self.conv_0.forward_act_linked(rows) if rows.is_cuda else None
        if fused is not None:
            rows = self.conv_1(fused[0], in_link=fused[1])
        else:
            for part in (self.conv_0, self.nlin_0, self.conv_1):
                rows = part(rows)
        return ops.softmax_lastdim(rows).view(shape)


class ActivatedBaseConv(nn.Module):
    starts_with_activation = True

    def __init__(self, in_features, out_features, conv, kernel=5, stride=1, pad=2, cfg=None):
        super().__init__()
        cfg = cfg or get_default()
        mid = in_features * cfg.feature_multiplier
        self.conv_0 = SpectralNorm(conv(in_channels=in_features, kernel_size=kernel, stride=stride, padding=pad, bias=False,
                                        out_channels=mid, groups=in_features if cfg.separable else 1))
        self.conv_1 = SpectralNorm(conv(kernel_size=1, stride=1, padding=0, out_channels=out_features, bias=False,
                                        in_channels=mid))

    def forward(self, function_input, pre_activated=False):
        h = function_input if pre_activated else ops.root_tanh(function_input)
        fused = self.conv_0.forward_act_linked(h) if h.is_cuda else None
        if fused is not None:          # conv_0's launch writes the activation, conv_1's input gradient multiplies by its derivative
            return self.conv_1(fused[0], in_link=fused[1])
        return self.conv_1(ops.root_tanh(self.conv_0(h)))


class DeepResidualConv(nn.Module):
    """A chain of `depth` ActivatedBaseConv stages (libs/conv.py:27-72), built from `arch.conv_chain`: stage 0 carries the
    stride / transposition with kernel 2*stride + (0 if transposed else 1) and maps to the bottleneck width when
    depth > 1; the later stages are 5x5, wrapped in Norm from the second one on and in a ResModule(m=1) whenever their
    widths agree."""
    starts_with_activation = True

    def __init__(self, in_features, out_features, transpose, stride, use_bottleneck=True, dim=2, depth=1, cfg=None):
        super().__init__()
        if dim != 2:
            raise NotImplementedError("dim = 2 only")
        cfg = cfg or get_default()
        self.layers = []
        for link in arch.conv_chain(in_features, out_features, transpose, stride, use_bottleneck, depth, cfg):
            conv_cls = nn.ConvTranspose2d if link.transposed else nn.Conv2d
            layer = ActivatedBaseConv(link.cin, link.cout, conv_cls, kernel=link.kernel, stride=link.stride, pad=link.pad, cfg=cfg)
            if link.normalized:
                layer = Norm(link.cin, layer, dim)
            if link.residual:
                layer = ResModule(_identity, layer, m=1)
            setattr(self, "conv_%d" % len(self.layers), layer)
            self.layers.append(layer)

    def forward(self, function_input, pre_activated=False):
        first, *rest = self.layers
        out = first(function_input, pre_activated=pre_activated)
        for layer in rest:
            out = layer(out)
        return out


class LinearModule(nn.Module):
    def __init__(self, *args):
        super().__init__()
        self.module = SpectralNorm(nn.Linear(*args))
        self.nlin = NonLinear()

    def forward(self, function_input):
        out = self.module(function_input)
        return self.nlin(out), out

    def pre_activation(self, function_input):
        """The linear alone (the chain's last link: nothing consumes its activation)."""
        return self.module(function_input)

    def pre_and_next_input(self, function_input, latent):
        """(pre-activation, cat([latent, RootTanh(pre-activation)], 1)): the link's norm scale and the NEXT link's input
        (libs/block.py:119-125) from one launch."""
        fused = self.module.forward_activated(function_input, latent)
        if fused is None:
            pre = self.module(function_input)
            return pre, ops.act_cat(latent, pre)
        return fused


class Block(nn.Module):
    """One up / down stage (libs/block.py:15-52): a gated conv branch over the resampled skip branch and, on attention
    stages, a channel gate and a position gate on the result.  Gate k of the stage takes style scale k."""

    def __init__(self, in_size, in_features, out_features, stride, transpose, block_number, cat_out=True, dim=2, cfg=None):
        super().__init__()
        cfg = cfg or get_default()
        self.scale_layer = Scale(in_features, out_features, stride, transpose, dim=dim)
        conv = DeepResidualConv(in_features, out_features, transpose, stride, depth=cfg.depth, dim=dim, cfg=cfg)
        self.res_module_i = ResModule(_identity, Norm(in_features, conv, dim=dim), m=3)
        self.attention = arch.stage_has_attention(in_size, block_number, cfg)
        self._gates = []
        if self.attention:
            self.res_module_f = ResModule(_identity, Norm(out_features, feature_attention(in_size, out_features, dim=dim, cfg=cfg),
                                                          dim=dim))
            self.res_module_s = ResModule(_identity, Norm(out_features, SelfAttention(out_features), dim=dim))
            self._gates = [self.res_module_f, self.res_module_s]
        self.cat_out = cat_out

    def forward(self, function_input, scales=None):
        scales = list(scales) if scales is not None else []
        scales += [None] * (1 + len(self._gates) - len(scales))
        skip_in = conv_in = function_input
        stages = list(self.scale_layer) if isinstance(self.scale_layer, nn.Sequential) else [self.scale_layer]
        # (Pooling FIRST on the down-sampling skip branch - the 2x2 mean commutes with the concatenation and the 1x1 conv - was
        # measured at -0.16 ms per step, but it changes the rounding order enough to move the ill-conditioned d(gamma) sums of
        # the 128x128 record from 5e-4 to 1e-3 of their reference values: the reference's order stays.)
        conv_alias = None
        if isinstance(stages[0], (FeaturePooling, AvgPool2)):
            skip_in, conv_in = ops.fork(function_input)      # the pooling and the conv branch's norm share the gradient buffer
        elif isinstance(stages[0], CatModule) and stages[0].in_place(function_input):
            # three consumers (identity half of the concatenation, its 1x1 conv, the conv branch's norm): one summing launch
            skip_in, conv_alias, conv_in = ops.fork3(function_input)
        skip = skip_in
        if (len(stages) == 2 and isinstance(stages[0], FeaturePooling) and isinstance(stages[1], Upsample2x)
                and ops.pool_upsample_ok(skip, stages[0].out_features)):
            skip = ops.pool_upsample(skip, stages[0].out_features)      # one launch, the pooled map is never stored
        else:
            for i, stage in enumerate(stages):
                skip = stage(skip, conv_alias=conv_alias) if (i == 0 and conv_alias is not None) else stage(skip)
        out = self.res_module_i(skip, conv_in, scales[0])
        for gate, scale in zip(self._gates, scales[1:]):
            out = gate(out, scale=scale)
        return out


class BlockBlock(nn.Module):
    """A stack of Blocks and - in the generator (`mul_channel`) - the style chain that feeds their norms (libs/block.py:55-127).
    The stack is described first (`arch.stack_plan`: one `Stage` per block, with the (in, out) widths of its style linears);
    all blocks are built before any style linear, which fixes the RNG draw order."""

    def __init__(self, block_count, in_size, features, strides, transpose, mul_channel=False, dim=2, cfg=None):
        super().__init__()
        cfg = cfg or get_default()
        self.block_count = block_count
        self.plan = arch.stack_plan(block_count, in_size, features, strides, transpose, mul_channel, cfg)
        self.blocks = []
        for st in self.plan:
            block = Block(st.side, st.cin, st.cout, st.stride, transpose, st.index, dim=dim, cfg=cfg)
            setattr(self, "block_%d" % st.index, block)
            self.blocks.append(block)
        # style chain: flat list of linears, `first_style[i]` = position of stage i's first one
        self.mul_blocks = []
        self.first_style = []
        for st in self.plan:
            self.first_style.append(len(self.mul_blocks))
            for fan_in, fan_out in st.style:
                linear = LinearModule(fan_in, fan_out)
                setattr(self, "mul_block_%d" % len(self.mul_blocks), linear)
                self.mul_blocks.append(linear)
        self.depths = [len(st.style) for st in self.plan] if mul_channel else []
        self.sums = self.first_style + [len(self.mul_blocks)] if mul_channel else [0]
        self.out_features = features[block_count]

    def _style_scales(self, stage, latent, carry, last_stage):
        """Runs stage `stage`'s style linears: each sees [latent, previous activated output] (`carry`, written by the previous
        link's own launch) and yields its pre-activation as a [B, C, 1, 1] norm scale plus the next link's input."""
        scales = []
        start = self.first_style[stage]
        links = self.mul_blocks[start:start + len(self.plan[stage].style)]
        for k, linear in enumerate(links):
            inp = latent if carry is None else carry
            if last_stage and k == len(links) - 1:
                pre, carry = linear.pre_activation(inp), None
            else:
                pre, carry = linear.pre_and_next_input(inp, latent)
            scales.append(pre.view(*pre.shape, 1, 1))
        return scales, carry

    def forward(self, function_input, noise=None):
        out, carry = function_input, None
        if noise is None:
            for block in self.blocks:
                out = block(out, scales=None)
            return out
        # the whole style chain first (it depends on the latent alone)
        per_stage = []
        last = max((i for i, st in enumerate(self.plan) if st.style), default=-1)
        for stage in range(len(self.blocks)):
            scales, carry = self._style_scales(stage, noise, carry, stage == last)
            per_stage.append(scales)
        for block, scales in zip(self.blocks, per_stage):
            out = block(out, scales=scales)
        return out
