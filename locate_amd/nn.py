"""Host-side mirror of the reference's layer interface (same class names, constructor arguments, forward
signatures, parameter names and therefore the same `state_dict()` key layout), computing on MI355X through
`locate_amd.ops`.  torch.nn.Conv2d / ConvTranspose2d / Conv1d / Linear objects are used only as parameter
containers (shape, default initialisation, hyper-parameters): their own forward is never called.

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

from . import ops
from .config import get_default


def _identity(x):
    return x


class RootTanhModule(nn.Module):
    def forward(self, function_input):
        return ops.root_tanh(function_input)


NonLinear = RootTanhModule


class InPlaceNorm(nn.Module):
    """Global-statistics normalisation + per-channel affine (or per-sample style scale)."""

    def __init__(self, features=1, dim=2):
        super().__init__()
        self.weight = nn.Parameter(torch.ones((1, features, *[1] * dim)))
        self.bias = nn.Parameter(torch.zeros((1, features, *[1] * dim)))

    def forward(self, function_input, scale=None, with_act=False):
        return ops.inplace_norm(function_input, self.weight if scale is None else scale, self.bias, with_act)


class Norm(nn.Module):
    def __init__(self, features, module, dim=2):
        super().__init__()
        self.i_norm = InPlaceNorm(features, dim=dim)
        self.module = module

    def forward(self, function_input, scale=None):
        if getattr(self.module, "starts_with_activation", False):
            # norm + the wrapped conv stage's leading RootTanh in one kernel
            return self.module(self.i_norm(function_input, scale, with_act=True), pre_activated=True)
        return self.module(self.i_norm(function_input, scale))


class SpectralNorm(nn.Module):
    """Wraps a torch.nn conv / linear: every forward runs one power iteration on (weight_u, weight_v) in place
    and applies the layer with weight_bar / sigma.  Parameters live on the wrapped module under the reference's
    names (`module.weight_bar|weight_u|weight_v`)."""

    def __init__(self, module, name="weight", power_iterations=1):
        super().__init__()
        if power_iterations != 1:
            raise NotImplementedError("power_iterations = 1 is the only value the reference uses")
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        self._pre = None          # (sigma, wv) left by a batched update for the next forward
        if not self._made_params():
            self._make_params()

    def _made_params(self):
        return all(hasattr(self.module, self.name + s) for s in ("_u", "_v", "_bar"))

    def _make_params(self):
        w = getattr(self.module, self.name)
        height = w.data.shape[0]
        width = w.data.numel() // height
        u = nn.Parameter(w.data.new(height).normal_(0, 1), requires_grad=False)
        v = nn.Parameter(w.data.new(width).normal_(0, 1), requires_grad=False)
        u.data = u.data / (u.data.norm() + 1e-12)
        v.data = v.data / (v.data.norm() + 1e-12)
        w_bar = nn.Parameter(w.data)
        del self.module._parameters[self.name]
        self.module.register_parameter(self.name + "_u", u)
        self.module.register_parameter(self.name + "_v", v)
        self.module.register_parameter(self.name + "_bar", w_bar)

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
        """(sigma, wv) of this forward's power iteration: left by a batched update, or run now."""
        m = self.module
        pre, self._pre = self._pre, None
        if pre is None:
            pre = ops.sn_power_iteration(m.weight_bar, m.weight_u, m.weight_v)
        return pre

    def forward(self, x):
        m = self.module
        x4, w4, spec, restore = self._plan(x)
        y = ops.sn_conv(x4, w4, m.weight_u, m.weight_v, getattr(m, "bias", None), spec, self.take_pre())
        return restore(y)


class CatModule(nn.Module):
    def __init__(self, residual_module, layer_module):
        super().__init__()
        self.residual_module = residual_module
        self.layer_module = layer_module

    def forward(self, function_input, layer_input=None, scale=None):
        args = [function_input if layer_input is None else layer_input]
        if scale is not None:
            args.append(scale)
        return ops.cat_channels(self.residual_module(function_input), self.layer_module(*args))


class ResModule(nn.Module):
    """out = (gamma * layer(x) + 1) * residual(x) with a scalar gamma = (+-1) + m + 1."""

    def __init__(self, residual_module, layer_module, m=0):
        super().__init__()
        self.residual_module = residual_module
        self.layer_module = layer_module
        self.gamma = nn.Parameter(torch.ones((1, 1)))
        nn.init.orthogonal_(self.gamma.data)
        self.gamma.data.add_(m + 1)

    def forward(self, function_input, layer_input=None, scale=None):
        args = [function_input if layer_input is None else layer_input]
        if scale is not None:
            args.append(scale)
        res = self.residual_module(function_input)
        layer_out = self.layer_module(*args)
        compact = getattr(layer_out, "_locate_compact", None)   # un-expanded [B, C, 1, 1] source (see Expand)
        return ops.residual_gate(res, layer_out if compact is None else compact, self.gamma)


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
    cfg = cfg or get_default()
    if dim != 2:
        raise NotImplementedError("dim = 2 only")
    bfeatures = features // cfg.bottleneck
    layers = []
    input_features = features
    min_features = min(input_features, bfeatures)
    if cfg.separable and input_features % min_features == 0 and bfeatures % min_features == 0:
        # one grouped full-size conv instead of the (S x 1), (1 x S) pair - and no RootTanh after it (attention.py:15-21)
        layers.append(SpectralNorm(nn.Conv2d(input_features, bfeatures, kernel_size=[in_size] * dim, bias=False,
                                             groups=min_features)))
    else:
        for i in range(dim):
            kernel_size = [1] * dim
            kernel_size[i] = in_size
            layers.extend([SpectralNorm(nn.Conv2d(input_features, bfeatures, kernel_size=kernel_size, bias=False)), NonLinear()])
            input_features = bfeatures
    layers.extend([SpectralNorm(nn.Conv2d(bfeatures, features, kernel_size=1, bias=False)), ChannelSoftmax(),
                   Expand(-1, features, *([in_size] * dim))])
    return nn.Sequential(*layers)


class SelfAttention(nn.Module):
    """softmax over the N = H*W positions of conv1x1(RootTanh(conv1x1(x))) - a gate, not a QK^T attention."""

    def __init__(self, features):
        super().__init__()
        self.conv_0 = SpectralNorm(nn.Conv1d(features, features, 1, bias=False))
        self.nlin_0 = NonLinear()
        self.conv_1 = SpectralNorm(nn.Conv1d(features, features, 1, bias=False))

    def forward(self, function_input):
        batch, features, *size = function_input.size()
        out = function_input.reshape(batch, features, -1)
        out = ops.softmax_lastdim(self.conv_1(self.nlin_0(self.conv_0(out))))
        return out.view(batch, features, *size)


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
        return self.conv_1(ops.root_tanh(self.conv_0(h)))


class DeepResidualConv(nn.Module):
    """A chain of `depth` ActivatedBaseConv stages (libs/conv.py:27-72).  Stage 0 carries the stride / transposition with
    kernel 2*stride + (0 if transposed else 1) and maps to the bottleneck width when depth > 1; the depth - 2 middle
    stages (5x5, bottleneck -> bottleneck) and the last one (5x5, bottleneck -> out) are wrapped in Norm from the second
    one on and in a ResModule(m=1) whenever their widths agree.  (The reference passes its conv class in the `residual`
    slot of its helper - a truthy value - so every stage after the first is residual.)"""
    starts_with_activation = True

    def __init__(self, in_features, out_features, transpose, stride, use_bottleneck=True, dim=2, depth=1, cfg=None):
        super().__init__()
        if dim != 2:
            raise NotImplementedError("dim = 2 only")
        cfg = cfg or get_default()
        min_features = min(in_features, out_features)
        if use_bottleneck and max(in_features, out_features) // min_features < cfg.bottleneck:
            min_features //= cfg.bottleneck
        if depth > 1 and min_features < 1:
            raise ValueError("DeepResidualConv(%d -> %d, depth %d): the bottleneck width is 0" % (in_features, out_features, depth))
        kernel = stride * 2 + int(not transpose)
        pad = max(kernel // 2 - stride // 2, 0) if transpose else kernel // 2
        self.layers = []

        def add_conv(cin, cout, residual, normalize, conv=nn.Conv2d, **kw):
            layer = ActivatedBaseConv(cin, cout, conv, cfg=cfg, **kw)
            if normalize:
                layer = Norm(cin, layer, dim)
            if residual and cin == cout:
                layer = ResModule(_identity, layer, m=1)
            setattr(self, "conv_%d" % len(self.layers), layer)
            self.layers.append(layer)

        add_conv(in_features, min_features if depth > 1 else out_features, False, False,
                 conv=nn.ConvTranspose2d if transpose else nn.Conv2d, kernel=kernel, stride=stride, pad=pad)
        for i in range(depth - 2):
            add_conv(min_features, min_features, True, bool(i))
        if depth > 1:
            add_conv(min_features, out_features, True, bool(depth - 2))

    def forward(self, function_input, pre_activated=False):
        out = self.layers[0](function_input, pre_activated=pre_activated)
        for layer in self.layers[1:]:
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


class Block(nn.Module):
    def __init__(self, in_size, in_features, out_features, stride, transpose, block_number, cat_out=True, dim=2, cfg=None):
        super().__init__()
        cfg = cfg or get_default()
        self.scale_layer = Scale(in_features, out_features, stride, transpose, dim=dim)
        self.res_module_i = ResModule(_identity, Norm(in_features, DeepResidualConv(in_features, out_features, transpose, stride,
                                                                                   depth=cfg.depth, dim=dim, cfg=cfg), dim=dim),
                                      m=3)
        self.attention = bool(in_size >= cfg.min_attention_size and block_number % cfg.attention_every_nth_layer == 0)
        if self.attention:
            self.res_module_f = ResModule(_identity, Norm(out_features, feature_attention(in_size, out_features, dim=dim, cfg=cfg),
                                                          dim=dim))
            self.res_module_s = ResModule(_identity, Norm(out_features, SelfAttention(out_features), dim=dim))
        self.cat_out = cat_out

    def forward(self, function_input, scales=None):
        if scales is None:
            scales = [None] * 4
        scaled = self.scale_layer(function_input)
        out = self.res_module_i(scaled, function_input, scales[0])
        if self.attention:
            out = self.res_module_f(out, scale=scales[1])
            out = self.res_module_s(out, scale=scales[2])
        return out


class BlockBlock(nn.Module):
    def __init__(self, block_count, in_size, features, strides, transpose, mul_channel=False, dim=2, cfg=None):
        super().__init__()
        cfg = cfg or get_default()
        self.block_count = block_count
        z = cfg.input_vector_z
        size = float(in_size)
        blocks = []
        for i in range(block_count):
            size = size * strides[i] if transpose else size / strides[i]
            blocks.append(Block(int(size + 1 - 1e-12), features[i], features[i + 1], strides[i], transpose, i, dim=dim, cfg=cfg))
        self.blocks = blocks
        for i, block in enumerate(blocks):
            setattr(self, "block_%d" % i, block)
        sums, depths = [0], []
        if mul_channel:
            mul_blocks = []
            prev_out = 0
            for i in range(block_count):
                extra = 2 * int(blocks[i].attention)
                depths.append(1 + extra)
                sums.append(sums[-1] + extra + 1)
                inp, out = features[i], features[i + 1]
                group_inp = prev_out if (prev_out and prev_out != inp) else inp
                mul_blocks.append(LinearModule(group_inp + z * bool(i), inp))
                if extra:
                    mul_blocks.append(LinearModule(inp + z, out))
                    mul_blocks.extend(LinearModule(out + z, out) for _ in range(1, extra))
                    prev_out = out
                else:
                    prev_out = inp
            self.mul_blocks = mul_blocks
            for i, block in enumerate(mul_blocks):
                setattr(self, "mul_block_%d" % i, block)
        self.depths = depths
        self.sums = sums
        self.out_features = features[block_count]

    def forward(self, function_input, noise=None):
        chain = None
        for i in range(self.block_count):
            operand = None
            if noise is not None:
                operand = []
                for idx in range(self.depths[i]):
                    chain = noise if chain is None else ops.cat_channels(noise, chain)
                    chain, factor = self.mul_blocks[self.sums[i] + idx](chain)
                    operand.append(factor.view(*factor.size(), 1, 1))
            function_input = self.blocks[i](function_input, scales=operand)
        return function_input
