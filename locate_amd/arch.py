"""Architecture DESCRIPTIONS: plain data computed from a NetConfig, from which `locate_amd.nn` / `locate_amd.models` build
the modules.  Nothing here touches torch or the RNG - a description can be printed, compared and tested on its own
(tests/test_host_models.py checks the tables against SURVEY.md Appendix A and the reference's state_dict shapes).

Behaviour described (citations into the reference):
    widths of the two networks              libs/models.py:12-31,43-52,76-78
    one conv stage / chain of a block       libs/conv.py:27-67
    the channel gate's squeeze convs        libs/attention.py:9-37
    which stages carry attention            libs/block.py:28-29
    sizes along a stack, the style chain    libs/block.py:61-106
"""
from collections import namedtuple

ConvLink = namedtuple("ConvLink", "cin cout kernel stride pad transposed normalized residual")
SqueezeConv = namedtuple("SqueezeConv", "cin cout kernel groups activated")
Stage = namedtuple("Stage", "index cin cout side stride attention style")


def _mult4(n):
    return n // 4 * 4


def generator_widths(cfg):
    """[Z, w_{n-1}, ..., w_0], w_k = GEN_FEATURES * FACTOR^(k - n) rounded down to a multiple of 4, n = LAYERS - 1 stages:
    the generator starts from the latent width and halves its width per stage down to 3 * BASE_FEATURE_FACTOR * 2."""
    n = cfg.layers - 1
    return [cfg.input_vector_z] + [_mult4(int(cfg.gen_features * cfg.factor ** (k - n))) for k in reversed(range(n))]


def discriminator_widths(cfg):
    """[w_0, ..., w_{n-1}, w_{n-1}], w_k = DIS_FEATURES * FACTOR^(k + 1 - n) rounded down to a multiple of 4; the last stage
    keeps its width."""
    n = cfg.layers - 1
    widths = [_mult4(int(cfg.dis_features * cfg.factor ** (k + 1 - n))) for k in range(n)]
    return widths + widths[-1:]


def conv_chain(in_features, out_features, transpose, stride, use_bottleneck, depth, cfg):
    """The `depth` conv stages of one DeepResidualConv.  Link 0 resamples (kernel 2 * stride, +1 unless transposed); with
    depth > 1 it maps to the bottleneck width and 5x5 stride-1 links follow: depth - 2 of bottleneck -> bottleneck, then
    bottleneck -> out.  Links from the third on are preceded by a norm; every link after the first whose two widths agree is
    residual."""
    narrow = min(in_features, out_features)
    if use_bottleneck and max(in_features, out_features) // narrow < cfg.bottleneck:
        narrow //= cfg.bottleneck
    if depth > 1 and narrow < 1:
        raise ValueError("DeepResidualConv(%d -> %d, depth %d): the bottleneck width is 0" % (in_features, out_features, depth))
    kernel = 2 * stride + (0 if transpose else 1)
    pad = max(kernel // 2 - stride // 2, 0) if transpose else kernel // 2
    widths = [in_features] + [narrow] * (depth - 1) + [out_features]
    chain = [ConvLink(widths[0], widths[1], kernel, stride, pad, bool(transpose), False, False)]
    for pos in range(1, depth):
        cin, cout = widths[pos], widths[pos + 1]
        chain.append(ConvLink(cin, cout, 5, 1, 2, False, pos >= 2, cin == cout))
    return chain


def squeeze_plan(side, features, cfg):
    """Convs of the channel gate on a side x side map: (side x 1) then (1 x side), each followed by RootTanh, to
    features / BOTTLENECK channels - or, with SEPARABLE (when the widths divide), one grouped full-map conv without an
    activation - then a 1x1 back to `features`."""
    squeezed = features // cfg.bottleneck
    narrow = min(features, squeezed)
    if cfg.separable and features % narrow == 0 and squeezed % narrow == 0:
        plan = [SqueezeConv(features, squeezed, (side, side), narrow, False)]
    else:
        plan = [SqueezeConv(features, squeezed, (side, 1), 1, True), SqueezeConv(squeezed, squeezed, (1, side), 1, True)]
    plan.append(SqueezeConv(squeezed, features, (1, 1), 1, False))
    return plan


def stage_has_attention(side, index, cfg):
    return bool(side >= cfg.min_attention_size and index % cfg.attention_every_nth_layer == 0)


def stack_plan(count, in_side, features, strides, transpose, styled, cfg):
    """One Stage per block of a stack: widths, output side, attention flag and - for a styled (generator) stack - the
    (fan_in, fan_out) of the stage's style linears.  A stage needs one style scale for its conv branch's norm (width = its
    input width) and, with attention, one for each gate's norm (width = its output width).  The linears form ONE chain
    across all stages: each but the very first sees [latent, previous link's output]."""
    z = cfg.input_vector_z
    side = float(in_side)
    carry = 0                      # width of the previous style linear's output
    plan = []
    for i in range(count):
        side = side * strides[i] if transpose else side / strides[i]
        out_side = int(side + 1 - 1e-12)                      # ceil for the fractional sides of tiny inputs
        cin, cout = features[i], features[i + 1]
        attention = stage_has_attention(out_side, i, cfg)
        style = []
        if styled:
            for pos, width in enumerate([cin] + [cout] * (2 if attention else 0)):
                if pos == 0:
                    fan_in = (carry or cin) + (z if i else 0)
                else:
                    fan_in = carry + z
                style.append((fan_in, width))
                carry = width
        plan.append(Stage(i, cin, cout, out_side, strides[i], attention, tuple(style)))
    return plan
