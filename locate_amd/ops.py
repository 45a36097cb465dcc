"""Operator layer: torch.autograd.Functions whose forward AND backward are the gfx950 kernels behind the C ABI.

Mirrors the reference's operator-level boundary: `root_tanh` (libs/activation.py:39),
`mean_sub_mul_div_add`/InPlaceNorm (libs/inplace_norm.py:31), `residual_function` (libs/merge.py:43) and the
spectral-normalised convolutions (libs/spectral_norm.py:57-59 + the wrapped torch.nn conv/linear forward).
PyTorch is used for device memory, streams and the autograd graph only - no ATen arithmetic on the hot path.
"""
import ctypes
import os

import torch

from ._lib import check, lib

_IntArr12 = ctypes.c_int * 12


class _ActEpilogue(ctypes.Structure):
    """LocateActEpilogue of include/locate_hip.h: the activated second output of locate_conv_fwd / the multiplication by the
    activation's derivative in locate_conv_dgrad."""
    _fields_ = [("act_out", ctypes.c_void_p), ("act_bs", ctypes.c_int64), ("lat", ctypes.c_void_p), ("lat_bs", ctypes.c_int64),
                ("lat_z", ctypes.c_int32), ("pad", ctypes.c_int32), ("mul_pre", ctypes.c_void_p), ("mul_bs", ctypes.c_int64),
                ("out_absmax", ctypes.c_void_p)]


ACT_LINKS = [os.environ.get("LOCATE_ACT_LINKS", "1") != "0"]      # the fused activations of a stage (ActLink); off: launches of their own
# ... up to this many activated elements: RootTanh is ~30 vector instructions per element, which a contraction's epilogue runs at
# the contraction's occupancy (two or three blocks per CU) - on the large maps that costs what the separate launch and its
# extra pass over the tensor cost (profiles/notes_r04_experiments.md section 6); on the small ones the launch is the cost
# (same-box A/B at config 2, bench.py --step-only, two rounds each: off 9.56 / 9.55, up to 0.3 M elements 9.49 / 9.50, 1 M 9.51 /
# 9.51, 4 M 9.48 / 9.51, 8 M 9.50 / 9.50, every stage 9.54 / 9.54)
ACT_LINK_MAX_NUMEL = [int(os.environ.get("LOCATE_ACT_LINK_MAX", str(1 << 22)))]


class ActLink:
    """Ties a conv whose launch also wrote a = RootTanh(y) (SNConvFn, act = {"link": ...}) to the ONE consumer of a, the next
    conv of the stage (libs/conv.py:19-20, libs/attention.py:44-46): that conv's input-gradient launch multiplies by RootTanh'(y)
    in its epilogue and says so here, and the producer's backward then takes the arriving gradient as the gradient of y -
    the activation has no launch of its own in either direction."""
    __slots__ = ("pre", "premultiplied")

    def __init__(self):
        self.pre = None
        self.premultiplied = False


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, what):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError("%s must be a float32 CUDA/HIP tensor (the kernels run on MI355X only), got %s on %s"
                        % (what, t.dtype, t.device))
    return t


def _c(t, what="tensor"):
    _chk(t, what)
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _p(t):
    return t.data_ptr() if t is not None else None


# ------------------------------------------------------------------------------------------------
# largest-magnitude words for the contractions' fp16-piece form (csrc/conv.hip, precision 2)
# ------------------------------------------------------------------------------------------------
class _AmaxArena:
    """The fp32-faithful contractions run fastest with both operands split into TWO fp16 pieces (three MFMAs per slice instead
    of the six of the three-piece bf16 split) - which needs every operand tensor scaled by a power of two into fp16's range,
    i.e. its largest magnitude known when the consumer starts.  The kernels that PRODUCE a contraction's operand (norm + RootTanh
    in front of a conv stage, RootTanh's backward in front of its input / weight gradients) fold that maximum into one device
    word - an atomic max on the bit pattern: order-independent, so deterministic - and the word rides along on the tensor
    (`_locate_amax`).  A tensor without one (any other producer, a gradient autograd had to sum, a non-contiguous view that
    was copied) simply takes the six-product path: same fp32-level results, nothing is ever guessed.

    Words come from zeroed blocks, each word set handed out once (never reused: no aliasing between passes or networks); a
    new block costs one fill launch.  The step driver starts one block per training iteration (new_step: ~90 tensors);
    stand-alone use simply takes a new block whenever the current one is used up."""
    SLOTS = 32            # per block when nobody announced a step
    STEP_SLOTS = 160      # per training iteration (TrainStep): two generator and two discriminator forwards, three backward passes
                          # - a first guess, raised to what an iteration actually took (end_step) before the next one starts

    def __init__(self):
        self.block, self.used, self.cap = None, 0, 0
        self.enabled = True
        self.words = None            # words per tensor (locate_absmax_words: producers spread their atomics over them)
        self._step = False
        self._step_used = 0          # word sets the running iteration has taken in all (over every block)

    def new_pass(self):
        """A forward / backward pass begins: outside a training iteration it gets a block of its own."""
        if not self._step:
            self.block = None

    def new_step(self, device=None):
        """A training iteration begins (TrainStep.begin_iteration / d_generate): ONE block serves all its passes.  Idempotent
        until end_step().  With a device the zeroed block is created NOW - before passes of the iteration that may run
        concurrently (two generator passes on two streams) publish into it."""
        if not self._step:
            self.block = None
            self._step = True
            self._step_used = 0
        if device is not None and self.enabled and (self.block is None or self.block.device != device):
            if self.words is None:
                self.words = lib().locate_absmax_words()
            self.cap = self.STEP_SLOTS
            self.block = torch.zeros(self.cap * self.words, dtype=torch.int32, device=device)
            self.used = 0

    def end_step(self):
        """The iteration is over: the next one's block holds what this one took, with room to spare - a replayed (captured)
        iteration can never need more than the eager iterations before it did."""
        if self._step and self._step_used + 16 > self.STEP_SLOTS:
            self.STEP_SLOTS = self._step_used + self._step_used // 4 + 16
        self._step = False

    def slot(self, device):
        if not self.enabled:
            return None
        if self.words is None:
            self.words = lib().locate_absmax_words()
        if self.block is None or self.used >= self.cap or self.block.device != device:
            overflow = self._step and self.block is not None and self.block.device == device
            if overflow and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("hipGraph capture needs more largest-magnitude word sets (%d) than the iteration's block holds: "
                                   "run one eager iteration of this configuration first (the block is sized from it)" % (self._step_used + 1))
            self.cap = self.STEP_SLOTS if self._step else self.SLOTS
            self.block = torch.zeros(self.cap * self.words, dtype=torch.int32, device=device)
            self.used = 0
            if overflow:
                # a block created in the MIDDLE of an iteration: its zero-fill runs on this stream, while the word sets it hands
                # out later may go to producers on other streams (the generator pass beside the discriminator's) - the fill must
                # be complete before any of them can publish.  Eager iterations only, and only until end_step has sized the block.
                torch.cuda.synchronize()
        s = self.block[self.used * self.words:(self.used + 1) * self.words]
        self.used += 1
        if self._step:
            self._step_used += 1
        return s


AMAX = _AmaxArena()
# contractions below this much work per launch are latency-bound - nothing to gain from fewer matrix instructions (measured on the
# whole step, one box: threshold 0.5 / 1 / 2.45 / 3 / 12 GFLOP -> 10.45 / 10.50 / 10.55 / 10.59 / 10.61 ms, form off 11.06)
F16_MIN_FLOPS = 0.5e9


F16_CALLS = {"fwd": 0, "dgrad": 0, "wgrad": 0}      # launches that took the fp16-piece form (tests check the path is live)
# producers leave their output's largest magnitude from this many elements up: below it no contraction of F16_MIN_FLOPS follows.
# The fp8 form needs EVERY operand's maximum (Model.set_precision("fp8") lowers this to 1: a producer's few extra words cost less than
# the separate pass _ensure_amax would launch)
AMAX_MIN_NUMEL = [1 << 16]


def _tag(t, words):
    """Attach a tensor's largest-magnitude words, stamped with the version of its data: an in-place change afterwards - autograd
    summing a second consumer's gradient INTO this tensor - bumps the version counter and the stale words are ignored (the
    contraction then takes the six-product form; a wrong maximum would mis-scale the fp16 pieces)."""
    t._locate_amax = (words, t._version)
    return t


def _amax_of(t):
    """The largest-magnitude words of t, or None.  A full view of a tagged tensor (autograd's own reshape of a gradient on its
    way through a view node) finds them on its base (a view shares its base's version counter)."""
    if not AMAX.enabled:
        return None
    a = getattr(t, "_locate_amax", None)
    if a is None:
        base = t._base
        if base is not None and base.numel() == t.numel() and base.data_ptr() == t.data_ptr():
            a = getattr(base, "_locate_amax", None)
    if a is None or a[1] != t._version:
        return None
    return a[0]


def carry_amax(src, view):
    """`view` shows the same elements as `src` (reshape / view of a contiguous tensor): it inherits src's largest-magnitude words,
    which Python attributes do not do by themselves."""
    if view is not src and view.numel() == src.numel() and view.data_ptr() == src.data_ptr():
        a = getattr(src, "_locate_amax", None)
        if a is not None and a[1] == src._version:
            view._locate_amax = (a[0], view._version)
    return view


def _ensure_amax(t):
    """t's largest-magnitude words: the producer's, or - a tensor that came without (small maps, foreign producers) - from a pass
    of its own (the fp8 form cannot fall back to an unscaled arithmetic the way the fp16-piece form falls back to six products)."""
    a = _amax_of(t)
    if a is None:
        a = AMAX.slot(t.device)
        if a is None:
            raise RuntimeError("fp8 contractions need the largest-magnitude arena (ops.AMAX.enabled)")
        tc = t if _dense_planes(t) and t.is_contiguous() else t.contiguous()
        check(lib().locate_absmax(_p(tc), tc.numel(), _p(a), _stream()), "locate_absmax")
    return a


def tag_amax(t):
    """Computes t's largest-magnitude word with a pass of its own and attaches it (tests, tools; the hot path gets the word from
    the kernel that produces the tensor)."""
    slot = AMAX.slot(t.device)
    if slot is not None:
        check(lib().locate_absmax(_p(_c(t)), t.numel(), _p(slot), _stream()), "locate_absmax")
        _tag(t, slot)
    return t


# ------------------------------------------------------------------------------------------------
# RootTanh / tanh
# ------------------------------------------------------------------------------------------------
class RootTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot=None, amax=None):
        ctx.slot = slot
        x = _c(x, "root_tanh input")
        y = torch.empty_like(x)
        check(lib().locate_roottanh_fwd(_p(x), _p(y), x.numel(), _p(amax), _stream()), "locate_roottanh_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        g = _c(g)
        gx, acc = ctx.slot.claim(x) if ctx.slot is not None else (torch.empty_like(x), 0)
        # a fresh buffer of its own: its largest magnitude rides along for the contraction that consumes it (a shared fan-out
        # buffer does not get one - the other branch still adds into it)
        # (only where a contraction of F16_MIN_FLOPS could consume it: 4-d maps of some size, like the forward's rule)
        amax = AMAX.slot(x.device) if (ctx.slot is None and x.dim() >= 3 and x.numel() >= AMAX_MIN_NUMEL[0]) else None
        check(lib().locate_roottanh_bwd(_p(x), _p(g), _p(gx), x.numel(), acc, _p(amax), _stream()), "locate_roottanh_bwd")
        if amax is not None:
            _tag(gx, amax)
        return gx, None, None


class TanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x, "tanh input")
        y = torch.empty_like(x)
        check(lib().locate_tanh_fwd(_p(x), _p(y), x.numel(), _stream()), "locate_tanh_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        g = _c(g)
        gx = torch.empty_like(y)
        check(lib().locate_tanh_bwd(_p(y), _p(g), _p(gx), y.numel(), _stream()), "locate_tanh_bwd")
        return gx


class ActCatFn(torch.autograd.Function):
    """cat([latent, RootTanh(pre)], dim=1) for [B, z] / [B, w] matrices in ONE launch - a style-chain link
    (libs/block.py:119-125: the next style linear sees the latent next to the previous link's activated output).
    Second output: pre itself as the [B, w, 1, 1] norm scale it also is (libs/block.py:124) - so that pre has ONE consumer and
    its two gradients (through the activation, and as a scale) meet inside one backward kernel instead of an autograd add."""

    @staticmethod
    def forward(ctx, latent, pre):
        latent, pre = _c(latent, "style latent"), _c(pre, "style pre-activation")
        rows, z = latent.shape
        w = pre.shape[1]
        out = torch.empty(rows, z + w, dtype=torch.float32, device=pre.device)
        check(lib().locate_act_cat_rows_fwd(_p(latent), _p(pre), _p(out), rows, z, w, _stream()), "locate_act_cat_rows_fwd")
        ctx.save_for_backward(pre)
        ctx.z = z
        return out, pre.view(rows, w, 1, 1)

    @staticmethod
    def backward(ctx, g, g_scale):
        pre, = ctx.saved_tensors
        z = ctx.z
        if g is None:
            return None, (None if g_scale is None else g_scale.reshape(pre.shape))
        g = _chk(g, "style chain gradient")
        if g.stride(1) != 1:
            g = g.contiguous()
        add = None if g_scale is None else _c(g_scale, "style scale gradient")
        gpre = torch.empty_like(pre)
        check(lib().locate_act_rows_bwd(_p(pre), g.data_ptr() + 4 * z, g.stride(0), _p(add), _p(gpre), pre.shape[0], pre.shape[1],
                                        _stream()), "locate_act_rows_bwd")
        return (g[:, :z] if ctx.needs_input_grad[0] else None), gpre


def root_tanh(x):
    # the activation in front of a conv: its output's largest magnitude rides along for the contraction's fp16-piece form
    # (only worth a word where a contraction of F16_MIN_FLOPS could follow: 4-d maps, not the style chain's rows)
    amax = AMAX.slot(x.device) if x.dim() >= 3 and x.numel() >= AMAX_MIN_NUMEL[0] else None
    y = RootTanhFn.apply(x, _slot_of(x) if x.is_contiguous() else None, amax)
    if amax is not None:
        _tag(y, amax)
    return y
tanh = TanhFn.apply


def act_cat(latent, pre):
    """cat([latent, RootTanh(pre)], 1) as a launch of its own: the fall-back of the style chain for a link whose linear does not
    run through the 1x1-map kernel (which writes this concatenation itself, nn.LinearModule.pre_and_next_input)."""
    return ActCatFn.apply(latent, pre)[0]


# ------------------------------------------------------------------------------------------------
# InPlaceNorm
# ------------------------------------------------------------------------------------------------
class Runtime:
    """Host-side state of ONE network's forward / backward passes: how many independent calls the current forward stacks
    along the batch, the side stream its weight gradients go to, and what its running backward pass has deferred to its
    end (spectral-norm dv, side-stream results).  Every network owns one (`Generator.runtime`, `Discriminator.runtime`;
    its InPlaceNorm / SpectralNorm layers point at it), so two networks - or two models of the same class - in one process
    never see each other's state; layers built on their own share `DEFAULT_RUNTIME`.  The C ABI below this layer is
    stateless; this object is the Python layer's only mutable state besides the tensors themselves.

    Not supported (nor by the reference): two backward passes of the SAME network at the same time - they would race on
    its `.grad` buffers anyway."""

    def __init__(self):
        self.precision = 0               # contractions: 0 = fp32-faithful (the reference's arithmetic), 1 = bf16 operands
        self.stacked = 1                 # forward: the batch holds this many independent calls (main.py:149-152 as one pass)
        self.weight_grad_stream = None   # backward: weight gradients are launched on this stream (None: in line)
        self.weight_grad_side_min_flops = float(os.environ.get("LOCATE_WGRAD_SIDE_MIN_GFLOP", "0")) * 1e9
        self._dv_layers = {}             # id(v) -> (v, u, w, h, wd, state): layers whose dv is still to be finalised
        self._dv_tables = {}             # tuple of addresses -> (device table, max_h, max_wd)
        self._end_scheduled = False
        self._side = None                # side stream with work of the running backward pass on it
        self._side_results = []          # (parameter, gradient assigned at the end of the pass instead of through autograd)
        self._keep = []                  # operands of side-stream / deferred kernels: alive until the end of the pass
        # defer_finalisers: the small per-layer launches that only feed PARAMETER gradients - the spectral-norm rank-1 term
        # with du / dsigma, the stacked calls' activation-side dots, the gates' d(gamma) sums - are queued and run once per
        # backward pass, for all layers in one launch each (csrc/finalise.hip), from the end-of-pass callback; those
        # gradients are then assigned to .grad there instead of travelling through autograd.  Off: every layer finishes its
        # own gradients inside its backward node (what a data-parallel reducer's post-accumulate hooks need).
        self.defer_finalisers = True
        self._fin_dots = []              # packed FinRec records (csrc/finalise.hip)
        self._fin_rank1 = []
        self._fin_sums = []
        self._fin_chan = []
        self._fin_norm = []
        self._fin_wgrad = []             # packed SkwRec records (conv.hip): the pass's small-map weight gradients
        self._fin_slab = []              # packed SlabRec records (conv.hip): the split reductions of its other weight gradients
        self._streams = {}               # streams on which this pass queued deferred work / produced late gradients
        self._homes_taken = set()        # parameters whose bucket home (see _grad_home) a gradient of this pass already occupies

    # the copy of a network (copy.deepcopy in tests, DP replicas) gets a fresh runtime state, never the streams / tables
    def __deepcopy__(self, memo):
        rt = Runtime()
        rt.precision = self.precision
        rt.defer_finalisers = self.defer_finalisers
        return rt

    def stacked_calls(self, n):
        return _Stacked(self, n)

    def weight_grads_on(self, stream):
        return _OnStream(self, stream)

    def finalisers_deferred(self, on):
        return _Deferral(self, on)

    def reset(self):
        """Forget whatever an aborted backward pass left behind (an exception inside the autograd engine never runs its
        callbacks); the step driver calls this before every backward()."""
        self._end_scheduled = False
        self._side = None
        self._side_results = []
        self._keep = []
        self._streams = {}
        self._homes_taken = set()
        AMAX.new_pass()
        self._fin_dots, self._fin_rank1, self._fin_sums, self._fin_chan, self._fin_norm = [], [], [], [], []
        self._fin_wgrad, self._fin_slab = [], []
        for entry in self._dv_layers.values():
            entry[5]["k"] = 0
        self._dv_layers = {}

    # ---- end-of-backward work ----------------------------------------------------------------------------------------
    def _schedule_end(self):
        # whatever is finished at the end of the pass must first be complete on the stream it was produced on: the backward
        # nodes of a multi-stream forward (three-stream D-step) run on their forward's streams, and a layer whose gradients are
        # all deferred leaves no AccumulateGrad on its stream for the engine's own end-of-pass join to find
        st = torch.cuda.current_stream()
        self._streams[st.cuda_stream] = st
        if not self._end_scheduled:
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            self._end_scheduled = True

    def _end_of_backward(self):
        self._end_scheduled = False
        side, self._side = self._side, None
        cur = torch.cuda.current_stream()
        if side is not None:
            cur.wait_stream(side)
        used, self._streams = self._streams, {}
        for sid, st_ in used.items():
            if sid != cur.cuda_stream:
                cur.wait_stream(st_)
        # the queued finalisers, one launch per kind for the whole pass; the rank-1 launch also fills the dsigma slots the
        # batched dv below reads
        L = lib()
        st = _stream()
        if self._fin_wgrad:              # first: the rank-1 launch below reads their <G, W_bar> partials and corrects their gw
            cap = L.locate_wgrad_batch_max()
            for i in range(0, len(self._fin_wgrad), cap):
                chunk = self._fin_wgrad[i:i + cap]
                check(L.locate_wgrad_batch(b"".join(chunk), len(chunk), st), "locate_wgrad_batch")
            self._fin_wgrad = []
        if self._fin_slab:
            cap = L.locate_slab_reduce_max()
            for i in range(0, len(self._fin_slab), cap):
                chunk = self._fin_slab[i:i + cap]
                check(L.locate_slab_reduce_batch(b"".join(chunk), len(chunk), st), "locate_slab_reduce_batch")
            self._fin_slab = []
        for queue, fn, name in ((self._fin_dots, L.locate_fin_sn_dots, "locate_fin_sn_dots"),
                                (self._fin_rank1, L.locate_fin_sn_rank1, "locate_fin_sn_rank1"),
                                (self._fin_sums, L.locate_fin_sums, "locate_fin_sums"),
                                (self._fin_chan, L.locate_fin_channel_sums, "locate_fin_channel_sums"),
                                (self._fin_norm, L.locate_fin_norm_channels, "locate_fin_norm_channels")):
            if queue:
                blob = b"".join(queue)
                check(fn(blob, len(queue), st), name)
        self._fin_dots, self._fin_rank1, self._fin_sums, self._fin_chan, self._fin_norm = [], [], [], [], []
        results, self._side_results = self._side_results, []
        self._keep = []
        for param, grad in results:
            if param.grad is None or param.grad.data_ptr() == grad.data_ptr():
                param.grad = grad                 # (a gradient written to the parameter's bucket home is the .grad tensor itself)
            else:
                param.grad.add_(grad)
        self._homes_taken = set()
        self._finalize_dv()

    _FIN = __import__("struct").Struct("<8Q2q8i")

    @classmethod
    def _rec(cls, ptrs, longs=(), ints=()):
        ptrs = [0 if t is None else (t if isinstance(t, int) else t.data_ptr()) for t in ptrs]
        return cls._FIN.pack(*(ptrs + [0] * (8 - len(ptrs))), *(list(longs) + [0] * (2 - len(longs))), *(list(ints) + [0] * (8 - len(ints))))

    def queue_sn_dots(self, gy, y, bias, groups, Bg, M, plane, partial):
        self._fin_dots.append(self._rec([gy, y, bias, partial], [gy.stride(0), y.stride(0)], [groups, Bg, M, plane]))
        self._keep.append((gy, y, bias, partial))
        self._schedule_end()

    def queue_sn_rank1(self, partial, npartial, groups, sigma, sigma_stride, u, v, wv, wv_stride, gw, gu, dsig, h, wd):
        self._fin_rank1.append(self._rec([partial, sigma, u, v, wv, gw, gu, dsig], [wv_stride], [npartial, groups, sigma_stride, h, wd]))
        self._keep.append((partial, sigma, u, v, wv, gw, gu, dsig))
        self._schedule_end()

    def queue_sum(self, partial, count, out):
        self._fin_sums.append(self._rec([partial, out], [], [count]))
        self._keep.append((partial, out))
        self._schedule_end()

    def queue_channel_sum(self, g, out):
        """out[c] = sum over batch and space of g[:, c] (a bias gradient), with the pass's other channel sums."""
        Bn, Cn = g.shape[0], g.shape[1]
        hw = g.numel() // (Bn * Cn)
        slices = lib().locate_fin_channel_slices(Bn, Cn, hw)
        part = torch.empty(slices * Cn, dtype=torch.float32, device=g.device) if slices > 1 else None
        self._fin_chan.append(self._rec([g, out, part], [g.stride(0)], [Bn, Cn, hw]))
        self._keep.append((g, out, part))
        self._schedule_end()

    def queue_norm_channels(self, ws, plane_offset, planes, stats, dscale, dbias, B, C, groups):
        """dscale[c] / dbias[c] of one InPlaceNorm out of the plane sums its two-launch backward left in `ws`."""
        s1 = ws.data_ptr() + plane_offset
        self._fin_norm.append(self._rec([s1, s1 + 4 * planes, stats, dscale, dbias], [], [B, C, groups]))
        self._keep.append((ws, stats, dscale, dbias))
        self._schedule_end()

    def queue_small_wgrad(self, record, operands):
        self._fin_wgrad.append(record)
        self._keep.append(operands)
        self._schedule_end()

    def queue_slab_reduce(self, record, operands):
        self._fin_slab.append(record)
        self._keep.append(operands)
        self._schedule_end()

    def late_grad(self, param, grad):
        """`grad` is complete only at the end of the pass: assigned to / accumulated into param.grad by the end-of-pass callback."""
        self._side_results.append((param, grad))
        self._schedule_end()

    def defer_dv(self, v_param, u_param, w, h, wd):
        """Registers a layer for the batched dv of this backward pass; returns the slot its dsigma goes to."""
        st = v_param.__dict__.get("_locate_dv")
        if st is None or st["dv"].device != v_param.device:
            nch = (h + 63) // 64
            st = {"dv": torch.empty_like(v_param.detach()), "dsig": torch.zeros(4, dtype=torch.float32, device=v_param.device),
                  "scratch": torch.empty(wd + h + nch * wd, dtype=torch.float32, device=v_param.device), "k": 0}
            v_param.__dict__["_locate_dv"] = st
        home = v_param.__dict__.get("_locate_grad_buf")
        if home is not None and home.device == v_param.device and st["dv"].data_ptr() != home.data_ptr():
            st["dv"] = home.view(st["dv"].shape)          # v's gradient lives in its data-parallel bucket (see _grad_home)
        k = st["k"]
        if k >= 4:
            raise RuntimeError("a spectral-norm layer was differentiated through more than 4 forwards in one backward pass")
        st["k"] = k + 1
        self._dv_layers[id(v_param)] = (v_param, u_param, w, h, wd, st)
        self._schedule_end()
        return st["dsig"][k:]

    def _finalize_dv(self):
        import struct
        layers = list(self._dv_layers.values())
        self._dv_layers = {}
        if not layers:
            return
        # every field a record holds: equal keys mean byte-identical tables, whatever objects the addresses belonged to before
        key = tuple((w.data_ptr(), u.data_ptr(), st["dv"].data_ptr(), st["dsig"].data_ptr(), st["scratch"].data_ptr(), h, wd)
                    for _, u, w, h, wd, st in layers)
        tab = self._dv_tables.get(key)
        if tab is None:
            rec = struct.Struct("<8Q4i")
            buf = bytearray()
            for v, u, w, h, wd, st in layers:
                base = st["scratch"].data_ptr()
                buf += rec.pack(w.data_ptr(), u.data_ptr(), st["dv"].data_ptr(), st["dsig"].data_ptr(), 0, base, base + 4 * wd,
                                base + 4 * (wd + h), h, wd, (h + 63) // 64, 0)
            host = torch.frombuffer(buf, dtype=torch.uint8).clone()
            # the table holds raw addresses only (no tensor references): the layers' own parameters keep the memory alive
            tab = (host.to(layers[0][0].device), max(l[3] for l in layers), max(l[4] for l in layers))
            if len(self._dv_tables) > 8:
                self._dv_tables.clear()
            self._dv_tables[key] = tab
        check(lib().locate_sn_dv_batched(tab[0].data_ptr(), len(layers), tab[1], tab[2], _stream()), "locate_sn_dv_batched")
        for v, u, w, h, wd, st in layers:
            st["k"] = 0
            if v.grad is None or v.grad is st["dv"]:
                v.grad = st["dv"]
            else:
                v.grad.add_(st["dv"])


class _Stacked:
    """with runtime.stacked_calls(n): the batch of every op inside stacks `n` independent forward calls of the network (the
    reference runs them one after the other, main.py:149-152).  Ops whose result depends on the whole batch - InPlaceNorm's
    global statistics - then treat each of the n equal slices on its own."""

    def __init__(self, rt, n):
        self.rt, self.n = rt, int(n)

    def __enter__(self):
        self.prev = self.rt.stacked
        self.rt.stacked = self.n

    def __exit__(self, *exc):
        self.rt.stacked = self.prev
        return False


class _OnStream:
    """with runtime.weight_grads_on(stream): during backward(), the weight gradients of the spectral-normalised contractions
    (and their spectral-norm backward) are launched on `stream` instead of the stream of the backward pass.  They sit
    beside the pass's critical path - the chain of input gradients - and most kernels of that chain are too small to fill
    the chip, so the two run concurrently.  The results are NOT returned through autograd (a consumer on the main stream
    could read them too early): they are assigned to / accumulated into `.grad` of the weight, u and v parameters by a
    callback at the end of the backward pass, after the main stream has joined the side stream.  Consequences: only for
    `.backward()` (torch.autograd.grad would see no weight gradients), and tensor hooks on those parameters do not fire."""

    def __init__(self, rt, stream):
        self.rt, self.stream = rt, stream

    def __enter__(self):
        self.prev = self.rt.weight_grad_stream
        self.rt.weight_grad_stream = self.stream

    def __exit__(self, *exc):
        self.rt.weight_grad_stream = self.prev
        return False


class _Deferral:
    """with runtime.finalisers_deferred(False): every layer finishes its parameter gradients inside its own backward node
    (see Runtime.defer_finalisers)."""

    def __init__(self, rt, on):
        self.rt, self.on = rt, bool(on)

    def __enter__(self):
        self.prev = self.rt.defer_finalisers
        self.rt.defer_finalisers = self.on

    def __exit__(self, *exc):
        self.rt.defer_finalisers = self.prev
        return False


def _grad_home(rt, param, shape=None):
    """The buffer a parameter's gradient of this backward pass is written to: the parameter's HOME inside its data-parallel
    bucket (parallel.GradAllReducer.make_homes: a view of the bucket's resident flat buffer - the bucket is then all-reduced
    where it lies, nothing is packed or scattered) if it has one and nobody has claimed it in this pass yet, else a fresh tensor."""
    shape = tuple(param.shape if shape is None else shape)
    home = param.__dict__.get("_locate_grad_buf") if param is not None and hasattr(param, "__dict__") else None
    if home is not None and home.device == param.device and home.numel() == param.numel() and id(param) not in rt._homes_taken:
        rt._homes_taken.add(id(param))
        return home.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=param.device)


DEFAULT_RUNTIME = Runtime()     # layers used on their own (not inside a Generator / Discriminator)
# stand-alone layers finish every parameter gradient inside their own backward node: torch.autograd.grad, tensor hooks and a
# GradAllReducer used without TrainStep then see them (the networks' own runtimes defer to the end of the pass, TrainStep knows)
DEFAULT_RUNTIME.defer_finalisers = False


def stacked_calls(n):
    """Stacked-call context of the default runtime (stand-alone layers)."""
    return DEFAULT_RUNTIME.stacked_calls(n)


def weight_grad_stream(stream):
    """Side-stream context of the default runtime (stand-alone layers)."""
    return DEFAULT_RUNTIME.weight_grads_on(stream)


def reset_backward_state(*runtimes):
    for rt in runtimes or (DEFAULT_RUNTIME,):
        rt.reset()


# ------------------------------------------------------------------------------------------------
# fan-out of one tensor into two consumers: the second backward kernel to arrive ADDS into the first one's buffer
# ------------------------------------------------------------------------------------------------
class _GradSlot:
    """Gradient buffer shared by the backward kernels of the two consumers of a forked tensor.  claim() hands out the buffer
    and says whether it already holds the other consumer's share (then the kernel accumulates)."""
    __slots__ = ("buf", "claims")

    def __init__(self):
        self.buf = None
        self.claims = 0

    def claim(self, like, shape=None):
        shape = tuple(like.shape if shape is None else shape)
        self.claims += 1
        if self.claims > 2:
            raise RuntimeError("fork: more than two backward kernels claimed one fan-out buffer - each alias of a forked tensor "
                               "must feed exactly one participating consumer")
        if self.buf is None:
            self.buf = like.new_empty(shape)
            return self.buf, 0
        return (self.buf if tuple(self.buf.shape) == shape else self.buf.view(shape)), 1


class ForkFn(torch.autograd.Function):
    """x -> (x, x) as two autograd branches.  Where one tensor feeds two of this package's kernels (a block's input: skip
    branch and conv branch; an attention gate's input: the gate itself and the branch's norm), autograd would add the two
    gradients with one more launch and one more pass over the tensor.  Consumers that know about the fork (inplace_norm,
    residual_gate, feature_pool) write into ONE shared buffer instead - whichever backward kernel runs second accumulates
    (`accumulate` flags of the C ABI) - and both return that buffer; then there is nothing left to add here.  A consumer that
    does not take part simply returns its own gradient and the sum is formed as usual.

    Invariant (checked): an alias has ONE direct consumer.  If a participating consumer's alias also fed a second op, autograd
    would sum the shared buffer into a fresh tensor before the other branch has accumulated into it, and the first share would
    be counted twice - so a branch that claimed the buffer must deliver exactly that buffer."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        slot, ctx.slot = ctx.slot, None
        buf, claims = slot.buf, slot.claims
        slot.buf, slot.claims = None, 0
        if ga is None or gb is None:
            return (gb if ga is None else ga), None
        shared = [g for g in (ga, gb) if buf is not None and g.data_ptr() == buf.data_ptr()]
        if len(shared) != claims:
            raise RuntimeError("fork: %d consumer(s) accumulated into the shared gradient buffer but %d branch(es) delivered it - an "
                               "alias of a forked tensor fed more than one op" % (claims, len(shared)))
        if ga.data_ptr() == gb.data_ptr():
            return ga, None
        return ga + gb, None


def fork(x):
    """Two aliases of x for its two consumers (see ForkFn); a no-op when no gradient will flow."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    slot = _GradSlot()
    a, b = ForkFn.apply(x, slot)
    a._locate_slot = b._locate_slot = slot
    stats = getattr(x, "_locate_stats", None)      # partial norm statistics left by x's producer (residual_gate)
    if stats is not None:
        a._locate_stats = b._locate_stats = stats
    return a, b


class Fork3Fn(torch.autograd.Function):
    """x -> three aliases for a tensor with THREE consumers - a discriminator block's input feeds the conv branch's norm, the
    identity half of the skip branch's concatenation and that branch's 1x1 conv (libs/block.py:38-52, libs/scale.py:28-34).
    Autograd would sum the three gradients with two launches (one of them strided: the concatenation's gradient arrives as a
    channel slice); here they are added in ONE pass over batch-strided operands, in a fixed order: (norm + slice) + conv."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_identity, g_conv, g_norm):
        gs = [g for g in (g_norm, g_identity, g_conv) if g is not None]
        if len(gs) == 3 and all(_batch_strided(g) for g in gs) and gs[0].shape == gs[1].shape == gs[2].shape:
            a, b, c = gs
            out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
            per = a[0].numel()
            check(lib().locate_add3(_p(a), a.stride(0), _p(b), b.stride(0), _p(c), c.stride(0), _p(out), a.shape[0], per, _stream()),
                  "locate_add3")
            return out
        total = None
        for g in gs:
            total = g if total is None else total + g
        return total


def _batch_strided(g):
    """contiguous inside a batch element, any batch stride: what a channel slice of a contiguous tensor is"""
    return (g.is_cuda and g.dtype == torch.float32 and g.dim() >= 2 and g.shape[0] > 0 and g[0].is_contiguous()
            and (g.shape[0] == 1 or g.stride(0) >= g[0].numel()))


def fork3(x):
    """Three aliases of x for its three consumers, in the order (identity copy, conv, norm) - see Fork3Fn; x itself three
    times when no gradient will flow."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x, x
    a, b, c = Fork3Fn.apply(x)
    stats = getattr(x, "_locate_stats", None)      # partial norm statistics left by x's producer (residual_gate): the norm behind
    if stats is not None:                          # alias c would otherwise take them again with a pass of its own
        a._locate_stats = b._locate_stats = c._locate_stats = stats
    return a, b, c


def _slot_of(x):
    return getattr(x, "_locate_slot", None)


class InPlaceNormFn(torch.autograd.Function):
    """out = (x - mean(x)) * scale / std(x) + bias with global scalar statistics.  With `with_act` the forward
    returns RootTanh(out) instead; the backward recomputes out on the fly, nothing but x is kept."""

    @staticmethod
    def forward(ctx, x, scale, bias, with_act, groups, pre_partial=None, slot=None, amax=None, rt=None):
        ctx.slot = slot
        ctx.rt = rt or DEFAULT_RUNTIME
        ctx.scale_in, ctx.bias_in = scale, bias            # the parameters themselves (late gradients are assigned to them)
        x = _c(x, "norm input")
        B, C = x.shape[0], x.shape[1]
        hw = x.numel() // (B * C)
        if B % groups:
            raise ValueError("norm: batch %d does not split into %d stacked calls" % (B, groups))
        per_sample = scale.numel() == B * C and (B > 1 and scale.shape[0] == B)
        if not per_sample and scale.numel() != C:
            raise ValueError("norm scale must have C or B*C elements, got %s for input %s" % (tuple(scale.shape), tuple(x.shape)))
        scale_c, bias_c = _c(scale, "norm scale"), _c(bias, "norm bias")
        L = lib()
        st = _stream()
        stats = torch.empty(2 * groups, dtype=torch.float32, device=x.device)
        ws = _ws(L.locate_norm_stats_workspace_bytes(), x.device) if pre_partial is None else None
        out = torch.empty_like(x)       # RootTanh(norm(x)) when with_act, else norm(x)
        check(L.locate_norm_fwd(_p(x), _p(scale_c), int(per_sample), _p(bias_c), _p(out), int(bool(with_act)), _p(stats), B, C, hw,
                                groups, _p(ws), _p(pre_partial), _p(amax), st), "locate_norm_fwd")
        ctx.per_sample, ctx.with_act, ctx.groups = per_sample, bool(with_act), groups
        ctx.scale_shape, ctx.bias_shape = scale.shape, bias.shape
        ctx.save_for_backward(x, scale_c, bias_c, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        L = lib()
        st = _stream()
        g = _c(g)
        x, scale, bias, stats = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        hw = x.numel() // (B * C)
        dx, acc = ctx.slot.claim(x) if ctx.slot is not None else (torch.empty_like(x), 0)
        rt = ctx.rt
        need_scale, need_bias = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        # Two-launch form: dx (and the per-sample scale gradient) now; the per-channel sums dscale[c] / dbias[c] - parameter
        # gradients - with all the other norms of the pass at its end (locate_fin_norm_channels).  Taken when those gradients
        # go to leaf parameters (or are not wanted at all: a frozen discriminator).
        per_channel_scale_ok = ctx.per_sample or not need_scale or ctx.scale_in.is_leaf
        if rt.defer_finalisers and per_channel_scale_ok and (not need_bias or ctx.bias_in.is_leaf):
            ws = _ws(L.locate_norm_bwd_fused_workspace_bytes(B, C), x.device)
            dscale = torch.empty(ctx.scale_shape, dtype=torch.float32, device=x.device) if ctx.per_sample else None
            check(L.locate_norm_bwd_fused(_p(x), _p(g), _p(stats), _p(scale), int(ctx.per_sample), _p(bias), int(ctx.with_act), _p(dx),
                                          _p(dscale), B, C, hw, ctx.groups, _p(ws), acc, st), "locate_norm_bwd_fused")
            late_scale = _grad_home(rt, ctx.scale_in, ctx.scale_shape) if (need_scale and not ctx.per_sample) else None
            late_bias = _grad_home(rt, ctx.bias_in, ctx.bias_shape) if need_bias else None
            if late_scale is not None or late_bias is not None:
                rt.queue_norm_channels(ws, L.locate_norm_bwd_fused_plane_offset(), B * C, stats, late_scale, late_bias, B, C, ctx.groups)
                if late_scale is not None:
                    rt.late_grad(ctx.scale_in, late_scale)
                if late_bias is not None:
                    rt.late_grad(ctx.bias_in, late_bias)
            return dx, dscale, None, None, None, None, None, None, None
        dscale = torch.empty(ctx.scale_shape, dtype=torch.float32, device=x.device)
        dbias = torch.empty(ctx.bias_shape, dtype=torch.float32, device=x.device)
        ws = _ws(L.locate_norm_bwd_workspace_bytes(B, C), x.device)
        check(L.locate_norm_bwd(_p(x), _p(g), _p(stats), _p(scale), int(ctx.per_sample), _p(bias), int(ctx.with_act), _p(dx),
                                _p(dscale), _p(dbias), B, C, hw, ctx.groups, _p(ws), acc, st), "locate_norm_bwd")
        return dx, dscale, dbias, None, None, None, None, None, None


def inplace_norm(x, scale, bias, with_act=False, runtime=None):
    groups = (runtime or DEFAULT_RUNTIME).stacked
    # statistics partials left by the gate kernel that produced x (residual_gate): usable if taken for the same grouping
    pre = getattr(x, "_locate_stats", None)
    pre_partial = pre[0] if (pre is not None and pre[1] == groups and x.is_contiguous()) else None
    # the output's largest magnitude for the fp16-piece form of the contraction that consumes it (a conv stage behind
    # norm + RootTanh, the position gate's first 1x1 conv behind a plain norm)
    amax = AMAX.slot(x.device) if x.numel() >= AMAX_MIN_NUMEL[0] else None
    out = InPlaceNormFn.apply(x, scale, bias, with_act, groups, pre_partial, _slot_of(x) if x.is_contiguous() else None, amax, runtime)
    if amax is not None:
        _tag(out, amax)
    return out


# ------------------------------------------------------------------------------------------------
# residual gate
# ------------------------------------------------------------------------------------------------
class GateFn(torch.autograd.Function):
    """out = (gamma * a + 1) * x;  `a` has x's shape or is [B, C, 1, 1] (one value per plane)."""

    @staticmethod
    def forward(ctx, x, a, gamma, stats_groups=0, holder=None, slot=None, rt=None):
        ctx.slot = slot
        ctx.rt = rt or DEFAULT_RUNTIME
        ctx.gamma_param = gamma
        x = _c(x, "gate input")
        a = _c(a, "gate attention")
        gamma = _c(gamma, "gate gamma")
        planes = x.shape[0] * x.shape[1]
        hw = x.numel() // planes
        per_plane = a.numel() == planes and x.numel() != planes
        if not per_plane and a.numel() != x.numel():
            raise ValueError("gate: attention shape %s does not match input %s" % (tuple(a.shape), tuple(x.shape)))
        out = torch.empty_like(x)
        L = lib()
        n_g = x.numel() // stats_groups if stats_groups else 0
        if stats_groups and x.shape[0] % stats_groups == 0 and n_g > 1 and (stats_groups == 1 or n_g % 4 == 0):
            partial = torch.empty(L.locate_norm_stats_workspace_bytes() // 8, dtype=torch.float64, device=x.device)
            check(L.locate_gate_fwd_stats(_p(x), _p(a), int(per_plane), _p(gamma), _p(out), planes, hw, stats_groups, _p(partial),
                                          _stream()), "locate_gate_fwd_stats")
            holder.append((partial, stats_groups))
        else:
            check(L.locate_gate_fwd(_p(x), _p(a), int(per_plane), _p(gamma), _p(out), planes, hw, _stream()), "locate_gate_fwd")
        ctx.save_for_backward(x, a, gamma)
        ctx.per_plane = per_plane
        return out

    @staticmethod
    def backward(ctx, g):
        x, a, gamma = ctx.saved_tensors
        g = _c(g)
        L = lib()
        rt = ctx.rt
        planes = x.shape[0] * x.shape[1]
        hw = x.numel() // planes
        dx, acc = ctx.slot.claim(x) if ctx.slot is not None else (torch.empty_like(x), 0)
        da = torch.empty_like(a)
        need_gamma = ctx.needs_input_grad[2]
        # d(gamma) only feeds the parameter's gradient: with deferral its final sum joins the pass's batched finalisers and
        # the result is assigned at the end of the pass; a frozen gamma (the G-step's discriminator pass) needs none at all
        deferred = need_gamma and rt.defer_finalisers and ctx.gamma_param.is_leaf
        dgamma = (_grad_home(rt, ctx.gamma_param, gamma.shape) if deferred else torch.empty_like(gamma)) if need_gamma else None
        ws = _ws(L.locate_gate_bwd_workspace_bytes(planes), x.device)
        # full-map form: da goes straight into the branch's last conv (its data and weight gradients) - with its largest magnitude
        amax = AMAX.slot(x.device) if (not ctx.per_plane and da.numel() >= AMAX_MIN_NUMEL[0]) else None
        check(L.locate_gate_bwd(_p(x), _p(a), int(ctx.per_plane), _p(gamma), _p(g), _p(dx), _p(da), None if deferred else _p(dgamma),
                                planes, hw, _p(ws), acc, _p(amax), _stream()), "locate_gate_bwd")
        if amax is not None:
            _tag(da, amax)
        if deferred:
            rt.queue_sum(ws, L.locate_gate_bwd_partials(planes, hw), dgamma)
            rt.late_grad(ctx.gamma_param, dgamma.view(ctx.gamma_param.shape))
            dgamma = None
        return dx, da, dgamma, None, None, None, None


def residual_gate(x, a, gamma, runtime=None, with_stats=True):
    """out = (gamma a + 1) x.  with_stats: the kernel also leaves the InPlaceNorm statistics partials of `out` (for the
    runtime's current stacked-call grouping) on the result, so that a norm consuming it skips its own statistics pass."""
    slot = _slot_of(x) if x.is_contiguous() else None
    if not with_stats:
        return GateFn.apply(x, a, gamma, 0, None, slot, runtime)
    holder = []
    out = GateFn.apply(x, a, gamma, (runtime or DEFAULT_RUNTIME).stacked, holder, slot, runtime)
    if holder:
        out._locate_stats = holder[0]
    return out




# ------------------------------------------------------------------------------------------------
# softmax over the last dimension
# ------------------------------------------------------------------------------------------------
class SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x, "softmax input")
        n = x.shape[-1]
        y = torch.empty_like(x)
        check(lib().locate_softmax_fwd(_p(x), _p(y), x.numel() // n, n, _stream()), "locate_softmax_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        g = _c(g)
        n = y.shape[-1]
        gx = torch.empty_like(y)
        check(lib().locate_softmax_bwd(_p(y), _p(g), _p(gx), y.numel() // n, n, _stream()), "locate_softmax_bwd")
        return gx


softmax_lastdim = SoftmaxFn.apply


# ------------------------------------------------------------------------------------------------
# resampling / indexing
# ------------------------------------------------------------------------------------------------
class Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x, "upsample input")
        B, C, H, W = x.shape
        y = torch.empty(B, C, 2 * H, 2 * W, dtype=x.dtype, device=x.device)
        check(lib().locate_upsample2x_fwd(_p(x), _p(y), B * C, H, W, _stream()), "locate_upsample2x_fwd")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        g = _c(g)
        gx = torch.empty(B, C, H, W, dtype=g.dtype, device=g.device)
        check(lib().locate_upsample2x_bwd(_p(g), _p(gx), B * C, H, W, _stream()), "locate_upsample2x_bwd")
        return gx


class PoolUpsampleFn(torch.autograd.Function):
    """FeaturePooling to half the channels followed by the bilinear x2 upsample (the skip branch of a generator block,
    libs/scale.py:7-16,37-38) in one launch each way; the pooled map is never stored.  Bit for bit the two separate ops."""

    @staticmethod
    def forward(ctx, x, slot=None):
        ctx.slot = slot
        x = _c(x, "feature pooling input")
        B, C, H, W = x.shape
        y = torch.empty(B, C // 2, 2 * H, 2 * W, dtype=x.dtype, device=x.device)
        check(lib().locate_pool2_upsample2x_fwd(_p(x), _p(y), B * (C // 2), H, W, _stream()), "locate_pool2_upsample2x_fwd")
        ctx.in_shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.in_shape
        g = _c(g)
        if ctx.slot is not None:
            gx, acc = ctx.slot.claim(g, ctx.in_shape)
        else:
            gx, acc = torch.empty(ctx.in_shape, dtype=g.dtype, device=g.device), 0
        check(lib().locate_pool2_upsample2x_bwd(_p(g), _p(gx), B * (C // 2), H, W, acc, _stream()), "locate_pool2_upsample2x_bwd")
        return gx, None


def pool_upsample_ok(x, out_features):
    return x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] == 2 * out_features and POOL_UPSAMPLE[0]


def pool_upsample(x, out_features):
    """feature_pool(x, out_features) then upsample2x, fused (C == 2 * out_features)."""
    return PoolUpsampleFn.apply(x, _slot_of(x) if x.is_contiguous() else None)


POOL_UPSAMPLE = [os.environ.get("LOCATE_POOL_UPSAMPLE", "1") != "0"]


class AvgPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot=None):
        ctx.slot = slot
        x = _c(x, "avgpool input")
        B, C, H, W = x.shape
        y = torch.empty(B, C, H // 2, W // 2, dtype=x.dtype, device=x.device)
        check(lib().locate_avgpool2_fwd(_p(x), _p(y), B * C, H, W, _stream()), "locate_avgpool2_fwd")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        g = _c(g)
        if ctx.slot is not None:
            gx, acc = ctx.slot.claim(g, (B, C, H, W))
        else:
            gx, acc = torch.empty(B, C, H, W, dtype=g.dtype, device=g.device), 0
        check(lib().locate_avgpool2_bwd(_p(g), _p(gx), B * C, H, W, acc, _stream()), "locate_avgpool2_bwd")
        return gx, None


class FeaturePoolFn(torch.autograd.Function):
    """FeaturePooling: the raw-view mean over r adjacent flat elements (libs/scale.py:12-16)."""

    @staticmethod
    def forward(ctx, x, out_features, slot=None):
        ctx.slot = slot
        x = _c(x, "feature pooling input")     # the reference's .view() requires contiguity too
        B, C = x.shape[0], x.shape[1]
        if C % out_features:
            raise ValueError("FeaturePooling: %d channels not divisible by %d" % (C, out_features))
        r = C // out_features
        y = torch.empty((B, out_features) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        check(lib().locate_feature_pool_fwd(_p(x), _p(y), y.numel(), r, _stream()), "locate_feature_pool_fwd")
        ctx.in_shape, ctx.r = x.shape, r
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        if ctx.slot is not None:
            gx, acc = ctx.slot.claim(g, ctx.in_shape)
        else:
            gx, acc = torch.empty(ctx.in_shape, dtype=g.dtype, device=g.device), 0
        check(lib().locate_feature_pool_bwd(_p(g), _p(gx), g.numel(), ctx.r, acc, _stream()), "locate_feature_pool_bwd")
        return gx, None, None


upsample2x = Upsample2xFn.apply


def avgpool2(x):
    return AvgPool2Fn.apply(x, _slot_of(x) if x.is_contiguous() else None)


def feature_pool(x, out_features):
    return FeaturePoolFn.apply(x, out_features, _slot_of(x) if x.is_contiguous() else None)


def _copy_channels(src, dst, accumulate=False):
    """dst[:, :C] (+)= src for NCHW tensors whose channel planes are dense (batch stride free)."""
    B, C = src.shape[0], src.shape[1]
    hw = src[0, 0].numel() if src.dim() > 2 else 1
    check(lib().locate_copy_channels(_p(src), _p(dst), B, C, hw, src.stride(0), dst.stride(0), int(accumulate), _stream()),
          "locate_copy_channels")


def _dense_planes(t):
    """True if t is NCHW with contiguous [C, H, W] blocks per batch element (a channel slice of a contiguous tensor)."""
    if t.dim() < 2:
        return False
    expect = 1
    for size, stride in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
        if size != 1 and stride != expect:
            return False
        expect *= size
    return True


class CatChannelsFn(torch.autograd.Function):
    """torch.cat([a, b], dim=1) (libs/merge.py:15) as two strided plane copies - or one: when b was WRITTEN by its producer
    straight into the channel slice of `buf` that the concatenation assigns to it (cat_buffer / sn_conv(out=...)), only a is
    copied."""

    @staticmethod
    def forward(ctx, a, b, holder=None):
        a = _c(a, "cat input")
        ca, cb = a.shape[1], b.shape[1]
        buf = holder[0] if holder else None
        if buf is not None and buf.shape[1] == ca + cb and b.data_ptr() == buf[:, ca:].data_ptr() and b.stride() == buf[:, ca:].stride():
            out = buf
        else:
            b = _c(b, "cat input")
            out = torch.empty((a.shape[0], ca + cb) + tuple(a.shape[2:]), dtype=a.dtype, device=a.device)
            _copy_channels(b, out[:, ca:])
        _copy_channels(a, out[:, :ca])
        ctx.split = (ca, cb)
        return out

    @staticmethod
    def backward(ctx, g):
        # channel slices of the incoming gradient, as views: the conv kernels consume batch-strided operands in place,
        # everything else copies only if it has to
        ca, cb = ctx.split
        return g[:, :ca], g[:, ca:], None


def cat_channels(a, b, holder=None):
    return CatChannelsFn.apply(a, b, holder)


# ------------------------------------------------------------------------------------------------
# spectral-normalised dense contractions
# ------------------------------------------------------------------------------------------------
class ConvSpec:
    """Geometry of one wrapped torch.nn layer expressed through the regular convolution R of conv.hip.

    kind 'conv'  : y = R(x);            weight [M=C_out, C=C_in, KH, KW]
    kind 'convT' : y = R^T(x) (adjoint); weight [M=C_in, C=C_out, KH, KW]  (ConvTranspose2d layout)

    mode 'dense' (groups = 1), 'depthwise' (groups = C_in, weight [C_in * mult, 1, KH, KW] resp. [C_in, mult, KH, KW] for
    the transposed layer; grouped.hip) or 'groupdot' (a grouped conv whose kernel covers the whole map: per-group dot
    products; weight [G, C_in / G, H, W])."""
    __slots__ = ("kind", "kh", "kw", "stride", "pad_h", "pad_w", "mode")

    def __init__(self, kind, kh, kw, stride, pad_h, pad_w, mode="dense"):
        self.kind, self.kh, self.kw, self.stride, self.pad_h, self.pad_w, self.mode = kind, kh, kw, stride, pad_h, pad_w, mode

    def geometry(self, x_shape, w_shape):
        """Returns (geom[12] of R, output shape)."""
        B, Cx, H, W = x_shape
        M, C = w_shape[0], w_shape[1]
        s = self.stride
        if self.mode == "groupdot":
            if Cx % M or Cx // M != C or (H, W) != tuple(w_shape[2:]):
                raise ValueError("grouped full-size conv: input %s does not fit weight %s" % (tuple(x_shape), tuple(w_shape)))
            return [B, M, C * H * W] + [0] * 9, (B, M, 1, 1)
        if self.mode == "depthwise":
            if self.kind == "conv":
                if C != 1 or M % Cx:
                    raise ValueError("depthwise conv: input has %d channels, weight is %s" % (Cx, tuple(w_shape)))
                C = Cx
            else:
                if Cx != M:
                    raise ValueError("depthwise convT: input has %d channels, weight expects %d" % (Cx, M))
                C = M * C              # R's input side = the transposed layer's output side
        if self.kind == "conv":
            if Cx != C:
                raise ValueError("conv: input has %d channels, weight expects %d" % (Cx, C))
            OH = (H + 2 * self.pad_h - self.kh) // s + 1
            OW = (W + 2 * self.pad_w - self.kw) // s + 1
            return [B, C, H, W, M, self.kh, self.kw, s, self.pad_h, self.pad_w, OH, OW], (B, M, OH, OW)
        if Cx != M:
            raise ValueError("convT: input has %d channels, weight expects %d" % (Cx, M))
        # the input of the transposed conv is R's OUTPUT; R's input is the transposed conv's output
        RH = (H - 1) * s - 2 * self.pad_h + self.kh
        RW = (W - 1) * s - 2 * self.pad_w + self.kw
        return [B, C, RH, RW, M, self.kh, self.kw, s, self.pad_h, self.pad_w, H, W], (B, C, RH, RW)


def _geom(arr):
    return _IntArr12(*arr)


def _bs(t):
    return t.stride(0)


# Packed weight panels live ON the weight Parameter object (a plain Python attribute), validated by its version
# counter: re-packed only after the weight changed (optimizer step, load_state_dict), not on every forward /
# backward, and freed together with the parameter.
def _panel_owner(w):
    base = w._base
    return w if base is None else base


def _panel(owner, w, geom, garr, adjoint):
    L = lib()
    cache = owner.__dict__.setdefault("_locate_panels", {})
    key = (adjoint, tuple(geom[1:10]), w.data_ptr())
    hit = cache.get(key)
    ver = owner._version
    if hit is not None and hit[0] == ver and hit[1].device == w.device:
        return hit[1]
    nbytes = L.locate_conv_panel_bytes(garr, adjoint)
    buf = hit[1] if (hit is not None and hit[1].numel() == max(nbytes, 16) and hit[1].device == w.device) else _ws(nbytes, w.device)
    check(L.locate_conv_pack_panel(garr, adjoint, _p(w), _p(buf), _stream()), "locate_conv_pack_panel")
    cache[key] = (ver, buf, list(geom))
    return buf


# ------------------------------------------------------------------------------------------------
# gradient of the spectral-norm v vectors, batched per backward pass (Runtime.defer_dv / _finalize_dv)
#   dv = (sum_k dsigma_k) * W^T u_latest  - the same W^T u for every graph that used the layer, so it is computed
#   once per layer and pass, for all layers in three launches, from a callback at the end of the backward pass
#   (instead of three launches per layer and graph).  The result is written to a persistent buffer that becomes
#   v.grad (accumulated into an existing foreign .grad).
# ------------------------------------------------------------------------------------------------
DIRECT_REPACK = os.environ.get("LOCATE_DIRECT_REPACK", "1") != "0"


class _PackPlans:
    cache = {}       # signature of the stale set -> (device job table, n_jobs, total_blocks)


def refresh_panels(params):
    """Re-pack, on the current stream and in ONE launch, every cached weight panel of `params` whose weight has
    changed since it was packed (called after an optimizer step; otherwise the first user of a layer re-packs its
    panel lazily, one small launch per panel)."""
    import ctypes
    L = lib()
    stale = []
    for w in params:
        cache = w.__dict__.get("_locate_panels")
        if not cache:
            continue
        ver = w._version
        for key, (pver, buf, geom) in cache.items():
            if pver != ver and buf.device == w.device and w.is_contiguous():
                stale.append((w, key, buf, geom))
    if not stale:
        return
    # the optimizer kernel leaves the updated weights' largest magnitude per tensor (optim.Nadam._absmax_words, stamped with the
    # parameter version it belongs to): fp16-piece panels are then re-packed in one pass (conv.hip, direct form).  Anything else -
    # weights changed by other means, bf16-piece panels - takes the two-pass form.
    def wmax_of(w):
        hold = w.__dict__.get("_locate_wmax") if DIRECT_REPACK else None
        return hold[0].data_ptr() if (hold is not None and hold[1] == w._version and hold[0].device == w.device) else None

    wm = [wmax_of(w) if key[0] & 10 else None for w, key, _, _ in stale]          # (fp16-piece and fp8 panels are scaled)
    direct = int(DIRECT_REPACK)          # bf16-piece panels need no scale: always one pass
    sig = tuple((key[2], key[0], buf.data_ptr(), m, direct) + tuple(geom) for (_, key, buf, geom), m in zip(stale, wm))
    plan = _PackPlans.cache.get(sig)
    if plan is None:
        rec = L.locate_conv_pack_job_bytes()
        host = ctypes.create_string_buffer(rec * len(stale))
        start = 0
        two_pass = two_pass_f16 = 0
        for i, (w, key, buf, geom) in enumerate(stale):
            nb = ctypes.c_int(0)
            check(L.locate_conv_pack_job(_geom(geom), key[0], key[2], _p(buf), start, ctypes.addressof(host) + i * rec,
                                         ctypes.byref(nb), direct, wm[i]), "locate_conv_pack_job")
            start += nb.value
            if not L.locate_conv_pack_job_is_direct(ctypes.addressof(host) + i * rec):
                # pass bits of locate_conv_pack_panels: 1 = split launch (gather-kernel panels), 2 = absmax pre-pass (window panels)
                two_pass |= 2 if L.locate_conv_pack_job_is_window(ctypes.addressof(host) + i * rec) else 1
                two_pass_f16 |= int(bool(key[0] & 10))          # fp16-piece and fp8 panels: their absmax headers are cleared first
        table = torch.frombuffer(host, dtype=torch.uint8).clone().to(stale[0][0].device)
        plan = (table, len(stale), start, two_pass_f16, two_pass)
        if len(_PackPlans.cache) > 16:
            _PackPlans.cache.clear()
        _PackPlans.cache[sig] = plan
    check(L.locate_conv_pack_panels(_p(plan[0]), plan[1], plan[2], plan[3], plan[4], _stream()), "locate_conv_pack_panels")
    for w, key, buf, geom in stale:
        w.__dict__["_locate_panels"][key] = (w._version, buf, geom)


def sn_power_iteration(w_bar, u, v):
    """One power iteration on (u, v) IN PLACE (untracked, like the reference's `.data` writes,
    libs/spectral_norm.py:26-29).  Returns (sigma[2] = {sigma, 1/sigma}, wv[h] = W v)."""
    L = lib()
    h = w_bar.shape[0]
    wd = w_bar.numel() // h
    w = _c(w_bar.detach(), "weight_bar")
    sigma = torch.empty(2, dtype=torch.float32, device=w.device)
    wv = torch.empty(h, dtype=torch.float32, device=w.device)
    ws = _ws(L.locate_sn_workspace_bytes(h, wd), w.device)
    check(L.locate_sn_power_iter(_p(w), _p(_chk(u.detach(), "weight_u")), _p(_chk(v.detach(), "weight_v")), _p(sigma), _p(wv),
                                 h, wd, _p(ws), _stream()), "locate_sn_power_iter")
    return sigma, wv


# ---- shared pieces of the spectral-normalised contractions ------------------------------------------------------
def _sigma_args(sigma, batch):
    """sigma: {sigma, 1/sigma}, or [n, 2] for n stacked calls (one power iteration each, in call order).
    Returns (groups, scale_group_batch, scale_stride, pointer tensor to the first 1/sigma)."""
    groups = sigma.shape[0] if sigma.dim() == 2 else 1
    if groups == 1:
        return 1, 0, 0, sigma.reshape(-1)[1:]
    if batch % groups or groups > 4 or sigma.stride(1) != 1:
        raise ValueError("conv: batch %d cannot stack %d calls" % (batch, groups))
    return groups, batch // groups, sigma.stride(0), sigma[0, 1:]


_COUNTER_LANE = [0]


class counter_lane:
    """with counter_lane(1): the contractions launched (or captured) inside take a second set of split-K arrival counters.
    The same LAYER may then run twice at once - the two generator passes of one iteration on two streams (graph.py) - without
    the two launches sharing a counter block.  The lane's blocks are created by the eager iterations like all others."""

    def __init__(self, lane):
        self.lane = int(lane)

    def __enter__(self):
        self.prev = _COUNTER_LANE[0]
        _COUNTER_LANE[0] = self.lane

    def __exit__(self, *exc):
        _COUNTER_LANE[0] = self.prev


def _counters(owner, adjoint):
    """Arrival counters of one layer and direction for the in-launch split-K combine (locate_conv_counter_bytes: zero at
    creation, left zero by every launch).  Per layer, direction AND stream: the same layer may run on several streams at
    once (three-stream D-step), and launches that can overlap must not share a block.

    Under hipGraph capture nothing may be allocated or zero-filled here, so a capturing stream uses the block the eager
    warm-up iterations created FOR THAT STREAM (TrainStep's three D-step streams are the same objects eagerly and while
    capturing: their branches of one graph keep separate blocks); a capturing stream that never ran eagerly - the capture's
    origin stream - takes the spare block (key 0) created together with the first eager one.  One origin stream per capture,
    so the spare block is never shared by concurrent branches."""
    cache = owner.__dict__.setdefault("_locate_counters", {})
    sid = torch.cuda.current_stream().cuda_stream
    lane = _COUNTER_LANE[0]
    key = (adjoint, sid) if lane == 0 else (adjoint, "lane%d" % lane)
    buf = cache.get(key)
    if buf is not None and buf.device == owner.device:
        return buf
    if torch.cuda.is_current_stream_capturing():
        buf = cache.get((adjoint, 0)) if lane == 0 else None
        if buf is None or buf.device != owner.device:
            raise RuntimeError("hipGraph capture of a contraction that never ran eagerly: run one eager iteration first (its "
                               "arrival counters must exist before the capture)")
        return buf
    for k in {key, (adjoint, 0), (adjoint, "lane1")}:
        if k not in cache or cache[k].device != owner.device:
            cache[k] = torch.zeros(lib().locate_conv_counter_bytes(), dtype=torch.uint8, device=owner.device)
    return cache[key]


def _flops(geom):
    B, C, H, W, M, KH, KW, s_, ph, pw, OH, OW = geom
    return 2.0 * B * OH * OW * M * C * KH * KW


def _f16_ok(spec, geom, precision, *amax):
    """The fp16-piece form of a fp32-faithful contraction: every gathered operand's largest magnitude known, and enough work
    for the matrix instructions to matter."""
    return precision == 0 and spec.mode == "dense" and all(a is not None for a in amax) and _flops(geom) >= F16_MIN_FLOPS


# LOCATE_WINDOW: "0" = gather kernels everywhere, "all" = the window kernels wherever the geometry has the form (tests, A/B runs),
# default = where locate_conv_win_ok says they are the measured choice
WIN_MODE = {"0": 0, "all": 2}.get(os.environ.get("LOCATE_WINDOW", "1"), 1)
WIN_CALLS = [0]                                                 # launches that took the window form (tests check the path is live)
FP8_CALLS = [0]                                                 # forward / input-gradient launches with fp8 operands
_WIN_CACHE = {}


def _win_ok(geom, garr, adjfmt, x):
    """Whether this contraction takes the window form (locate_conv_win_ok: 1 = recommended, 2 = available), remembered per
    geometry / direction / alignment."""
    if not WIN_MODE:
        return False
    key = (tuple(geom), adjfmt, _bs(x) & 1, x.data_ptr() & 7)
    hit = _WIN_CACHE.get(key)
    if hit is None:
        hit = lib().locate_conv_win_ok(garr, adjfmt, _bs(x), _p(x))
        _WIN_CACHE[key] = hit
    return hit == 1 or (hit == 2 and WIN_MODE == 2)


def _win_ws_bytes(geom, garr, adjfmt):
    key = (tuple(geom), adjfmt, "ws")
    hit = _WIN_CACHE.get(key)
    if hit is None:
        hit = lib().locate_conv_win_workspace_bytes(garr, adjfmt)
        _WIN_CACHE[key] = hit
    return hit


def _contract(forward_of_r, x, w, owner, spec, geom, garr, sigma, bias, y, precision=0, amax=None, epilogue=None):
    """y = R(x) (forward_of_r) or R^T(x), times 1/sigma, plus bias - dispatched on the layer's grouping mode.
    amax: x's largest-magnitude word, if its producer left one (fp16-piece form, panel format bit 2).
    epilogue: _ActEpilogue (regular direction on a 1x1 map only) - RootTanh(y) written as a second output."""
    L = lib()
    st = _stream()
    _, sbg, sst, inv_sigma = _sigma_args(sigma, x.shape[0])
    if spec.mode == "dense":
        f16 = _f16_ok(spec, geom, precision, amax)
        prec, fmt, am = (2, 2, amax) if f16 else (precision, 0, None)
        if f16:
            F16_CALLS["fwd" if forward_of_r == (spec.kind == "conv") else "dgrad"] += 1
        if precision == 3:               # fp8 operands (csrc/convfp8.hip): panel format bit 3, the gathered tensor's largest magnitude
            prec, fmt, am = 3, 8, _ensure_amax(x)
            FP8_CALLS[0] += 1
        adj = 0 if forward_of_r else 1
        # the window form (csrc/convwin.hip): the gathered operand staged in LDS once for all taps - where the geometry has it
        win = precision != 3 and _win_ok(geom, garr, adj | fmt, x)
        if win:
            fmt |= 4
            prec |= 16
            WIN_CALLS[0] += 1
            ws = _ws(_win_ws_bytes(geom, garr, adj | (fmt & 2)), x.device)
        if forward_of_r:
            if not win:
                ws = _ws(L.locate_conv_fwd_workspace_bytes(garr), x.device)
            check(L.locate_conv_fwd(garr, _p(x), _bs(x), _p(_panel(owner, w, geom, garr, 0 | fmt)), _p(inv_sigma), sbg, sst, _p(bias), _p(y),
                                    _bs(y), _p(ws), _p(_counters(owner, 0)), prec, _p(am),
                                    ctypes.addressof(epilogue) if epilogue is not None else None, st), "locate_conv_fwd")
        else:
            if not win:
                ws = _ws(L.locate_conv_dgrad_workspace_bytes(garr), x.device)
            check(L.locate_conv_dgrad(garr, _p(x), _bs(x), _p(_panel(owner, w, geom, garr, 1 | fmt)), _p(inv_sigma), sbg, sst, _p(bias), _p(y),
                                      _bs(y), _p(ws), _p(_counters(owner, 1)), prec, _p(am),
                                      ctypes.addressof(epilogue) if epilogue is not None else None, st), "locate_conv_dgrad")
        return y
    if bias is not None:
        raise NotImplementedError("grouped convolutions carry no bias in the reference (libs/conv.py:15, libs/attention.py:18)")
    if spec.mode == "depthwise":
        fn, name = (L.locate_dwconv_fwd, "locate_dwconv_fwd") if forward_of_r else (L.locate_dwconv_dgrad, "locate_dwconv_dgrad")
        check(fn(garr, _p(x), _bs(x), _p(w), _p(inv_sigma), sbg, sst, _p(y), _bs(y), st), name)
        return y
    B, G, Ln = geom[:3]
    if forward_of_r:
        check(L.locate_groupdot_fwd(_p(x), _bs(x), _p(w), _p(inv_sigma), sbg, sst, _p(y), B, G, Ln, st), "locate_groupdot_fwd")
    else:
        check(L.locate_groupdot_dgrad(_p(x), _p(w), _p(inv_sigma), sbg, sst, _p(y), _bs(y), B, G, Ln, st), "locate_groupdot_dgrad")
    return y


def act_epilogue_ok(spec, x_shape):
    """The activated second output exists where the regular direction runs on a 1x1 map (skinny_rows_kernel)."""
    return spec.kind == "conv" and spec.mode == "dense" and (spec.kh, spec.kw, spec.stride, spec.pad_h, spec.pad_w) == (1, 1, 1, 0, 0) \
        and tuple(x_shape[2:]) == (1, 1)


def _conv_apply(x, w, owner, spec, geom, garr, sigma, bias, out_shape, precision=0, amax=None, out=None, epilogue=None):
    """y = conv(x, W_bar) / sigma + bias (Conv2d or ConvTranspose2d semantics per `spec`).  out: a dense-plane view of that
    shape to write into (a channel slice of a concatenation buffer) instead of a fresh tensor."""
    if out is not None and spec.mode == "dense" and tuple(out.shape) == tuple(out_shape) and _dense_planes(out) and out.dtype == torch.float32:
        y = out
    else:
        y = torch.empty(out_shape, dtype=torch.float32, device=x.device)
    return _contract(spec.kind == "conv", x, w, owner, spec, geom, garr, sigma, bias, y, precision, amax, epilogue)


def _conv_input_grad(gy, x_like, w, owner, spec, geom, garr, sigma, precision=0, amax=None, link=None):
    """Gradient w.r.t. the layer input - with `link` (the input was RootTanh(link.pre), ActLink) w.r.t. link.pre: the launch's
    epilogue multiplies by RootTanh'(link.pre), bit for bit what locate_roottanh_bwd makes of the plain input gradient."""
    if spec.mode == "groupdot":
        gy = gy.contiguous()
    gx = torch.empty_like(x_like)
    epi = None
    if link is not None:
        pre = link.pre
        slot = AMAX.slot(gx.device) if (gx.dim() >= 3 and gx.numel() >= AMAX_MIN_NUMEL[0]) else None
        epi = _ActEpilogue(None, 0, None, 0, 0, 0, pre.data_ptr(), _bs(pre), _p(slot))
    _contract(spec.kind != "conv", gy, w, owner, spec, geom, garr, sigma, None, gx, precision, amax, epi)
    if link is not None:
        link.premultiplied = True
        if slot is not None:
            _tag(gx, slot)
    return gx


def _raw_weight_grad(spec, geom, garr, xin, gout, gw, w_ref, inv_sigma, sbg, sst, partial, precision=0, amax_in=None, amax_out=None,
                     rt=None, group_dots=0):
    """gw = (sum over the batch of R's input x R's output gradient) / sigma, plus the partial sums of <G, W_bar>.
    With a runtime that defers its finalisers, the layers on 1x1 maps / with one output pixel are only queued: one launch for
    all of them at the end of the pass (Runtime.queue_small_wgrad)."""
    L = lib()
    st = _stream()
    if spec.mode == "dense" and rt is not None and rt.defer_finalisers:
        rec = ctypes.create_string_buffer(L.locate_wgrad_batch_record_bytes())
        if L.locate_wgrad_batch_record(garr, _p(xin), _bs(xin), _p(gout), _bs(gout), _p(gw), _p(w_ref), _p(inv_sigma), sbg, sst,
                                       _p(partial), rec) > 0:
            rt.queue_small_wgrad(rec.raw, (xin, gout, gw, w_ref, inv_sigma, partial))
            return
    if spec.mode == "dense":
        # group_dots: stacked calls with the per-call <G_k, W_bar> partials out of the split reduction (its own slab layout)
        nbytes = L.locate_conv_wgrad_group_workspace_bytes(garr, group_dots) if group_dots else L.locate_conv_wgrad_workspace_bytes(garr)
        defer = rt is not None and rt.defer_finalisers and nbytes > 0
        # a deferred split reduction reads its slab at the end of the pass: the layer gets a slab of its own instead of the
        # shared scratch buffer
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xin.device) if defer else _ws(nbytes, xin.device)
        rec = ctypes.create_string_buffer(L.locate_slab_reduce_record_bytes()) if defer else None
        f16 = _f16_ok(spec, geom, precision, amax_in, amax_out)
        F16_CALLS["wgrad"] += int(f16)
        if precision == 3:               # fp8 operands: both activations' largest magnitudes
            amax_in, amax_out = _ensure_amax(xin), _ensure_amax(gout)
        scaled = f16 or precision == 3
        check(L.locate_conv_wgrad(garr, _p(xin), _bs(xin), _p(gout), _bs(gout), _p(gw), _p(w_ref), _p(inv_sigma), sbg, sst, _p(partial),
                                  _p(ws), 2 if f16 else precision, _p(amax_in) if scaled else None, _p(amax_out) if scaled else None, rec, st),
              "locate_conv_wgrad")
        if defer and L.locate_slab_reduce_record_blocks(rec) > 0:
            rt.queue_slab_reduce(rec.raw, (ws, gw, w_ref, inv_sigma, partial))
    elif spec.mode == "depthwise":
        ws = _ws(L.locate_dwconv_wgrad_workspace_bytes(garr), xin.device)
        check(L.locate_dwconv_wgrad(garr, _p(xin), _bs(xin), _p(gout), _bs(gout), _p(gw), _p(w_ref), _p(inv_sigma), sbg, sst,
                                    _p(partial), _p(ws), st), "locate_dwconv_wgrad")
    else:
        B, G, Ln = geom[:3]
        gout = gout.contiguous()
        check(L.locate_groupdot_wgrad(_p(xin), _bs(xin), _p(gout), _p(gw), _p(w_ref), _p(inv_sigma), sbg, sst, _p(partial), B, G, Ln,
                                      st), "locate_groupdot_wgrad")


def _weight_grad_partials(spec, geom, garr):
    L = lib()
    if spec.mode == "dense":
        return L.locate_conv_wgrad_partials(garr)
    if spec.mode == "depthwise":
        return L.locate_dwconv_wgrad_partials(garr)
    return L.locate_groupdot_wgrad_partials(geom[1], geom[2])


def _conv_weight_grad(rt, x, gy, y, bias, w, u_param, v_param, sigma, wv, spec, geom, garr, need_u, need_v, amax_x=None, amax_gy=None,
                      owner=None):
    """dW_bar (incl. the rank-1 spectral-norm term) and du; dv is batched over the whole backward pass
    (Runtime.defer_dv).  y / bias are only read for stacked calls (<G_k, W_bar> taken on the activation side).
    With rt.defer_finalisers the rank-1 term, du and dsigma (and the stacked calls' dots) are only QUEUED here: the returned
    buffers are complete after the pass's batched finalisers have run (Runtime._end_of_backward)."""
    L = lib()
    st = _stream()
    groups, sbg, sst, inv_sigma = _sigma_args(sigma, x.shape[0])
    gw = _grad_home(rt, owner, w.shape) if (owner is not None and owner.is_leaf) else torch.empty_like(w)
    xin, gout = (x, gy) if spec.kind == "conv" else (gy, x)    # transposed: R's input is gy, its output-gradient x
    am_in, am_out = (amax_x, amax_gy) if spec.kind == "conv" else (amax_gy, amax_x)
    h = w.shape[0]
    wd = w.numel() // h
    u, v = u_param.detach(), v_param.detach()
    gu = (_grad_home(rt, u_param) if u_param.is_leaf else torch.empty_like(u)) if need_u else None
    if groups > 1:
        # gw = sum_k G_k / sigma_k in one pass (gy weighted per call while it is loaded); dsigma_k from
        # <gy_k, y_k - bias>; rank-1 correction with the summed dsigma
        npg = L.locate_conv_wgrad_group_partials(garr, groups) if (rt.defer_finalisers and spec.mode == "dense") else 0
        if npg > 0:
            # d(sigma_k) from the WEIGHT side: the split reduction's slabs never cross a call boundary, so its pass over them also
            # yields <G_k / sigma_k, W_bar> per call - the value of <gy_k, y_k - bias> without reading gy and y again
            partial = torch.empty(groups * npg, dtype=torch.float64, device=x.device)
            _raw_weight_grad(spec, geom, garr, xin, gout, gw, w, inv_sigma, sbg, sst, partial, rt.precision, am_in, am_out, rt, groups)
            dsig = rt.defer_dv(v_param, u_param, w, h, wd) if need_v else None
            rt.queue_sn_rank1(partial, npg, groups, sigma, sigma.stride(0), u, v, wv, wv.stride(0), gw, gu, dsig, h, wd)
            return gw, gu
        _raw_weight_grad(spec, geom, garr, xin, gout, gw, None, inv_sigma, sbg, sst, None, rt.precision, am_in, am_out, rt)
        dsig = rt.defer_dv(v_param, u_param, w, h, wd) if need_v else None
        Bn, Mn = gy.shape[0], gy.shape[1]
        plane = gy.numel() // (Bn * Mn)
        if rt.defer_finalisers:
            npart = L.locate_fin_sn_dot_partials(Bn // groups, Mn, plane)
            partial = torch.empty(groups * npart, dtype=torch.float64, device=x.device)
            rt.queue_sn_dots(gy, y, bias, groups, Bn // groups, Mn, plane, partial)
            rt.queue_sn_rank1(partial, npart, groups, sigma, sigma.stride(0), u, v, wv, wv.stride(0), gw, gu, dsig, h, wd)
            return gw, gu
        gws = _ws(L.locate_sn_group_workspace_bytes(), x.device)
        check(L.locate_sn_weight_bwd_grouped(_p(gy), _bs(gy), _p(y), _bs(y), _p(bias), groups, Bn // groups, Mn, plane,
                                             _p(sigma), sigma.stride(0), _p(u), _p(v), _p(wv),
                                             wv.stride(0), _p(gw), _p(gu), _p(dsig), h, wd, _p(gws), st),
              "locate_sn_weight_bwd_grouped")
        return gw, gu
    # one pass: gw = G / sigma_k (G = gradient w.r.t. the normalised weight) plus the partial sums of <G, W_bar>;
    # then the rank-1 spectral-norm correction in place
    npart = _weight_grad_partials(spec, geom, garr)
    partial = torch.empty(npart, dtype=torch.float64, device=x.device)
    _raw_weight_grad(spec, geom, garr, xin, gout, gw, w, inv_sigma, 0, 0, partial, rt.precision, am_in, am_out, rt)
    dsig = rt.defer_dv(v_param, u_param, w, h, wd) if need_v else None
    if rt.defer_finalisers:
        rt.queue_sn_rank1(partial, npart, 0, sigma, 0, u, v, wv, 0, gw, gu, dsig, h, wd)
        return gw, gu
    check(L.locate_sn_weight_bwd(_p(partial), npart, _p(u), _p(v), _p(sigma), _p(wv), _p(gw), _p(gu), _p(dsig), h, wd, st),
          "locate_sn_weight_bwd")
    return gw, gu


def _bias_grad(gy):
    Bn, Cn = gy.shape[0], gy.shape[1]
    gb = torch.empty(Cn, dtype=torch.float32, device=gy.device)
    L = lib()
    hw = gy.numel() // (Bn * Cn)
    ws = _ws(L.locate_channel_sum_workspace_bytes(Bn, Cn, hw), gy.device)
    check(L.locate_channel_sum(_p(gy), _p(gb), Bn, Cn, hw, _bs(gy), _p(ws), _stream()), "locate_channel_sum")
    return gb


def _dense(t, what):
    _chk(t, what)
    return t if _dense_planes(t) else t.contiguous()


class SNConvFn(torch.autograd.Function):
    """y = conv(x, W_bar / sigma) + bias for Conv2d / ConvTranspose2d semantics (Conv1d(k=1), Linear and the
    (S x 1)/(1 x S) feature-attention convs are reshaped to 1x1 convs by the caller).  sigma / wv are the
    results of THIS forward's power iteration; u, v are inputs only so that their gradients can be returned
    (the reference's main.py:172 makes them trainable) - they are read at backward time, i.e. with the values
    left by the latest forward, exactly like the reference's autograd does."""

    @staticmethod
    def forward(ctx, x, w_bar, u, v, bias, sigma, wv, spec, rt=None, guard=None, out_holder=None, act=None, in_link=None):
        # out_holder: [view] - the output is written into this channel slice of a concatenation buffer (hidden in a list so
        # that autograd does not see an input being returned)
        ctx.guard = guard              # (SpectralNormBatch, its ring sets): sigma / wv are views of them, still intact at backward?
        x = _dense(x, "conv input")
        w = _c(w_bar, "weight_bar")
        owner = _panel_owner(w_bar)
        geom, out_shape = spec.geometry(tuple(x.shape), tuple(w.shape))
        garr = _geom(geom)
        b = _c(bias) if bias is not None else None
        rt = rt or DEFAULT_RUNTIME
        ctx.amax_x = _amax_of(x)
        # act = {"latent": [B, z] or None}: second output RootTanh(y) from the same launch (1x1 maps) - alone as [B, M, 1, 1], or
        # as the next style link's input [B, z + M] = cat([latent, RootTanh(y)]) (libs/block.py:119-125)
        # act = {"link": ActLink}: the same on ANY map - second output RootTanh(y) [B, M, OH, OW] for the stage's next conv, whose
        # input-gradient launch then multiplies by RootTanh'(y) itself (ActLink); its largest magnitude rides along
        # in_link: this conv IS such a consumer (x = RootTanh(in_link.pre))
        epi = second = second_amax = None
        ctx.act_z = None
        ctx.act_link = act.get("link") if act is not None else None
        ctx.in_link = in_link if (in_link is not None and spec.mode == "dense") else None
        if act is not None:
            Bn, Mn = out_shape[0], out_shape[1]
            lat = act.get("latent")
            z = 0 if lat is None else lat.shape[1]
            second = torch.empty((Bn, z + Mn) if lat is not None else out_shape, dtype=torch.float32, device=x.device)
            if ctx.act_link is not None:
                if lat is not None or out_holder:
                    raise ValueError("an activation link excludes a latent prefix and an output slice")
                plane_elems = second.numel() // Bn
                if second.dim() >= 3 and second.numel() >= AMAX_MIN_NUMEL[0]:
                    second_amax = AMAX.slot(x.device)
                epi = _ActEpilogue(second.data_ptr(), plane_elems, None, 0, 0, 0, None, 0, _p(second_amax))
            else:
                epi = _ActEpilogue(second.data_ptr() + 4 * z, z + Mn, _p(lat), 0 if lat is None else lat.stride(0), z, 0, None, 0, None)
            if lat is not None and (lat.stride(1) != 1 or lat.shape[0] != Bn):
                raise ValueError("style latent must be [B, z] with contiguous rows")
            ctx.act_z = z
        y = _conv_apply(x, w, owner, spec, geom, garr, sigma, b, out_shape, rt.precision, ctx.amax_x, out_holder[0] if out_holder else None, epi)
        if ctx.act_link is not None:
            ctx.act_link.pre = y
            if second_amax is not None:
                _tag(second, second_amax)
        groups = sigma.shape[0] if sigma.dim() == 2 else 1
        ctx.groups = groups
        if groups > 1:
            ctx.save_for_backward(x, w, sigma, wv, y, b)     # stacked: <G_k, W_bar> is taken on the activation side
        elif second is not None:
            ctx.save_for_backward(x, w, sigma, wv, y)        # RootTanh' of the second output needs the pre-activation
        else:
            ctx.save_for_backward(x, w, sigma, wv)
        if second is not None:
            ctx.set_materialize_grads(False)                 # an unused output's gradient arrives as None, not as a zero tensor
        ctx.u, ctx.v = u, v            # live state, read at backward time
        ctx.bias_param = bias
        ctx.owner = owner
        ctx.rt = rt
        ctx.geom, ctx.spec, ctx.has_bias = geom, spec, bias is not None
        if second is None:
            return y
        return y, second

    @staticmethod
    def backward(ctx, gy, g_second=None):
        pre = None
        if ctx.groups > 1:
            x, w, sigma, wv, y, bsaved = ctx.saved_tensors
            pre = y
        elif ctx.act_z is not None:
            x, w, sigma, wv, pre = ctx.saved_tensors
            y = bsaved = None
        else:
            x, w, sigma, wv = ctx.saved_tensors
            y = bsaved = None
        if ctx.act_link is not None and g_second is not None and ctx.act_link.premultiplied:
            # the consumer's input-gradient launch has already multiplied by RootTanh'(y): g_second IS the gradient of y
            ctx.act_link.premultiplied = False
            if gy is not None:
                raise RuntimeError("a linked pre-activation has one consumer, its activation")
            gy = g_second.reshape(pre.shape) if g_second.shape != pre.shape else g_second
        elif ctx.act_z is not None and g_second is not None:
            # gradient through the activated second output joins the pre-activation's own (a norm's style scale, or none):
            # gy <- gy + RootTanh'(y) * g_second[:, z:]   - one launch (locate_act_rows_bwd), bit for bit autograd's sum
            Bn = pre.shape[0]
            Mn = pre.numel() // Bn          # (rows of a 1x1 map; a linked activation on a larger map whose consumer did not multiply)
            g2 = _chk(g_second, "activated output gradient")
            if g2.dim() != 2:
                g2 = g2.reshape(Bn, -1)
            if g2.stride(1) != 1:
                g2 = g2.contiguous()
            add = None if gy is None else _c(gy, "conv output gradient")
            total = torch.empty(pre.shape, dtype=torch.float32, device=pre.device)
            check(lib().locate_act_rows_bwd(_p(pre), g2.data_ptr() + 4 * ctx.act_z, g2.stride(0), _p(add), _p(total), Bn, Mn, _stream()),
                  "locate_act_rows_bwd")
            gy = total
        if gy is None:
            return (None,) * 13
        spec, garr = ctx.spec, _geom(ctx.geom)
        if ctx.guard is not None:
            ctx.guard[0].check(ctx.guard[1])
        gy = _dense(gy, "conv output gradient")
        amax_gy = _amax_of(gy)
        need_x, need_w, need_u, need_v, need_b = ctx.needs_input_grad[:5]
        gx = gw = gu = gb = None
        if need_x:
            gx = _conv_input_grad(gy, x, w, ctx.owner, spec, ctx.geom, garr, sigma, ctx.rt.precision, amax_gy, ctx.in_link)
        if need_w or need_u or need_v:
            rt = ctx.rt
            side = rt.weight_grad_stream
            if side is not None and spec.mode == "dense" and _flops(ctx.geom) < rt.weight_grad_side_min_flops:
                side = None              # a fork into the second stream costs more than a small launch gains there
            late = side is not None or rt.defer_finalisers     # the gradients bypass autograd: assigned at the end of the pass
            if side is None:
                sgw, sgu = _conv_weight_grad(rt, x, gy, y, bsaved, w, ctx.u, ctx.v, sigma, wv, spec, ctx.geom, garr, need_u, need_v,
                                             ctx.amax_x, amax_gy, ctx.owner if late else None)
            else:
                side.wait_stream(torch.cuda.current_stream())        # gy (and x) are complete on the pass's stream
                with torch.cuda.stream(side):
                    sgw, sgu = _conv_weight_grad(rt, x, gy, y, bsaved, w, ctx.u, ctx.v, sigma, wv, spec, ctx.geom, garr, need_u, need_v,
                                                 ctx.amax_x, amax_gy, ctx.owner if late else None)
                rt._side = side
            if late:
                rt._keep.append((x, gy, y, bsaved, w, sigma, wv))
                if need_w:
                    rt.late_grad(ctx.owner, sgw.view(ctx.owner.shape))
                if need_u:
                    rt.late_grad(ctx.u, sgu)
                rt._schedule_end()
            else:
                gw, gu = (sgw if need_w else None), sgu
        if ctx.has_bias and need_b:
            if ctx.rt.defer_finalisers and ctx.bias_param.is_leaf:
                late_gb = _grad_home(ctx.rt, ctx.bias_param, (gy.shape[1],))
                ctx.rt.queue_channel_sum(gy, late_gb)
                ctx.rt.late_grad(ctx.bias_param, late_gb.view(ctx.bias_param.shape))
            else:
                gb = _bias_grad(gy)
        # gv is assigned to v.grad by Runtime._finalize_dv at the end of this backward pass
        return gx, gw, gu, None, gb, None, None, None, None, None, None, None, None


def sn_conv(x, w_bar, u, v, bias, spec, sigma_wv=None, runtime=None, guard=None, out=None, act=None, in_link=None):
    """Spectral-normalised contraction.  Runs the power iteration unless (sigma, wv) of an already executed
    batched update is supplied (guard: see SNConvFn.forward).  out: view to write the result into (see _conv_apply).
    act: see SNConvFn.forward - returns (y, second output) then."""
    if sigma_wv is None:
        sigma_wv = sn_power_iteration(w_bar, u, v)
    sigma, wv = sigma_wv
    return SNConvFn.apply(x, w_bar, u, v, bias, sigma, wv, spec, runtime, guard, None if out is None else [out], act, in_link)
