"""Generator / Discriminator: drop-in nn.Modules with the reference's constructor behaviour, forward
signatures, `g_in` attribute and `state_dict()` key layout (reference libs/models.py:12-97), and the model
set-up helper of libs/utils.py:116-150.  Configuration comes from `locate_amd.config` instead of import-time
constants.  Seeded construction reproduces the reference's RNG draw order (tests/golden/g10_init_*.npz)."""
import torch
from torch import nn

from .config import get_default
from .nn import BlockBlock, DeepResidualConv, ResModule, Scale, SpectralNorm, _Bound, _identity
from . import arch, ops


def generator_features(cfg):
    return arch.generator_widths(cfg)


def discriminator_features(cfg):
    return arch.discriminator_widths(cfg)


class SpectralNormBatch:
    """All SpectralNorm layers of one network advanced by four launches (blockIdx.y = layer) at the start of a
    forward instead of four launches per layer.  Equivalent to the reference's per-layer update because W_bar, u
    and v of a layer only change at optimizer steps and every wrapped layer runs exactly once per model forward.

    Every iteration's results - sigma, 1/sigma per layer and W v - are kept until the backward of the forward that used them,
    so they go into a RING of `RING` result sets (one device table per set: the kernels write a set directly, nothing is
    copied), handed out in order.  A forward over k stacked calls takes k CONSECUTIVE sets and reads them as one strided
    view.  A set is overwritten RING iterations later; the backward of a forward whose set has been overwritten in the
    meantime raises (`check`), it never reads another iteration's sigma."""
    RING = 8

    def __init__(self, model):
        self.layers = [m for m in model.modules() if isinstance(m, SpectralNorm)]
        rounds = {m.power_iterations for m in self.layers}
        if len(rounds) > 1:
            raise NotImplementedError("the batched spectral norm advances all layers together: one power_iterations value per network")
        self.rounds = rounds.pop() if rounds else 1
        self._key = None
        self._tables = None
        self._meta = None
        self._next = 0
        self._gen = [0] * self.RING

    def _build(self, device):
        import struct
        from ._lib import lib
        L = lib()
        rec = L.locate_sn_table_record_bytes()
        assert rec == 80, rec
        scratch_sizes = []
        for sn in self.layers:
            w = sn.module.weight_bar
            h = w.shape[0]
            wd = w.numel() // h
            scratch_sizes.append((h, wd, (h + 63) // 64))
        total = sum(wd + h + nch * wd for h, wd, nch in scratch_sizes)
        self._scratch = torch.empty(total, dtype=torch.float32, device=device)
        self._meta = scratch_sizes
        self.max_h = max(h for h, _, _ in scratch_sizes)
        self.max_wd = max(wd for _, wd, _ in scratch_sizes)
        self._rec = rec
        self._struct = struct.Struct("<8Q4i")

    def _ensure_tables(self, dev):
        n = len(self.layers)
        key = tuple((sn.module.weight_bar.data_ptr(), sn.module.weight_u.data_ptr(), sn.module.weight_v.data_ptr())
                    for sn in self.layers)
        if key == self._key:
            return
        tot_h = sum(h for h, _, _ in self._meta)
        self._sig = torch.empty(self.RING, n, 2, dtype=torch.float32, device=dev)
        self._wvs = torch.empty(self.RING, tot_h, dtype=torch.float32, device=dev)
        self._tables = []
        base = self._scratch.data_ptr()
        for r in range(self.RING):
            buf = bytearray()
            off, hoff = 0, 0
            for i, (sn, (h, wd, nch)) in enumerate(zip(self.layers, self._meta)):
                m = sn.module
                t = base + 4 * off
                s = t + 4 * wd
                tp = s + 4 * h
                buf += self._struct.pack(m.weight_bar.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr(),
                                         self._sig[r].data_ptr() + 8 * i, self._wvs[r].data_ptr() + 4 * hoff, t, s, tp, h, wd, nch, 0)
                off += wd + h + nch * wd
                hoff += h
            host = torch.frombuffer(buf, dtype=torch.uint8)
            table = torch.empty(host.numel(), dtype=torch.uint8, device=dev)
            table.copy_(host)
            self._tables.append(table)
        self._key = key

    def run(self, count=1):
        """`count` power iterations for every layer, one after the other (u, v advanced in place), into `count` consecutive
        result sets; returns the list of their ring positions [(set, generation)]."""
        from ._lib import check, lib
        if not self.layers:
            return None
        dev = self.layers[0].module.weight_bar.device
        if self._meta is None:
            self._build(dev)
        self._ensure_tables(dev)
        if count > self.RING:
            raise ValueError("at most %d stacked calls" % self.RING)
        if self._next + count > self.RING:
            self._next = 0                       # k stacked calls read their sets as ONE strided view: no wrap-around inside
        out = []
        for _ in range(count):
            r = self._next
            self._next = (r + 1) % self.RING
            self._gen[r] += 1
            for _ in range(self.rounds):            # power_iterations rounds into the same set: its sigma is the last round's
                check(lib().locate_sn_power_iter_batched(self._tables[r].data_ptr(), len(self.layers), self.max_h, self.max_wd,
                                                         torch.cuda.current_stream().cuda_stream), "locate_sn_power_iter_batched")
            out.append((r, self._gen[r]))
        return out

    def check(self, sets):
        """Raises if any of the result sets [(set, generation)] has been overwritten by a later iteration."""
        for r, gen in sets:
            if self._gen[r] != gen:
                raise RuntimeError("spectral norm: the sigma of this forward was overwritten - more than %d forwards of the "
                                   "network ran before its backward (SpectralNormBatch.RING)" % self.RING)

    def assign(self, sets):
        """Hand the results of one iteration - or of k iterations for a forward over k stacked calls, in call order - to the
        layers (consumed by their next forward): per layer (sigma [2], wv [h]), resp. (sigma [k, 2], wv [k, h])."""
        hoff = 0
        k = len(sets)
        first = sets[0][0]
        consecutive = all(sets[j][0] == first + j for j in range(k))
        if k == 1:
            sig, wvs = self._sig[first], self._wvs[first]
        elif consecutive:
            sig, wvs = self._sig[first:first + k].permute(1, 0, 2), self._wvs[first:first + k]          # views: [n, k, 2], [k, H]
        else:
            idx = [r for r, _ in sets]
            sig, wvs = self._sig[idx].permute(1, 0, 2).contiguous(), self._wvs[idx]
        guard = (self, tuple(sets))
        for i, (sn, (h, _, _)) in enumerate(zip(self.layers, self._meta)):
            sn._pre = (sig[i], wvs[:, hoff:hoff + h] if k > 1 else wvs[hoff:hoff + h], guard)
            hoff += h


class _NetBase(nn.Module):
    batched_spectral_norm = False   # see SpectralNormBatch; off by default (per-layer update inside each layer)

    def set_precision(self, name):
        """"fp32": the reference's arithmetic (default; contractions as exact three-piece bf16 splits, six MFMAs per slice).
        "bf16": contraction operands rounded to bf16, one MFMA per slice, fp32 accumulation; everything else - storage,
        statistics, sigma, RootTanh, the optimizer - stays fp32 (the mixed-precision variant BASELINE.json configs[1] names).
        "fp8": both operands of every dense contraction as OCP e4m3 with per-tensor power-of-two scales on the fp8 matrix
        instruction (csrc/convfp8.hip; BASELINE.json configs[4]), fp32 accumulation, everything else fp32 as for "bf16"."""
        self.runtime.precision = {"fp32": 0, "f32": 0, "bf16": 1, "fp8": 3}[name]
        if name == "fp8":
            ops.AMAX_MIN_NUMEL[0] = 1          # every contraction operand's largest magnitude comes from its producer (process-wide)
        return self

    def adopt(self):
        """Gives the network its own ops.Runtime (stacked-call count, backward-pass deferrals) and points every layer that
        needs one at it - called at the end of the constructors.  Two networks therefore never share mutable host state."""
        object.__setattr__(self, "runtime", ops.Runtime())
        for m in self.modules():
            if isinstance(m, _Bound):
                object.__setattr__(m, "runtime", self.runtime)

    def _batch(self):
        if getattr(self, "_sn_batch", None) is None:
            object.__setattr__(self, "_sn_batch", SpectralNormBatch(self))
            object.__setattr__(self, "_sn_queue", [])
        return self._sn_batch

    def prefetch_spectral_norm(self, n_forwards):
        """Run the power iterations of the next `n_forwards` forwards now, in order, on the current stream.  The
        forwards themselves may then run concurrently on different streams: each consumes one queued result, exactly
        the (u, v, sigma) sequence the reference produces by iterating at the start of every forward."""
        sets = self._batch().run(n_forwards)
        if sets is not None:
            self._sn_queue.extend(sets)

    def _sn_prologue(self, stacked=1):
        ops.AMAX.new_pass()          # this forward's largest-magnitude words come from a block of their own
        if stacked > 1 and not self.batched_spectral_norm:
            raise RuntimeError("a forward over stacked calls needs batched_spectral_norm = True")
        if stacked > 4:
            raise ValueError("at most 4 stacked calls per forward (the kernels' per-call 1/sigma groups and the deferred finalisers' records hold four)")
        if self.batched_spectral_norm:
            b = self._batch()
            if not b.layers:
                return
            sets = [self._sn_queue.pop(0) for _ in range(min(stacked, len(self._sn_queue)))]
            if len(sets) < stacked:
                sets += b.run(stacked - len(sets))
            b.assign(sets)


class Generator(_NetBase):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or get_default()
        self.cfg = cfg
        clayers = cfg.layers - 1
        strides = [cfg.g_stride] * clayers
        feature_list = generator_features(cfg)
        self.input_block = _identity                      # START_LAYER = 0 (config.py:39)
        self.conv_block = BlockBlock(len(strides), 2, feature_list, strides, True, True, cfg=cfg)
        self.out_conv = DeepResidualConv(self.conv_block.out_features, 3, False, 1, False, 2, 1, cfg=cfg)
        self.g_in = feature_list[0]
        # a plain tensor like in the reference (models.py:59): not a Parameter, not in state_dict()
        self.noise = torch.randn(1, cfg.input_vector_z, 2, 2)
        self.adopt()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.noise = fn(self.noise)                       # the reference creates it on DEVICE; follow .to()/.cuda()
        return out

    def forward(self, function_input):
        self._sn_prologue()
        # the constant noise map repeated over the batch (models.py:62): a copy that only changes with the map or the batch size
        key = (function_input.size(0), self.noise.data_ptr(), self.noise._version)
        cached = self.__dict__.get("_noise_batch")
        if cached is None or cached[0] != key:
            cached = (key, self.noise.expand(function_input.size(0), -1, -1, -1).contiguous())
            self.__dict__["_noise_batch"] = cached
        expanded_noise = cached[1]
        conv_out = self.conv_block(self.input_block(expanded_noise), function_input)
        return ops.tanh(self.out_conv(conv_out))


class Discriminator(_NetBase):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or get_default()
        self.cfg = cfg
        clayers = cfg.layers - 1
        strides = [cfg.d_stride] * clayers
        feature_list = discriminator_features(cfg)
        stem = ResModule(Scale(3, feature_list[0], 2, False), DeepResidualConv(3, feature_list[0], False, 2, False, 2, 1, cfg=cfg))
        block_block = BlockBlock(len(strides), cfg.image_size // 2, feature_list, strides, False, cfg=cfg)
        head = DeepResidualConv(block_block.out_features, 1, False, 1, False, 2, 1, cfg=cfg)   # END_LAYER = 1
        self.main = nn.Sequential(stem, block_block, head)
        self.adopt()

    def forward(self, function_input, stacked=1, cut_after=None):
        """stacked = k > 1: `function_input` stacks k independent calls along the batch (k equal slices); the result
        equals running them one after the other (k power iterations in order, per-call InPlaceNorm statistics), in
        one pass over the network - the three D passes of the reference's D-step (main.py:149-152).

        cut_after = j: the same values, but as TWO autograd graphs cut behind block j - 1 of the stack, so that the backward
        pass can be run (and captured, and its gradients sent to the other ranks) in two segments: first
        `backward(out, g)` through the head and the blocks j.., then `backward(*self.take_cut())` through the blocks ..j - 1
        and the stem.  The deep, narrow-map segment holds ~95 % of the discriminator's parameters, the other one most of
        its time: the first segment's all-reduce hides behind the second segment's backward (locate_amd.parallel)."""
        self._sn_prologue(stacked)
        with self.runtime.stacked_calls(stacked):
            if cut_after is None:
                return self.main(function_input)
            stem, stack, head = self.main
            if stack.mul_blocks or not 0 < cut_after <= len(stack.blocks):
                raise ValueError("cut_after must name a block boundary of the discriminator stack")
            h = stem(function_input)
            for block in stack.blocks[:cut_after]:
                h = block(h)
            h_in = h.detach().requires_grad_(True)
            out = h_in
            for block in stack.blocks[cut_after:]:
                out = block(out)
            object.__setattr__(self, "_cut", (h, h_in))
            return head(out)

    def take_cut(self):
        """(tensor, gradient) for the second backward segment of a cut forward: call after the first segment's backward."""
        h, h_in = self._cut
        object.__setattr__(self, "_cut", None)
        return h, h_in.grad

    def segment_parameters(self, cut_after):
        """The parameters of the two backward segments, in the order their gradients become complete."""
        stem, stack, head = self.main
        late = [p for m in list(stack.blocks[cut_after:]) + [head] for p in m.parameters()]
        early = [p for m in [stem] + list(stack.blocks[:cut_after]) for p in m.parameters()]
        return [late, early]


def init(module):
    """Per-module initialiser for `model.apply` (libs/utils.py:116-130), as it behaves when applied before the first
    forward - which is when `get_model` applies it: a module whose class name contains "norm" gets its weight drawn from
    U(0.998, 1.002), any other module that owns a `weight` an orthogonal one, and every `bias` is zeroed.  Spectral-norm
    wrapped convs / linears expose no `weight` then (it has become `weight_bar`), so in the networks of this package only
    InPlaceNorm weights and the few biases are touched; the wrapped weights keep torch's default initialisation."""
    weight = getattr(module, "weight", None)
    if torch.is_tensor(weight):
        if "norm" in type(module).__name__.lower():
            nn.init.uniform_(weight.data, 0.998, 1.002)
        else:
            nn.init.orthogonal_(weight.data)
    bias = getattr(module, "bias", None)
    if torch.is_tensor(bias):
        nn.init.constant_(bias.data, 0)


def get_model(model, learning_rate, device, cfg=None):
    """libs/utils.py:146-150: move, init, build the Nadam optimizer."""
    from .optim import Nadam
    cfg = cfg or getattr(model, "cfg", None) or get_default()
    model.apply(init)          # RNG draws happen on the CPU generator, like the CPU reference
    model = model.to(device)
    opt = Nadam(model.parameters(), lr=learning_rate, betas=(cfg.beta1, cfg.beta2))
    return model, opt


def parameter_count(net):
    return sum(p.numel() for p in net.parameters() if p.requires_grad)
