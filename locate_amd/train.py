"""One G+D training iteration: the loop body of the reference's main.py:142-172 (miniter = 1), with the loss
glue (libs/utils.py:133-134 hinge, libs/grad_penalty.py:1-2 consistency penalty) computed by one small HIP
kernel per phase that also emits the gradients w.r.t. the discriminator outputs.

Sequencing kept from the reference, including its quirks:
  * D-step: G forward (result detached), `dis.zero_grad()`, three D forwards in the order real / fake /
    augmented (every forward advances the spectral-norm u, v), ONE backward over the three graphs, D Nadam step;
  * G-step: `dis.requires_grad_(False)`, MINIBATCHES x {`gen.zero_grad()`, forward, backward} on the SAME noise
    (only the last pass's gradients survive), G Nadam step, `dis.requires_grad_(True)` - which also makes D's
    spectral-norm u, v trainable from the second D-step on.
With `world_size > 1` the gradients are averaged across ranks by `locate_amd.parallel.GradAllReducer`
(RCCL all-reduce, bucketed, on a side stream) before each optimizer step.
"""
import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def d_loss(d_true, d_fake, d_aug, gamma=100.0):
    """Returns (losses[3] = {d_error, penalty, total}, g_true, g_fake, g_aug); the three gradients are the rows of one
    [3, B] tensor (g_true._base)."""
    B = d_true.numel()
    dev = d_true.device
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    g = torch.empty(3, B, dtype=torch.float32, device=dev)
    t, f, a = (x.detach().contiguous() for x in (d_true, d_fake, d_aug))
    check(lib().locate_d_loss(t.data_ptr(), f.data_ptr(), a.data_ptr(), B, float(gamma), losses.data_ptr(), g[0].data_ptr(),
                              g[1].data_ptr(), g[2].data_ptr(), _stream()), "locate_d_loss")
    return losses, g[0], g[1], g[2]


def g_loss(d_fake):
    B = d_fake.numel()
    loss = torch.empty(1, dtype=torch.float32, device=d_fake.device)
    g = torch.empty(B, dtype=torch.float32, device=d_fake.device)
    f = d_fake.detach().contiguous()
    check(lib().locate_g_loss(f.data_ptr(), B, loss.data_ptr(), g.data_ptr(), _stream()), "locate_g_loss")
    return loss, g


class TrainStep:
    def __init__(self, gen, dis, gen_opt, dis_opt, penalty_gamma=100.0, minibatches=1, reducer_g=None, reducer_d=None,
                 concurrent_d=False, stacked_d=None, overlap_wgrad=False, d_cut=None):
        self.gen, self.dis, self.gen_opt, self.dis_opt = gen, dis, gen_opt, dis_opt
        # d_cut = j (data parallel): the stacked D-step's backward runs in two segments cut behind block j - 1 of the
        # discriminator stack (Discriminator.forward(cut_after=j)); the reducer sends the first segment's gradients - the
        # deep layers, ~95 % of D's parameters - while the second segment's backward runs.  Same values as one pass.
        self.d_cut = d_cut
        self._d_pending = None
        # overlap_wgrad: the weight gradients of a backward pass run on a second stream beside its chain of input gradients
        # (ops.Runtime.weight_grads_on); same values, the parameters' .grad are complete when backward() returns.  Measured
        # neutral under hipGraph replay while the step's kernels were larger (12.76 vs 12.74 ms at config 2), -0.25 ms
        # (11.24 -> 10.99) since the small contractions got their own kernels and leave most of the chip idle: bench.py
        # turns it on.  Off by default here: slower eagerly (more host work per layer).  With a data-parallel reducer the
        # gradients that bypass autograd's accumulation are exchanged by the reducer's finish() (eager) or by the segment's
        # launch_group() (graph replay) - tests/test_gpu_dp.py runs both forms with the flag on.
        self.overlap_wgrad = bool(overlap_wgrad)
        self._wgrad_side = None
        # stacked_d: the D-step's three discriminator passes (real / fake / augmented) run as ONE pass over a [3B] batch
        # with per-call InPlaceNorm statistics and per-call spectral-norm sigma (three power iterations up front, in the
        # reference's order) - same values, a third of the launches, no cross-graph gradient accumulation.
        # Default: on whenever the discriminator uses the batched spectral norm.
        self.stacked_d = (bool(getattr(dis, "batched_spectral_norm", False)) and not concurrent_d) if stacked_d is None \
            else bool(stacked_d)
        self.penalty_gamma = penalty_gamma
        self.minibatches = minibatches
        self.reducer_g, self.reducer_d = reducer_g, reducer_d
        for red in (reducer_g, reducer_d):
            if red is not None:
                red.make_homes()         # parameter gradients are produced inside their buckets' resident buffers (parallel.py)
        # concurrent_d: run the D-step's three discriminator passes (and the generator forward feeding the second)
        # on three HIP streams.  Most of the step's ~1900 launches are tiny, launch-latency-bound kernels; three
        # independent chains side by side fill the chip where one cannot.  Needs the batched spectral norm (the power
        # iterations of the three forwards are done first, in the reference's order real / fake / augmented).
        self.concurrent_d = concurrent_d
        self._streams = None

    def _backward(self, outputs, grads):
        import contextlib
        from . import ops
        runtimes = [getattr(net, "runtime", ops.DEFAULT_RUNTIME) for net in (self.gen, self.dis)]
        ops.reset_backward_state(*runtimes)
        # A data-parallel reducer driven by post-accumulate hooks (eager mode) needs every gradient complete when autograd
        # hands it over, so the layers then finish their own gradients; otherwise the small per-layer finalisers are batched
        # at the end of the pass (ops.Runtime.defer_finalisers).  Under hipGraph capture the reducers are detached
        # (graph.py) and the exchange is launched between the graphs: deferral stays on there.
        defer = self.reducer_g is None and self.reducer_d is None
        with contextlib.ExitStack() as stack:
            for rt in runtimes:
                stack.enter_context(rt.finalisers_deferred(defer))
            if self.overlap_wgrad:
                if self._wgrad_side is None:
                    self._wgrad_side = torch.cuda.Stream()
                for rt in runtimes:
                    stack.enter_context(rt.weight_grads_on(self._wgrad_side))
            return torch.autograd.backward(outputs, grads)

    # ---- the four phases of one iteration (kept separate so that each can be its own hipGraph) --------------
    def _d_forwards_concurrent(self, latent, real, aug):
        from . import ops
        gen, dis = self.gen, self.dis
        if not (gen.batched_spectral_norm and dis.batched_spectral_norm):
            raise RuntimeError("concurrent_d needs batched_spectral_norm on both networks")
        main = torch.cuda.current_stream()
        if self._streams is None:
            self._streams = [torch.cuda.Stream() for _ in range(3)]
        s_real, s_fake, s_aug = self._streams
        # everything the three chains share is brought up to date on the main stream first
        ops.refresh_panels([m.module.weight_bar for m in dis._batch().layers])
        gen.prefetch_spectral_norm(1)
        dis.prefetch_spectral_norm(3)              # sigma_1 (real), sigma_2 (fake), sigma_3 (augmented)
        for s in self._streams:
            s.wait_stream(main)
        with torch.cuda.stream(s_real):
            d_true = dis(real)                     # consumes sigma_1
        with torch.cuda.stream(s_fake):
            with torch.no_grad():
                generated = gen(latent)
            d_fake = dis(generated)                # consumes sigma_2
        with torch.cuda.stream(s_aug):
            d_aug = dis(aug)                       # consumes sigma_3
        for s in self._streams:
            main.wait_stream(s)
        for t_ in (d_true, d_fake, d_aug, generated):
            t_.record_stream(main)
        return generated, d_true, d_fake, d_aug

    def begin_iteration(self, device):
        """What both generator passes of an iteration (main.py:146 and :164: same latent, same weights) have in common, run
        ahead of them so that they can execute CONCURRENTLY on two streams (graph.py): the iteration's zeroed block of
        largest-magnitude words, and the generator's two power iterations in the reference's order - each pass then consumes
        its queued (sigma, W v), exactly the sequence two forwards one after the other produce."""
        from . import ops
        ops.AMAX.new_step(device)
        if getattr(self.gen, "batched_spectral_norm", False) and self.minibatches == 1:
            self.gen.prefetch_spectral_norm(2)

    def d_generate(self, latent):
        """main.py:146: the D-step's generator pass (the reference builds a graph and drops it with .detach())."""
        from . import ops
        ops.AMAX.new_step()            # the iteration's first phase: one zeroed block of largest-magnitude words for all its passes
        with torch.no_grad():
            return self.gen(latent)

    def d_forward_backward(self, latent, real, aug, generated=None, segment=None):
        """`generated`: result of d_generate(latent) when that phase ran on its own (graph.py), else computed here.
        segment (only with d_cut): 0 = forward, loss and the first backward segment; 1 = the second backward segment;
        None = both.  graph.py captures the two segments as separate hipGraphs and launches the first segment's
        all-reduce between their replays."""
        gen, dis = self.gen, self.dis
        if segment == 1:
            h, gh = dis.take_cut()
            self._backward([h], [gh])
            out, self._d_pending = self._d_pending, None
            return out
        dis.zero_grad()                            # main.py:148
        if self.stacked_d and getattr(dis, "batched_spectral_norm", False):
            if generated is None:
                generated = self.d_generate(latent)
            B = real.shape[0]
            cut = self.d_cut
            d_all = dis(torch.cat([real, generated, aug], dim=0), stacked=3, cut_after=cut)   # :149, :150, grad_penalty.py:2
            d_true, d_fake, d_aug = d_all[:B], d_all[B:2 * B], d_all[2 * B:]
            losses, g_t, _, _ = d_loss(d_true, d_fake, d_aug, self.penalty_gamma)
            out = {"d_error": losses[0], "penalty": losses[1], "d_true": d_true.detach().view(-1),
                   "d_gen": -d_fake.detach().view(-1), "generated": generated}
            if self.reducer_d is not None:
                self.reducer_d.begin()
            self._backward([d_all], [g_t._base.view_as(d_all)])                 # :156
            if cut is not None:
                if segment == 0:
                    self._d_pending = out
                    return out
                h, gh = dis.take_cut()
                self._backward([h], [gh])
            if self.reducer_d is not None:
                self.reducer_d.finish()
            return out
        if self.d_cut is not None:
            raise RuntimeError("d_cut needs the stacked D-step (batched spectral norm, no concurrent_d)")
        if self.concurrent_d:
            generated, d_true, d_fake, d_aug = self._d_forwards_concurrent(latent, real, aug)
        else:
            if generated is None:
                generated = self.d_generate(latent)
            d_true = dis(real)                     # :149
            d_fake = dis(generated)                # :150 (the reference negates it; the loss kernel takes it raw)
            d_aug = dis(aug)                       # grad_penalty.py:2
        losses, g_t, g_f, g_a = d_loss(d_true, d_fake, d_aug, self.penalty_gamma)
        if self.reducer_d is not None:
            self.reducer_d.begin()
        self._backward([d_true, d_fake, d_aug], [g_t.view_as(d_true), g_f.view_as(d_fake), g_a.view_as(d_aug)])  # :156
        if self.reducer_d is not None:
            self.reducer_d.finish()
        return {"d_error": losses[0], "penalty": losses[1], "d_true": d_true.detach().view(-1),
                "d_gen": -d_fake.detach().view(-1), "generated": generated}

    @staticmethod
    def _repack(net):
        from . import ops
        ops.refresh_panels([p for p in net.parameters() if "_locate_panels" in p.__dict__])

    def d_optimizer(self):
        self.dis_opt.step()                        # :159
        self._repack(self.dis)                     # all of D's weight panels in one launch

    def g_forward_backward(self, latent):
        gen, dis = self.gen, self.dis
        dis.requires_grad_(False)                  # main.py:161
        try:
            for _ in range(self.minibatches):      # :162-169 (same noise, zero_grad inside the loop)
                gen.zero_grad()
                fake = gen(latent)
                d_out = dis(fake)
                loss, g = g_loss(d_out)
                if self.reducer_g is not None:
                    self.reducer_g.begin()
                self._backward([d_out], [g.view_as(d_out)])
                if self.reducer_g is not None:
                    self.reducer_g.finish()
        finally:
            dis.requires_grad_(True)               # :172 (u, v included; the reference does it after GEN_OPTIM.step())
        return {"g_error": loss[0], "fake": fake.detach()}

    # the G-step in two phases (minibatches == 1): its generator pass depends on nothing the D-step changes, so graph.py
    # replays it on a second stream while the D-step's discriminator work is still running
    def g_forward(self, latent):
        self.gen.zero_grad()                       # main.py:163
        return self.gen(latent)

    def g_backward(self, fake):
        dis = self.dis
        dis.requires_grad_(False)                  # main.py:161
        try:
            d_out = dis(fake)
            loss, g = g_loss(d_out)
            if self.reducer_g is not None:
                self.reducer_g.begin()
            self._backward([d_out], [g.view_as(d_out)])   # :169
            if self.reducer_g is not None:
                self.reducer_g.finish()
        finally:
            dis.requires_grad_(True)               # :172
        return {"g_error": loss[0], "fake": fake.detach()}

    def g_optimizer(self):
        from . import ops
        self.gen_opt.step()                        # :171
        self._repack(self.gen)
        ops.AMAX.end_step()

    def d_step(self, latent, real, aug):
        out = self.d_forward_backward(latent, real, aug)
        self.d_optimizer()
        return out

    def g_step(self, latent):
        out = self.g_forward_backward(latent)
        self.g_optimizer()
        return out

    def __call__(self, latent, real, aug):
        out = self.d_step(latent, real, aug)
        out.update(self.g_step(latent))
        return out


class TrainLoop:
    """The reference's batch schedule around the step (main.py:139-172), quirks included:
      * every batch i = 1, 2, ... runs the D-step's forward/backward with `dis.zero_grad()` first - so with
        `miniter` > 1 the discriminator optimizer, which only steps when i % miniter == 0, sees the gradients of the
        LAST batch only (nothing accumulates);
      * the G-step (MINIBATCHES passes on the same noise, `gen.zero_grad()` inside the loop, then one optimizer step)
        follows every `diters`-th discriminator step: (i // miniter) % diters == 0;
      * `minibatch_function(epoch) = (epoch + 1) * MINIBATCHES` gives the reference's `miniter` per epoch
        (libs/config.py:20-30); BASELINE's metric uses miniter = MINIBATCHES = DITERS = 1."""

    def __init__(self, step, miniter=1, diters=1):
        if miniter < 1 or diters < 1:
            raise ValueError("miniter and diters must be >= 1")
        self.step, self.miniter, self.diters = step, int(miniter), int(diters)
        self.i = 0

    def iteration(self, latent, real, aug):
        """One batch of the loop; returns the D-step record, plus the G-step record when one ran."""
        self.i += 1
        i, step = self.i, self.step
        out = step.d_forward_backward(latent, real, aug)          # main.py:146-156
        if i % self.miniter == 0:                                  # :158
            step.d_optimizer()                                     # :159
            if (i // self.miniter) % self.diters == 0:             # :160
                out.update(step.g_forward_backward(latent))        # :161-169
                step.g_optimizer()                                 # :171
        return out
