"""Whole-step hipGraph capture.  One G+D iteration is ~1100 launches (the reference's Python issues several thousand
ATen calls); replaying them as graphs removes the host from the loop - the MI355X-native replacement
for a tracing compiler.  Everything the step mutates (weights, spectral-norm u/v, Nadam moments and schedule
counters) lives in device memory at fixed addresses, so one replay == one more training iteration.

The iteration is captured as FOUR graphs sharing one memory pool (D forward/backward, D Nadam, G
forward/backward, G Nadam): the optimizer's device tables hold the addresses of the gradient buffers, which
only exist once the preceding backward has been captured, and building them needs pinned-host staging, which is
not allowed inside a capture - so they are built eagerly between two captures."""
import torch


class GraphedTrainStep:
    def __init__(self, step, latent, real, aug, warmup=2):
        self.step = step
        self.inputs = tuple(t.clone() for t in (latent, real, aug))
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # eager iterations first: lazily created state (optimizer moments, spectral-norm tables) must exist and
            # the discriminator's u/v must already be trainable (reference main.py:172) so that the captured
            # iteration has the steady-state autograd structure
            for _ in range(max(warmup, 2)):
                step(*self.inputs)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        lat, real_, aug_ = self.inputs
        self._params = [p for net in (step.gen, step.dis) for p in net.parameters()]
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs = []
        self.outputs = {}

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.pool):
                out = fn()
            self.graphs.append(g)
            if out:
                self.outputs.update(out)

        # data parallel: the collectives stay OUTSIDE the graphs (replay: backward graph -> eager bucketed all-reduce ->
        # optimizer graph), so nothing RCCL-related is ever captured
        self.reducers = (step.reducer_d, step.reducer_g)
        step.reducer_d = step.reducer_g = None
        try:
            capture(lambda: step.d_forward_backward(lat, real_, aug_))
            self._prime(step.dis_opt)
            capture(step.d_optimizer)
            capture(lambda: step.g_forward_backward(lat))
            self._prime(step.gen_opt)
            capture(step.g_optimizer)
        finally:
            step.reducer_d, step.reducer_g = self.reducers

    @staticmethod
    def _prime(opt):
        """Build the optimizer's device tables for the gradient buffers the captured backward just allocated."""
        for group in opt.param_groups:
            plist = [p for p in group["params"] if p.grad is not None]
            if plist:
                opt._table(plist)
        torch.cuda.synchronize()

    def replay(self, latent=None, real=None, aug=None):
        for dst, src in zip(self.inputs, (latent, real, aug)):
            if src is not None:
                dst.copy_(src)
        red_d, red_g = self.reducers
        self.graphs[0].replay()
        if red_d is not None:
            red_d.reduce_now()
        self.graphs[1].replay()
        self.graphs[2].replay()
        if red_g is not None:
            red_g.reduce_now()
        self.graphs[3].replay()
        # the replayed optimizer kernels changed the weights behind Python's back: bump the version counters so
        # that any later EAGER use (sampling, evaluation) re-packs its weight panels instead of trusting the cache
        torch._C._increment_version(self._params)
        return self.outputs
