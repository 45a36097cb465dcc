"""Whole-step hipGraph capture.  One G+D iteration is ~3000 small launches (the reference's Python issues about
as many ATen calls); replaying them as ONE graph removes the host from the loop - the MI355X-native
replacement for a tracing compiler.  Everything the step mutates (weights, spectral-norm u/v, Nadam moments and
schedule counters) lives in device memory at fixed addresses, so replay == one more training iteration."""
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
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = step(*self.inputs)

    def replay(self, latent=None, real=None, aug=None):
        for dst, src in zip(self.inputs, (latent, real, aug)):
            if src is not None:
                dst.copy_(src)
        self.graph.replay()
        return self.outputs
