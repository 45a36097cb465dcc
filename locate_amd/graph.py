"""Whole-step hipGraph capture.  One G+D iteration is ~1100 launches (the reference's Python issues several thousand
ATen calls); replaying them as graphs removes the host from the loop - the MI355X-native replacement
for a tracing compiler.  Everything the step mutates (weights, spectral-norm u/v, Nadam moments and schedule
counters) lives in device memory at fixed addresses, so one replay == one more training iteration.

The iteration is captured as SIX graphs (D-step generator pass | D forward/backward | D Nadam | G-step generator
pass | D(fake) + backward | G Nadam; four when the overlapped schedule is off): the optimizer's device tables hold
the addresses of the gradient buffers, which only exist once the preceding backward has been captured, and building
them needs pinned-host staging, which is not allowed inside a capture - so they are built eagerly between two
captures.  Concurrency exists only BETWEEN graphs replayed on different streams (fork/join branches inside one
captured graph run one after the other on this runtime, tools/graph_branch_probe.py), hence the separate graph for
the G-step's generator pass."""
import torch


class GraphedTrainStep:
    """SIDE EFFECT OF CONSTRUCTION: `max(warmup, 2)` REAL training iterations are run eagerly on the sample batch before the
    capture - lazily created state (optimizer moments and device tables, spectral-norm tables, arrival counters) must exist,
    and the discriminator's u / v must already be trainable (reference main.py:172) so that the captured iteration has the
    steady-state autograd structure.  Weights, u / v and the Nadam states therefore ADVANCE by that many steps; wrapping a
    run that is already in progress adds them on one batch (pass that run's own current batch).  The capture itself
    executes nothing.  After replays, use the networks eagerly as usual: replay() bumps the parameters' version counters so
    that weight panels are re-packed."""

    def __init__(self, step, latent, real, aug, warmup=2, overlap=None):
        """overlap (default: on for minibatches == 1 without the three-stream D-step): the iteration is captured as SIX
        graphs - D-step generator pass | D forward/backward | D Nadam | G-step generator pass | D(fake) + backward | G
        Nadam - and the G-step's generator pass is replayed on a second stream as soon as the D-step's generator pass is
        done: it reads nothing the D-step writes (G's weights, panels and spectral-norm state only), and most kernels of
        both chains are too small to fill the chip on their own.  It gets a memory pool of its own, because graphs that
        run concurrently must not share recycled allocations."""
        self.step = step
        self.overlap = (step.minibatches == 1 and not step.concurrent_d) if overlap is None else bool(overlap)
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

        cap_stream = torch.cuda.Stream()     # ONE capture stream: autograd replays a node's backward on its forward's stream
        self.pool_b = torch.cuda.graph_pool_handle()
        import os
        self.early_second_pass = os.environ.get("LOCATE_G2_EARLY", "0") == "1"
        prio = os.environ.get("LOCATE_SIDE_PRIORITY")
        self._side = torch.cuda.Stream(priority=int(prio)) if prio is not None else torch.cuda.Stream()
        self._e0, self._e1, self._e2, self._es, self._eb = (torch.cuda.Event() for _ in range(5))
        self._keep = []
        self.sn_graph = None
        self.d_tail = None
        self.begin_graph = None

        # With a process group alive its watchdog thread polls the events of earlier collectives (hipEventQuery) at any time;
        # under the default "global" capture mode that call, made by ANOTHER thread while this one captures, is an error that
        # kills the process (found by the one-rank RCCL rehearsal, bench.py LOCATE_DP_FORCE=1).  "thread_local" restricts
        # only the capturing thread; the captures themselves contain no collective (the exchange runs between the graphs).
        import torch.distributed as dist
        self._capture_mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"

        def graph_ctx(g, pool):
            return torch.cuda.graph(g, pool=pool, stream=cap_stream, capture_error_mode=self._capture_mode)

        def capture(fn, pool=None):
            g = torch.cuda.CUDAGraph()
            with graph_ctx(g, pool or self.pool):
                out = fn()
            self.graphs.append(g)
            if isinstance(out, dict):
                self.outputs.update(out)
            return out

        # data parallel: the collectives stay OUTSIDE the graphs (replay: backward graph -> eager bucketed all-reduce ->
        # optimizer graph), so nothing RCCL-related is ever captured
        self.reducers = (step.reducer_d, step.reducer_g)
        step.reducer_d = step.reducer_g = None
        try:
            if self.overlap:
                # LOCATE_G2_EARLY=1 (experiment, off by default: measured +0.10 ... +0.15 ms per step, same-box A/B in
                # profiles/r03_two_generator_passes_ab.txt).  Ahead of both generator passes: the iteration's zeroed absmax
                # block and G's two power iterations (a small graph of its own).  The two passes then depend on nothing of each
                # other and are replayed CONCURRENTLY - the G-step's pass on the second stream from the start of the iteration
                # instead of after the D-step's pass.  The second pass's contractions take their own split-K arrival counters
                # (ops.counter_lane).  Bit-identical results (test_graph_replay_equals_eager passes either way); slower because
                # the D-step's pass is on the critical path and loses more to the contention than the later overlap with the
                # discriminator work gives back.
                self.begin_graph = None
                if self.early_second_pass and getattr(step.gen, "batched_spectral_norm", False) and step.minibatches == 1:
                    self.begin_graph = torch.cuda.CUDAGraph()
                    with graph_ctx(self.begin_graph, self.pool):
                        step.begin_iteration(lat.device)
                    from . import ops as _ops
                    self._keep.append(_ops.AMAX.block)
                generated = capture(lambda: step.d_generate(lat))                              # 0
                # The discriminator's three power iterations of the D-step (sigma for real / fake / augmented, main.py:149-152)
                # read only what the PREVIOUS iteration left behind (D's weights after its optimizer step, u / v after the
                # G-step's discriminator pass), so they are a graph of their own, replayed on the second stream beside the
                # generator pass; the D-step's forward consumes the queued results.  Their output tensors are kept alive for
                # the life of the graphs: they live in the second pool, which the G-step's generator pass also allocates from.
                if step.stacked_d and getattr(step.dis, "batched_spectral_norm", False):
                    self.sn_graph = torch.cuda.CUDAGraph()
                    with graph_ctx(self.sn_graph, self.pool_b):
                        step.dis.prefetch_spectral_norm(3)
                if step.d_cut is not None:
                    # data parallel: the D-step's backward as two graphs; replay() launches the deep segment's all-reduce
                    # between them, so that it runs beside the second segment
                    capture(lambda: step.d_forward_backward(lat, real_, aug_, generated=generated, segment=0))   # 1
                    self.d_tail = torch.cuda.CUDAGraph()
                    with graph_ctx(self.d_tail, self.pool):
                        step.d_forward_backward(lat, real_, aug_, segment=1)
                else:
                    capture(lambda: step.d_forward_backward(lat, real_, aug_, generated=generated))  # 1
                self._prime(step.dis_opt)
                capture(step.d_optimizer)                                                      # 2
                if self.begin_graph is not None:
                    from . import ops as _ops
                    with _ops.counter_lane(1):
                        fake = capture(lambda: step.g_forward(lat), pool=self.pool_b)          # 3 (second stream)
                else:
                    fake = capture(lambda: step.g_forward(lat), pool=self.pool_b)              # 3 (second stream)
                capture(lambda: step.g_backward(fake))                                         # 4
                self._prime(step.gen_opt)
                capture(step.g_optimizer)                                                      # 5
            else:
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
        if self.overlap:
            main = torch.cuda.current_stream()
            g = self.graphs
            self._e0.record(main)                  # everything the previous iteration wrote is complete here
            if self.begin_graph is not None:
                self.begin_graph.replay()          # absmax block zeroed, G's two power iterations queued
            self._eb.record(main)
            g[0].replay()                          # D-step generator pass
            self._e1.record(main)
            with torch.cuda.stream(self._side):
                if self.sn_graph is not None:
                    self._side.wait_event(self._e0)
                    self.sn_graph.replay()         # D's three power iterations, beside the generator pass
                    self._es.record(self._side)
                # the G-step's generator pass: beside the D-step's one when both only consume queued power iterations, else
                # behind it (its own power iteration continues the first pass's)
                self._side.wait_event(self._eb if (self.begin_graph is not None and self.early_second_pass) else self._e1)
                g[3].replay()                      # ... and concurrent with the discriminator work below
                self._e2.record(self._side)
            if self.sn_graph is not None:
                main.wait_event(self._es)
            if red_d is not None:
                red_d.begin_replay()
            g[1].replay()
            if self.d_tail is not None:
                if red_d is not None:
                    red_d.launch_group(0)          # deep segment's gradients: on the wire beside the second segment
                self.d_tail.replay()
            if red_d is not None:
                red_d.finish()                     # remaining buckets, then the compute stream joins the side stream
            g[2].replay()
            main.wait_event(self._e2)
            if red_g is not None:
                red_g.begin_replay()
            g[4].replay()
            if red_g is not None:
                red_g.finish()
            g[5].replay()
        else:
            self.graphs[0].replay()
            if red_d is not None:
                red_d.reduce_now(replay=True)
            self.graphs[1].replay()
            self.graphs[2].replay()
            if red_g is not None:
                red_g.reduce_now(replay=True)
            self.graphs[3].replay()
        # the replayed optimizer kernels changed the weights behind Python's back: bump the version counters so
        # that any later EAGER use (sampling, evaluation) re-packs its weight panels instead of trusting the cache
        torch._C._increment_version(self._params)
        return self.outputs
