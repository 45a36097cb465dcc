"""Data parallelism for the G+D step: one process per GPU, identical replicas, ONE exchange per optimizer
step - the mean of every non-None parameter gradient over the ranks.

The reference has no distributed code at all; this is new capability, designed for the MI355X node:
  * `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for the tests;
  * gradients are packed into a few large flat buckets (default 32 MiB: the xGMI mesh is point-to-point, so few
    large collectives beat many small ones) in REVERSE parameter order, which is roughly the order in which the
    backward pass produces them;
  * a bucket's all-reduce is issued from a post-accumulate-grad hook as soon as its last gradient has been
    accumulated, on a side stream, so it overlaps the rest of the backward pass; `finish()` joins the side
    stream, flushes buckets that did not fill (parameters without gradient this step: the generator's unused
    `i_norm.weight`s, the discriminator's u/v before they become trainable) and scatters the averaged values
    back into the `.grad` tensors;
  * InPlaceNorm statistics stay per replica (no second collective); spectral-norm u/v need no traffic because
    they are a deterministic function of identical (W_bar, u).
Parity statement: the averaged gradient equals the mean of the gradients the reference would compute on each
rank's shard independently (tests/test_parallel_gloo.py)."""
import torch
import torch.distributed as dist


def broadcast_module_state(module, src=0, extra_tensors=()):
    """Make every rank's parameters, buffers and extra tensors (Generator.noise!) equal to rank `src`'s."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()) + list(extra_tensors):
            dist.broadcast(t.data, src)


class GradAllReducer:
    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, overlap=True, late=None, groups=None, reduce_op=None,
                 force=False, solo_bytes=0):
        """`late`: predicate (or collection) of parameters whose .grad is only assigned at the very end of the backward
        pass, outside autograd's accumulation (the spectral-norm v vectors, ops.Runtime._finalize_dv).  They get buckets
        of their own, all-reduced by finish(), so that they never hold back a bucket of ordinary gradients.
        `groups`: parameter collections in the order their gradients become complete when the backward pass is run in
        SEGMENTS (Discriminator.forward(cut_after=...)): buckets never span two groups, so `launch_group(i)` can send a
        finished segment's gradients while the next segment's backward is still running - the form the hipGraph replay
        uses, where post-accumulate hooks do not exist.
        `reduce_op`: "avg" (the collective averages: RCCL) or "sum" (sum, then scale by 1 / world on unpack: gloo); default by
        backend.  `force`: run the whole exchange - buckets, side stream, collective, unpack - at world size 1 as well (a
        one-rank rehearsal of the RCCL path on a single GPU; otherwise a lone rank skips it).
        `solo_bytes` (default 0: off): a gradient of at least this size is a bucket of its own and is all-reduced IN PLACE - no
        copy into a flat buffer and back (at config 2 four generator and five discriminator weights are ~90 % of the 103 MB
        payload).  Measured only in the one-rank RCCL rehearsal, where it LOSES (10.23 -> 10.40 ms per step: nine more
        collectives cost more launch latency than 0.2 ms of copy traffic on the side stream saves); whether it pays over xGMI at
        8 ranks is open, hence opt-in (bench.py: LOCATE_DP_SOLO_BYTES)."""
        self.params = [p for p in params]
        if late is None:
            is_late = [False] * len(self.params)
        elif callable(late):
            is_late = [bool(late(p)) for p in self.params]
        else:
            late_ids = {id(p) for p in late}
            is_late = [id(p) in late_ids for p in self.params]
        self.is_late = is_late
        group_of = [0] * len(self.params)
        if groups is not None:
            index = {id(p): i for i, p in enumerate(self.params)}
            seen = set()
            for gi, members in enumerate(groups):
                for p in members:
                    i = index.get(id(p))
                    if i is not None:
                        group_of[i] = gi
                        seen.add(i)
            last = len(groups)
            for i in range(len(self.params)):
                if i not in seen:
                    group_of[i] = last              # anything not listed: a trailing group of its own
        self.group_of = group_of
        self.n_groups = max(group_of) + 1 if group_of else 1
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.enabled = self.world > 1 or (bool(force) and dist.is_initialized())
        self.overlap = overlap
        self.bucket_bytes = bucket_bytes
        # reverse order ~ gradient production order
        self.buckets = []          # list of lists of parameter indices
        self.bucket_group = []     # segment group of each bucket
        self.bucket_solo = []      # one large gradient, reduced in place
        solo_bytes = int(solo_bytes or 0)
        for gi in range(self.n_groups):
            for group_late in (False, True):
                cur, cur_bytes = [], 0
                for idx in reversed(range(len(self.params))):
                    if is_late[idx] != group_late or group_of[idx] != gi:
                        continue
                    p = self.params[idx]
                    nbytes = p.numel() * p.element_size()
                    if solo_bytes and nbytes >= solo_bytes:
                        # in production order: the packed bucket under construction is closed first
                        if cur:
                            self.buckets.append(cur)
                            self.bucket_group.append(gi)
                            self.bucket_solo.append(False)
                            cur, cur_bytes = [], 0
                        self.buckets.append([idx])
                        self.bucket_group.append(gi)
                        self.bucket_solo.append(True)
                        continue
                    if cur and cur_bytes + nbytes > bucket_bytes:
                        self.buckets.append(cur)
                        self.bucket_group.append(gi)
                        self.bucket_solo.append(False)
                        cur, cur_bytes = [], 0
                    cur.append(idx)
                    cur_bytes += nbytes
                if cur:
                    self.buckets.append(cur)
                    self.bucket_group.append(gi)
                    self.bucket_solo.append(False)
        # RCCL averages in the collective itself; gloo (CPU tests, single-GPU rehearsals) sums and scales afterwards
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        if reduce_op not in (None, "avg", "sum"):
            raise ValueError("reduce_op must be 'avg' or 'sum'")
        self._avg = (backend == "nccl") if reduce_op is None else reduce_op == "avg"
        self._replaying = False    # the running step is a hipGraph replay (begin_replay) - see _check_agreement
        self._replays = 0
        self._copy_tables = {}     # (bucket, gradient / flat addresses) -> device tables of locate_multi_copy
        self.bucket_of = {}
        for b, idxs in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self._flat = [None] * len(self.buckets)
        # resident bucket memory: every parameter's gradient has a HOME inside its bucket's flat buffer (`_locate_grad_buf`, a view);
        # the backward kernels of this package write parameter gradients there (ops._grad_home), so a bucket whose gradients all
        # sit at home is all-reduced where it lies - no packing before, no scattering after the collective
        self.sent_in_place = self.sent_packed = 0          # buckets sent from their resident buffer / through a packed copy (bench.py)
        self.collectives = 0                               # collectives issued for them (a segment group's buckets can share one)
        self._home = [None] * len(self.buckets)
        self._home_off = {}
        self._home_span = {}       # bucket -> (segment group, start, end) inside the group's allocation
        self._ghome = {}           # segment group -> its buckets' resident buffers, back to back
        self._handles = []
        self._ready = None
        self._launched = None
        self._active = False
        self._side = None
        self._hooks = []
        self._hooked = set()
        # optional timing (bench.py): HIP events around every bucket's pack -> all-reduce -> unpack on the side stream, and
        # around the compute stream's join in finish() - the part of the exchange that is NOT hidden behind backward work
        self.timing = False
        # direct: a bucket that is reduced where it lies and needs no scaling afterwards (resident bucket / solo gradient, RCCL's own
        # averaging) is handed to the process group straight from the compute stream - torch runs collectives on the group's own
        # stream anyway, behind an event of the caller's - and the compute stream waits for it in finish().  Through the reducer's
        # side stream the same bucket crossed four stream boundaries (compute -> side -> group -> side -> compute: ~75 us of idle
        # chip per pass in the one-rank RCCL rehearsal, profiles/notes_r04_experiments.md section 4) instead of two.  Every
        # collective still runs on the group's one stream, in issue order.  Buckets that are packed or scaled keep the side stream.
        import os
        self.direct = os.environ.get("LOCATE_DP_DIRECT", "1") != "0"
        self._side_used = False
        self._t_comm = []          # (start, end) event pairs on the side stream
        self._t_wait = []          # (start, end) event pairs on the compute stream

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._hooked = set()

    # ------------------------------------------------------------------------------------------
    def begin(self):
        """Call right before backward()."""
        if not self.enabled:
            return
        self._replaying = False
        # hooks are (re)registered lazily: a tensor can only carry one once it requires grad, and the
        # discriminator's u/v only start to after the first G-step (reference main.py:172)
        for i, p in enumerate(self.params):
            if p.requires_grad and i not in self._hooked and not self.is_late[i]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
                self._hooked.add(i)
        self._ready = [0] * len(self.buckets)
        self._seen = set()
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._active = True
        self._expected = [sum(1 for i in idxs if self.params[i].requires_grad and not self.is_late[i]) or -1
                          for idxs in self.buckets]     # -1: a late-only bucket is never launched from a hook

    def _make_hook(self, i):
        def hook(p):
            # (the hook also fires when autograd reaches a parameter with an UNDEFINED gradient - e.g. the weights whose
            # gradients TrainStep(overlap_wgrad=True) produces on a second stream and assigns at the end of the pass: such a
            # parameter has nothing to send yet; finish() picks it up)
            if not self._active or i in self._seen or p.grad is None:
                return
            self._seen.add(i)
            b = self.bucket_of[i]
            self._ready[b] += 1
            if self.overlap and self._ready[b] == self._expected[b]:
                self._launch(b)
        return hook

    def _launch(self, b):
        members = [i for i in self.buckets[b] if self.params[i].grad is not None]
        self._launched[b] = True
        self._check_agreement(b, members)
        if not members:
            return
        grads = [self.params[i].grad for i in members]
        dev = grads[0].device
        total = sum(g.numel() for g in grads)
        own = self._in_place(b, grads, members)
        in_place = own is not None
        self.sent_in_place += int(in_place)
        self.sent_packed += int(not in_place)
        direct = self.direct and in_place and dev.type == "cuda" and (self._avg or self.world == 1)
        if in_place:
            flat = own                          # the gradient itself / the resident bucket: reduced where it lies
        else:
            flat = self._flat[b]
            if flat is None or flat.numel() != total or flat.device != dev:
                flat = torch.empty(total, dtype=grads[0].dtype, device=dev)
                self._flat[b] = flat
        if direct:
            ev = None
            if self.timing:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(dev))
            work = dist.all_reduce(flat, op=self._op(), group=self.group, async_op=True)
            self.collectives += 1
            self._handles.append((b, members, work, flat, ev, True))
            return
        if dev.type == "cuda":
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            self._side.wait_stream(torch.cuda.current_stream(dev))   # the gradients are produced on the compute stream
            with torch.cuda.stream(self._side):
                if self.timing:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(self._side)
                    self._t_comm.append([ev, None])
                if not in_place:
                    self._pack(b, flat, grads)
                work = dist.all_reduce(flat, op=self._op(), group=self.group, async_op=True)
            self._side_used = True
        else:
            if not in_place:
                self._pack(b, flat, grads)
            work = dist.all_reduce(flat, op=self._op(), group=self.group, async_op=True)
        self.collectives += 1
        self._handles.append((b, members, work, flat if in_place else None, None, False))

    def _in_place(self, b, grads, members=None):
        """The tensor to all-reduce in place, or None: a solo gradient itself; or the bucket's resident buffer when every gradient
        of the bucket lies at its home inside it (slots of parameters without a gradient ride along unread)."""
        if self.bucket_solo[b] and len(grads) == 1 and grads[0].is_contiguous():
            return grads[0].view(-1)
        home = self._home[b]
        if home is None or members is None or not grads or home.device != grads[0].device:
            return None
        base = home.data_ptr()
        for i, g in zip(members, grads):
            if not g.is_contiguous() or g.data_ptr() != base + 4 * self._home_off[i] or g.dtype != torch.float32:
                return None
        return home

    def make_homes(self):
        """Creates the resident bucket buffers and points every parameter at its slice (idempotent; call once the parameters are
        on their device - TrainStep does, before the first backward pass)."""
        if not self.enabled:
            return
        # the buckets of one segment group lie back to back in ONE allocation: under hipGraph replay, where every gradient of a
        # segment is complete when launch_group() / finish() runs, the group goes out as a single collective (_launch_merged)
        todo = {}
        for b, idxs in enumerate(self.buckets):
            if self._home[b] is not None or not idxs or self.bucket_solo[b]:
                continue
            ps = [self.params[i] for i in idxs]
            dev = ps[0].device
            if any(q.device != dev or q.dtype != torch.float32 for q in ps):
                continue
            todo.setdefault((self.bucket_group[b], dev), []).append(b)
        for (gi, dev), bs in todo.items():
            sizes = [sum(self.params[i].numel() for i in self.buckets[b]) for b in bs]
            ghome = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
            start = 0
            for b, n in zip(bs, sizes):
                home = ghome[start:start + n]
                off = 0
                for i in self.buckets[b]:
                    q = self.params[i]
                    self._home_off[i] = off
                    q.__dict__["_locate_grad_buf"] = home[off:off + q.numel()].view(q.shape)
                    off += q.numel()
                self._home[b] = home
                self._home_span[b] = (gi, start, start + n)
                start += n
            self._ghome[gi] = ghome

    def _check_agreement(self, b, members):
        """Which parameters have a gradient is decided per rank (`grad is not None`); ranks that disagreed would exchange
        buckets of different sizes - a hang or silent garbage.  Before a bucket goes out all ranks compare a digest of
        (bucket, members) with one tiny max-reduction of (h, -h): equal everywhere iff both come back unchanged.
        WHETHER the check runs is itself rank-invariant - it depends on the launch mode and a step count only, never on what
        this rank has seen before (a rank that skipped the check while another ran it would pair the check's collective with
        the bucket's): every eager (hook-driven) step checks every bucket; under hipGraph replay, where the member lists are
        frozen into the captured graphs, the first two replays do and the rest do not (no host read in the replayed step)."""
        if self._replaying and self._replays > 2:
            return
        import zlib
        sig = (b, tuple(members))
        h = zlib.crc32(repr(sig).encode()) & 0x7fffffff
        dev = self.params[self.buckets[b][0]].device if self.buckets[b] else torch.device("cpu")
        t = torch.tensor([h, -h], dtype=torch.int64, device=dev)
        # (asynchronous like every bucket, then joined: all of this communicator's collectives go through the process group's one
        # stream in issue order, none runs on the caller's stream beside them)
        work = dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        if work is not None:
            work.wait()
        hi, lo = t.tolist()
        if hi != h or -lo != h:
            raise RuntimeError("data-parallel ranks disagree on which parameters of bucket %d carry a gradient (this rank: %d "
                               "tensors): the replicas' backward passes differ" % (b, len(members)))

    def _op(self):
        return dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM

    # ---- segment-driven form (hipGraph replay: no hooks) --------------------------------------------------------
    def begin_replay(self):
        """Before replaying a captured backward: nothing is launched yet."""
        if not self.enabled:
            return
        self._replaying = True
        self._replays += 1
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._active = False

    def launch_group(self, gi):
        """Every gradient of segment group `gi` is complete on the current stream: send its buckets now, on the side
        stream (the next segment's backward, replayed right after this call, runs beside them)."""
        if not self.enabled:
            return
        first = len(self._handles)
        self._launch_buckets([b for b, g in enumerate(self.bucket_group) if g == gi and not self._launched[b]])
        # their results go back into the .grad tensors as soon as each collective is done - on the side stream, beside the next
        # segment's backward - instead of at finish(), where the compute stream would wait for the copies as well
        self._unpack_handles(first)

    def _launch_buckets(self, bs):
        """Send the given buckets of ONE segment group: as one collective over the group's allocation where every one of them is
        reduced in place from its resident buffer (replayed steps, direct hand-over), else one by one."""
        if len(bs) > 1 and self.direct and self._replaying and all(b in self._home_span for b in bs):
            own = []
            for b in bs:
                members = [i for i in self.buckets[b] if self.params[i].grad is not None]
                grads = [self.params[i].grad for i in members]
                own.append(self._in_place(b, grads, members) if members else None)
            if all(o is not None and o is self._home[b] and o.is_cuda for o, b in zip(own, bs)) and (self._avg or self.world == 1):
                gi = self._home_span[bs[0]][0]
                lo = min(self._home_span[b][1] for b in bs)
                hi = max(self._home_span[b][2] for b in bs)
                flat = self._ghome[gi][lo:hi]
                for b in bs:
                    members = [i for i in self.buckets[b] if self.params[i].grad is not None]
                    self._launched[b] = True
                    self._check_agreement(b, members)
                    self.sent_in_place += 1
                ev = None
                if self.timing:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(torch.cuda.current_stream(flat.device))
                work = dist.all_reduce(flat, op=self._op(), group=self.group, async_op=True)
                self.collectives += 1
                self._handles.append((bs[0], [], work, flat, ev, True))
                return
        for b in bs:
            self._launch(b)

    @staticmethod
    def _views(flat, grads):
        views, off = [], 0
        for g in grads:
            n = g.numel()
            views.append(flat[off:off + n].view_as(g))
            off += n
        return views

    def _copy_table(self, b, flat, grads):
        """Device tables of locate_multi_copy for bucket b: one record per gradient {grad, its place in the bucket, n} and the
        (tensor, chunk) list; cached on every address a record holds (stable under hipGraph replay, where the gradient
        buffers live in the graphs' memory pool)."""
        import struct
        from ._lib import lib
        key = (b, flat.data_ptr()) + tuple(g.data_ptr() for g in grads)
        tab = self._copy_tables.get(key)
        if tab is None:
            L = lib()
            assert L.locate_multi_copy_record_bytes() == 24
            chunk = L.locate_multi_copy_chunk_elems()
            rec, chunks, off = bytearray(), [], 0
            for i, g in enumerate(grads):
                rec += struct.pack("<2Qq", g.data_ptr(), flat.data_ptr() + 4 * off, g.numel())
                chunks.extend((i, c) for c in range((g.numel() + chunk - 1) // chunk))
                off += g.numel()
            t_dev = torch.frombuffer(rec, dtype=torch.uint8).clone().to(flat.device)
            c_dev = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(flat.device)
            tab = (t_dev, c_dev, len(chunks))
            if len(self._copy_tables) > 4 * max(len(self.buckets), 1):
                self._copy_tables.clear()
            self._copy_tables[key] = tab
        return tab

    def _pack(self, b, flat, grads):
        # one table-driven launch per bucket (csrc/parallel.hip) instead of one copy per gradient (~140 tensors per network)
        if flat.is_cuda and all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):
            from ._lib import check, lib
            t_dev, c_dev, n_chunks = self._copy_table(b, flat, grads)
            check(lib().locate_multi_copy(t_dev.data_ptr(), c_dev.data_ptr(), n_chunks, 0, 1.0,
                                          torch.cuda.current_stream(flat.device).cuda_stream), "locate_multi_copy")
            return
        torch._foreach_copy_(self._views(flat, grads), [g if g.is_contiguous() else g.contiguous() for g in grads])

    def finish(self):
        """Call after backward(), before optimizer.step(): afterwards every .grad holds the rank mean."""
        if not self.enabled:
            return
        self._active = False
        for gi in range(self.n_groups):
            self._launch_buckets([b for b, g in enumerate(self.bucket_group) if g == gi and not self._launched[b]])
        self._unpack_handles(0, final=True)
        if self._side is not None and self._side_used:
            self._side_used = False
            cur = torch.cuda.current_stream(self._side.device)
            if self.timing:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
                cur.wait_stream(self._side)
                b.record(cur)
                self._t_wait.append((a, b))
            else:
                cur.wait_stream(self._side)
        self._handles = []

    def _unpack_handles(self, first, final=False):
        """Wait (stream-ordered on the GPU) for the collectives launched since handle `first` and scatter their results.
        Direct buckets (see __init__) are joined by the compute stream itself, and only at finish() (final)."""
        inv = None if self._avg else 1.0 / self.world
        for k in range(first, len(self._handles)):
            if self._handles[k] is None:
                continue
            b, members, work, own, ev, direct = self._handles[k]
            if direct:
                if not final:
                    continue
                self._handles[k] = None
                cur = torch.cuda.current_stream(own.device)
                a = None
                if self.timing:
                    a = torch.cuda.Event(enable_timing=True)
                    a.record(cur)
                work.wait()                      # the compute stream waits for the group's stream
                if inv is not None:
                    own.mul_(inv)
                if a is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(cur)
                    self._t_wait.append((a, e))
                    self._t_comm.append([ev, e])          # (launch ... join on the compute stream: an upper bound of the exchange)
                continue
            self._handles[k] = None
            flat = own if own is not None else self._flat[b]
            dev = flat.device
            if dev.type == "cuda":
                with torch.cuda.stream(self._side):
                    work.wait()
                    if own is None:
                        self._unpack(b, flat, members, inv)
                    elif inv is not None:
                        own.mul_(inv)
                    if self.timing:
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(self._side)
                        for rec in self._t_comm:
                            if rec[1] is None:
                                rec[1] = ev
                                break
            else:
                work.wait()
                if own is None:
                    self._unpack(b, flat, members, inv)
                elif inv is not None:
                    own.mul_(inv)

    def pop_timing(self):
        """(side-stream milliseconds spent on the exchange, milliseconds the compute stream waited for it) since the last
        call; synchronises the device.  exposed / total is the fraction of the exchange NOT hidden behind backward work."""
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._t_comm if b is not None)
        exposed = sum(a.elapsed_time(b) for a, b in self._t_wait)
        self._t_comm, self._t_wait = [], []
        return total, exposed

    def reduce_now(self, replay=False):
        """Non-overlapped form: average every existing .grad across the ranks right now, on the current stream.
        Used between two captured hipGraphs (backward graph -> all-reduce -> optimizer graph), where the hook-driven
        overlap is not available; at config 2 the payload is 47 / 56 MB, ~1 ms of a 25 ms step over xGMI.
        replay: the gradients come from a replayed graph (frozen member lists: see _check_agreement)."""
        if not self.enabled:
            return
        self._replaying = bool(replay)
        if replay:
            self._replays += 1
        inv = None if self._avg else 1.0 / self.world
        for b, idxs in enumerate(self.buckets):
            members = [i for i in idxs if self.params[i].grad is not None]
            if not members:
                continue
            grads = [self.params[i].grad for i in members]
            self._check_agreement(b, members)
            own = self._in_place(b, grads, members)
            if own is not None:
                dist.all_reduce(own, op=self._op(), group=self.group)
                if inv is not None:
                    own.mul_(inv)
                continue
            total = sum(g.numel() for g in grads)
            flat = self._flat[b]
            if flat is None or flat.numel() != total or flat.device != grads[0].device:
                flat = torch.empty(total, dtype=grads[0].dtype, device=grads[0].device)
                self._flat[b] = flat
            self._pack(b, flat, grads)
            dist.all_reduce(flat, op=self._op(), group=self.group)
            self._unpack(b, flat, members, inv)

    def _unpack(self, b, flat, members, inv):
        grads = [self.params[i].grad for i in members]
        if flat.is_cuda and all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):
            from ._lib import check, lib
            t_dev, c_dev, n_chunks = self._copy_table(b, flat, grads)
            check(lib().locate_multi_copy(t_dev.data_ptr(), c_dev.data_ptr(), n_chunks, 1, 1.0 if inv is None else float(inv),
                                          torch.cuda.current_stream(flat.device).cuda_stream), "locate_multi_copy")
            return
        if inv is not None:
            flat.mul_(inv)
        if all(g.is_contiguous() for g in grads):
            torch._foreach_copy_(grads, self._views(flat, grads))
        else:
            for g, v in zip(grads, self._views(flat, grads)):
                g.copy_(v)
