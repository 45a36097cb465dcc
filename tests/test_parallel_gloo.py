"""Data-parallel gradient averaging (locate_amd/parallel.py) with world_size 2 on CPU over gloo.
Parity statement under test: after finish(), every rank's .grad equals the MEAN of the gradients each rank
computed on its own shard; parameters without a gradient stay None; replicas start identical after broadcast."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, bucket_bytes, overlap, q, solo_bytes=0):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from locate_amd.parallel import GradAllReducer, broadcast_module_state
        torch.manual_seed(100 + rank)            # different initial weights per rank on purpose
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        unused = torch.nn.Parameter(torch.randn(4))          # like G's unused i_norm.weight: grad stays None
        frozen = torch.nn.Parameter(torch.randn(2), requires_grad=False)   # like D's u/v before main.py:172
        extra = torch.randn(1, 3, 2, 2)                      # like Generator.noise
        broadcast_module_state(net, 0, extra_tensors=[extra, unused.data, frozen.data])
        late = torch.nn.Parameter(torch.randn(3))              # like D's spectral-norm v: .grad assigned after backward
        broadcast_module_state(torch.nn.ParameterList([late]), 0)
        params = list(net.parameters()) + [unused, frozen, late]
        red = GradAllReducer(params, bucket_bytes=bucket_bytes, overlap=overlap, late=[late], solo_bytes=solo_bytes)
        torch.manual_seed(7)                                  # same data stream on both ranks, shard by rank
        x = torch.randn(8, 6)
        shard = x[rank * 4:(rank + 1) * 4]
        results = []
        for it in range(2):
            for p in params:
                p.grad = None
            loss = net(shard).pow(2).mean() * (it + 1)
            red.begin()
            loss.backward()
            late.grad = torch.full((3,), float(rank + 1 + it))   # assigned outside autograd, before finish()
            red.finish()
            results.append([None if p.grad is None else p.grad.clone() for p in params])
        # single-process truth: mean over the two shards' gradients
        truth = []
        for it in range(2):
            acc = None
            for r in range(world):
                for p in params:
                    p.grad = None
                (net(x[r * 4:(r + 1) * 4]).pow(2).mean() * (it + 1)).backward()
                late.grad = torch.full((3,), float(r + 1 + it))
                g = [None if p.grad is None else p.grad.clone() for p in params]
                acc = g if acc is None else [None if a is None else a + b for a, b in zip(acc, g)]
            truth.append([None if a is None else a / world for a in acc])
        ok = True
        for got, want in zip(results, truth):
            for a, b in zip(got, want):
                if (a is None) != (b is None):
                    ok = False
                elif a is not None and not torch.allclose(a, b, rtol=1e-6, atol=1e-7):
                    ok = False
        # non-overlapped form used between captured graphs
        for p in params:
            p.grad = None
        (net(shard).pow(2).mean()).backward()
        late.grad = torch.full((3,), float(rank + 1))
        red.reduce_now()
        now = [None if p.grad is None else p.grad.clone() for p in params]
        for a, b in zip(now, truth[0]):
            if (a is None) != (b is None) or (a is not None and not torch.allclose(a, b, rtol=1e-6, atol=1e-7)):
                ok = False
        w0 = [p.detach().clone() for p in net.parameters()] + [extra]
        gathered = [None] * world
        dist.all_gather_object(gathered, [t.tolist() for t in w0])
        same_weights = gathered[0] == gathered[1]
        q.put((rank, ok, same_weights, len(red.buckets), sum(red.bucket_solo)))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, False, traceback.format_exc()))


@pytest.mark.parametrize("bucket_bytes,overlap,solo_bytes", [(32 << 20, True, 0), (64, True, 0), (64, False, 0),
                                                             (32 << 20, True, 60), (64, False, 60)])
def test_grad_allreduce_world2(bucket_bytes, overlap, solo_bytes):
    """solo_bytes = 60: the two weight matrices (120 and 60 bytes) are buckets of their own, all-reduced in place (sum, then scaled:
    gloo), between packed buckets of the biases - hook-driven, flushed by finish(), and in reduce_now()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, overlap, q, solo_bytes)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] and r[2], r
    if bucket_bytes == 64:
        assert res[0][3] > 1        # several buckets were exercised
    assert res[0][4] == (2 if solo_bytes == 60 else 0)        # in-place buckets


def test_single_process_is_a_no_op():
    from locate_amd.parallel import GradAllReducer
    p = torch.nn.Parameter(torch.randn(3))
    red = GradAllReducer([p])
    red.begin()
    (p * 2).sum().backward()
    red.finish()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def _worker_modes(rank, world, port, mode, q):
    """mode "avg": the averaging-collective bookkeeping (no 1/world scaling on unpack: what RCCL's ReduceOp.AVG takes) pinned
    over gloo, whose collectives cannot average - the reduction is a sum of inputs pre-scaled by 1/world, so the mean must
    come out of the `inv is None` path.  mode "disagree": rank 1 lacks one gradient - both ranks must raise, not hang.
    mode "force1": see test_forced_single_rank_runs_the_exchange."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from locate_amd.parallel import GradAllReducer
        torch.manual_seed(3)
        params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7, 3)]
        if mode == "avg":
            red = GradAllReducer(params, bucket_bytes=32, reduce_op="avg")
            red._op = lambda: dist.ReduceOp.SUM            # gloo has no AVG: sum of pre-scaled inputs instead
            assert red._avg
            for i, p in enumerate(params):
                p.grad = torch.full_like(p, float(rank + 1 + i)) / world        # pre-scaled
            red.begin()
            red.finish()
            ok = all(torch.allclose(p.grad, torch.full_like(p, (1 + 2) / 2 + i)) for i, p in enumerate(params))
            red.reduce_now()                                                   # the same values averaged again: sum of x / ... stays put
            q.put((rank, ok, ""))
        elif mode == "homes":
            # gradients produced INSIDE the buckets' resident buffers (make_homes: what ops._grad_home does on the GPU): such a bucket
            # is all-reduced where it lies - no flat copy is created; a parameter without a gradient (index 1 in the second round)
            # leaves its slot unread and its .grad None; a gradient that is NOT at home sends its bucket through the packed path
            red = GradAllReducer(params, bucket_bytes=40)              # buckets in reverse parameter order: [2, 1] (3 + 7 floats) | [0]
            assert red.buckets == [[2, 1], [0]]
            red.make_homes()
            assert all("_locate_grad_buf" in q_.__dict__ for q_ in params)
            ok = True
            for rnd in range(3):
                for i, p in enumerate(params):
                    if rnd == 1 and i == 1:
                        p.grad = None
                        continue
                    home = p.__dict__["_locate_grad_buf"]
                    home.copy_(torch.full_like(p, float(rank + 1 + i + rnd)))
                    p.grad = home if not (rnd == 2 and i == 0) else home.clone()      # round 2: parameter 0's gradient is elsewhere
                red.begin()
                red.finish()
                for i, p in enumerate(params):
                    if rnd == 1 and i == 1:
                        ok = ok and p.grad is None
                    else:
                        ok = ok and torch.allclose(p.grad, torch.full_like(p, (1 + 2) / 2 + i + rnd))
                at_home = [p.grad is not None and p.grad.data_ptr() == p.__dict__["_locate_grad_buf"].data_ptr() for p in params]
                ok = ok and at_home == ([True, True, True] if rnd == 0 else ([True, False, True] if rnd == 1 else [False, True, True]))
                ok = ok and (red._flat[0] is None) and ((red._flat[1] is None) == (rnd < 2))   # a packed copy only for round 2's second bucket
            q.put((rank, ok, ""))
        elif mode == "disagree":
            red = GradAllReducer(params, bucket_bytes=1 << 20)
            for i, p in enumerate(params):
                p.grad = None if (rank == 1 and i == 1) else torch.ones_like(p)
            red.begin()
            try:
                red.finish()
                q.put((rank, False, "no error raised"))
            except RuntimeError as e:
                q.put((rank, "disagree" in str(e), str(e)))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("mode", ["avg", "homes", "disagree"])
def test_reducer_modes_world2(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_modes, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, (rank, info)


def test_forced_single_rank_runs_the_exchange():
    """force=True: a lone rank still packs, all-reduces (a self-copy) and unpacks - the one-rank rehearsal of the RCCL path
    (`LOCATE_DP_FORCE=1 python -m torch.distributed.run --nproc-per-node 1 bench.py`)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from locate_amd.parallel import GradAllReducer
        params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7)]
        red = GradAllReducer(params, bucket_bytes=16, force=True)
        assert red.enabled and len(red.buckets) == 2
        red.begin()
        sum((p * (i + 2)).sum() for i, p in enumerate(params)).backward()
        assert all(red._launched)                      # both buckets went out from the post-accumulate hooks
        red.finish()
        assert all(torch.equal(p.grad, torch.full_like(p, float(i + 2))) for i, p in enumerate(params))
        lone = GradAllReducer(params)
        assert not lone.enabled
    finally:
        dist.destroy_process_group()
