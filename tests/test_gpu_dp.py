"""Data parallelism on the real networks and the HIP path: two ranks, each a fresh child process on
cuda:0 (gloo backend: RCCL does not allow two ranks on one device), the tiny g8 generator / discriminator, TrainStep with
GradAllReducer - bucketed exchange on a side stream, the D-step's backward in two segments with the deep segment sent while
the other runs.

Parity statement under test: the gradient every rank's optimizer sees == the MEAN of the gradients the reference (here: the
CPU oracle, pinned to the reference by tests/golden) computes on each rank's shard independently - InPlaceNorm statistics
are per replica by design.  And: hipGraph replay with the exchange between the graphs == the eager hook-driven form, and the
replicas stay bit-identical."""
import copy
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import GOLDEN_DIR, ROOT, assert_close, load_golden

pytestmark = pytest.mark.gpu
T = torch.as_tensor


def _run_ranks(tmp_path, mode, world=2, backend=None):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    golden = os.path.join(GOLDEN_DIR, "g8_tiny_e2e.npz")
    worker = os.path.join(ROOT, "tests", "helpers", "dp_worker.py")
    outs = [str(tmp_path / ("%s_rank%d.pt" % (mode, r))) for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend:
        env["LOCATE_TEST_BACKEND"] = backend
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, golden, outs[r], mode], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [torch.load(o, weights_only=True) for o in outs]


def _oracle_shard_means(z, world=2):
    """Per-shard reference gradients with the oracle, averaged: D-step on every shard from the common initial state, ONE
    Nadam step with the averaged D gradients (what every rank's optimizer does), then the G-step on every shard."""
    from oracle import locate_oracle as O
    cfg = O.NetConfig(image_size=32, base_feature_factor=1)
    sdG = {k[len("G/sd0/"):]: T(z[k]) for k in z.files if k.startswith("G/sd0/")}
    sdD = {k[len("D/sd0/"):]: T(z[k]) for k in z.files if k.startswith("D/sd0/")}
    noise = T(z["G/noise"])
    B = z["step1/latent"].shape[0]
    shards = [[T(z["step1/" + k])[r * B // world:(r + 1) * B // world] for k in ("latent", "real", "aug")] for r in range(world)]

    def mean(dicts):
        return {k: sum(d[k] for d in dicts) / len(dicts) for k in dicts[0]}
    d_grads, losses, states = [], [], []
    for latent, real, aug in shards:
        PG, PD = O.make_params(copy.deepcopy(sdG)), O.make_params(copy.deepcopy(sdD))
        with torch.enable_grad():
            generated = O.generator_forward(PG, noise, latent, cfg).detach()
            d_true = O.discriminator_forward(PD, real, cfg).view(-1)
            d_gen = -O.discriminator_forward(PD, generated, cfg).view(-1)
            d_error = (O.hinge(d_true) + O.hinge(d_gen)).mean()
            pen = O.consistency_penalty(d_true, O.discriminator_forward(PD, aug, cfg))
            (d_error + pen).backward()
        d_grads.append({k: g.clone() for k, g in O._grads_of(PD).items()})
        losses.append(float(d_error.detach()))
        states.append((PG, PD))
    d_mean = mean(d_grads)
    g_grads = []
    for (latent, real, aug), (PG, PD) in zip(shards, states):
        O.Nadam(cfg.dlr, (cfg.beta1, cfg.beta2)).step(PD, d_mean)          # every replica applies the SAME averaged gradient
        for p in PD.values():
            p.requires_grad_(False)
        O._zero_grad(PG)
        with torch.enable_grad():
            fake = O.generator_forward(PG, noise, latent, cfg)
            O.hinge(O.discriminator_forward(PD, fake, cfg).view(-1)).mean().backward()
        g_grads.append({k: g.clone() for k, g in O._grads_of(PG).items()})
    return d_mean, mean(g_grads), sum(losses) / world


@pytest.mark.parametrize("suffix", ["", "+wg"])
def test_dp_gradients_equal_mean_of_reference_shard_gradients(tmp_path, suffix):
    z = load_golden("g8_tiny_e2e")
    recs = _run_ranks(tmp_path, "eager" + suffix)
    d_want, g_want, d_loss = _oracle_shard_means(z)
    nb_d, nb_g, groups = recs[0]["buckets"]
    assert nb_d >= 3 and nb_g >= 2 and groups == 2             # several buckets per net, two segment groups for D
    for rank, rec in enumerate(recs):
        assert set(rec["d_grads"]) == set(d_want) and set(rec["g_grads"]) == set(g_want)
        for k, v in d_want.items():
            assert_close(rec["d_grads"][k], v, 3e-4, "rank %d D grad %s" % (rank, k))
        for k, v in g_want.items():
            assert_close(rec["g_grads"][k], v, 3e-4, "rank %d G grad %s" % (rank, k))
    # identical averaged gradients => identical replicas after the optimizer steps, bit for bit
    for net in ("G", "D"):
        for k, v in recs[0][net].items():
            assert torch.equal(v, recs[1][net][k]), (net, k)
    mean_loss = sum(float(r["d_error"]) for r in recs) / len(recs)
    assert abs(mean_loss - d_loss) <= 2e-5 * max(abs(d_loss), 1.0)


@pytest.mark.parametrize("suffix", ["", "+wg"])
def test_dp_graph_replay_with_segmented_backward_equals_eager(tmp_path, suffix):
    eager = _run_ranks(tmp_path, "eager4")
    graph = _run_ranks(tmp_path, "graph" + suffix)
    for net in ("G", "D"):
        for k, v in graph[0][net].items():
            assert torch.equal(v, graph[1][net][k]), ("replicas diverged under graph replay", net, k)
            assert_close(v, eager[0][net][k], 1e-5, "graph vs eager " + net + " " + k)


def test_one_rank_rccl_replay_sends_one_collective_per_segment_and_changes_nothing(tmp_path):
    """The RCCL path itself, as far as one GPU can run it (world size 1, reducers forced on): the replayed step hands every
    backward segment's resident buckets to the process group as ONE collective, straight from the compute stream (parallel.py:
    direct / _launch_buckets).  At world size 1 the average is the identity, so the run must end where a run without any
    reducer ends - a bucket that missed its exchange, was scaled twice or raced with the optimizer would show."""
    plain = _run_ranks(tmp_path, "eager4", world=1)[0]
    assert not plain["stats"]["enabled"]
    rccl = _run_ranks(tmp_path, "graph", world=1, backend="nccl")[0]
    st = rccl["stats"]
    assert st["enabled"]
    sent = sum(st["in_place"]) + sum(st["packed"])
    assert st["buckets"][0] >= 3 and st["buckets"][1] >= 2
    assert sum(st["collectives"]) < sent, ("no segment was merged into one collective", st)
    assert sum(st["in_place"]) > sum(st["packed"]), st
    for net in ("G", "D"):
        for k, v in rccl[net].items():
            assert_close(v, plain[net][k], 1e-5, "one-rank RCCL replay vs no reducer: " + net + " " + k)
