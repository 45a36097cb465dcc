"""One rank of the data-parallel GPU test (tests/test_gpu_dp.py): NOT a test module.  Started as a fresh child process per
rank (both ranks on cuda:0, gloo backend - RCCL refuses two ranks on one device; the control flow, the bucketing, the
segmented backward and the side-stream exchange are the same).  argv: rank world port golden out_path mode
mode = "eager": one step of the tiny g8 network on this rank's shard through TrainStep + GradAllReducer; records the
averaged gradients the optimizers saw.  mode = "graph": two eager + `replays` graph-replayed steps; mode = "eager4": the same
number of steps eagerly.  Both record the final parameters.  A "+wg" suffix: the weight gradients on a second stream
(TrainStep(overlap_wgrad=True)) - they reach `.grad` through an end-of-backward callback, not through autograd's accumulation."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, golden, out_path, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    overlap_wgrad = mode.endswith("+wg")
    mode = mode.split("+")[0]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    # LOCATE_TEST_BACKEND=nccl (world 1 only): the RCCL path itself - reducers forced on at world size 1, so that the direct
    # hand-over to the process group's stream and the one-collective-per-segment form of the replayed step are exercised
    backend = os.environ.get("LOCATE_TEST_BACKEND", "gloo")
    force = backend == "nccl"
    if force:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from locate_amd import Discriminator, Generator, Nadam, NetConfig, TrainStep
    from locate_amd.graph import GraphedTrainStep
    from locate_amd.parallel import GradAllReducer, broadcast_module_state
    z = np.load(golden, allow_pickle=False)
    T = torch.as_tensor
    dev = torch.device("cuda:0")
    cfg = NetConfig(image_size=32, base_feature_factor=1)
    torch.manual_seed(1000 + rank)                       # different initial weights per rank ON PURPOSE: the broadcast fixes it
    G, D = Generator(cfg), Discriminator(cfg)
    if rank == 0:
        G.load_state_dict({k[len("G/sd0/"):]: T(z[k]) for k in z.files if k.startswith("G/sd0/")})
        D.load_state_dict({k[len("D/sd0/"):]: T(z[k]) for k in z.files if k.startswith("D/sd0/")})
        G.noise = T(z["G/noise"]).clone()
    G, D = G.to(dev), D.to(dev)
    broadcast_module_state(G, 0, extra_tensors=[G.noise])
    broadcast_module_state(D, 0)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    d_cut = 2
    late_v = [p for n, p in D.named_parameters() if n.endswith("weight_v")]
    red_g = GradAllReducer(G.parameters(), bucket_bytes=64 << 10, force=force)          # small buckets: several per network
    red_d = GradAllReducer(D.parameters(), bucket_bytes=64 << 10, late=late_v, groups=D.segment_parameters(d_cut), force=force)
    step = TrainStep(G, D, Nadam(G.parameters(), lr=cfg.glr, betas=(cfg.beta1, cfg.beta2)),
                     Nadam(D.parameters(), lr=cfg.dlr, betas=(cfg.beta1, cfg.beta2)), reducer_g=red_g, reducer_d=red_d, d_cut=d_cut,
                     overlap_wgrad=overlap_wgrad)
    B = z["step1/latent"].shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = [T(z["step1/" + k])[lo:hi].to(dev) for k in ("latent", "real", "aug")]
    rec = {}
    if mode == "eager":
        d_orig, g_orig = step.dis_opt.step, step.gen_opt.step

        def d_hook(*a, **k):
            rec["d_grads"] = {n: p.grad.detach().cpu().clone() for n, p in D.named_parameters() if p.grad is not None}
            return d_orig(*a, **k)

        def g_hook(*a, **k):
            rec["g_grads"] = {n: p.grad.detach().cpu().clone() for n, p in G.named_parameters() if p.grad is not None}
            return g_orig(*a, **k)
        step.dis_opt.step, step.gen_opt.step = d_hook, g_hook
        out = step(*shard)
        rec["d_error"], rec["g_error"] = out["d_error"].detach().cpu(), out["g_error"].detach().cpu()
        rec["buckets"] = (len(red_d.buckets), len(red_g.buckets), red_d.n_groups)
    else:
        replays = 2
        if mode == "graph":
            runner = GraphedTrainStep(step, *shard, warmup=2)            # two eager steps inside
            assert runner.d_tail is not None                             # the segmented D backward was captured
            for _ in range(replays):
                runner.replay()
        else:
            for _ in range(2 + replays):
                step(*shard)
    torch.cuda.synchronize()
    rec["stats"] = {"enabled": bool(red_d.enabled), "collectives": (red_d.collectives, red_g.collectives),
                    "in_place": (red_d.sent_in_place, red_g.sent_in_place), "packed": (red_d.sent_packed, red_g.sent_packed),
                    "buckets": (len(red_d.buckets), len(red_g.buckets))}
    rec["G"] = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    rec["D"] = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
    torch.save(rec, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
