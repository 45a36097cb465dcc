"""Un-profiled time of each captured graph of one iteration (HIP events between the replays; the graphs run one after the other
here, so the figures are each graph's own length, not the overlapped schedule's).  usage: python tools/graph_times.py [--no-wgrad-overlap]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model  # noqa: E402
from locate_amd.graph import GraphedTrainStep  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = NetConfig(image_size=64)
    torch.manual_seed(cfg.seed)
    G, GO = get_model(Generator(cfg), cfg.glr, dev)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    step = TrainStep(G, D, GO, DO, overlap_wgrad="--no-wgrad-overlap" not in sys.argv)
    B, S = 64, 64
    args = (torch.randn(B, S, device=dev), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1))
    runner = GraphedTrainStep(step, *args, warmup=2)
    names = ["G fwd (D-step)", "D sn x3", "D fwd+bwd [3B]", "D nadam+pack", "G fwd (G-step)", "D fwd + bwd into G", "G nadam+pack"]
    graphs = [runner.graphs[0], runner.sn_graph, runner.graphs[1], runner.graphs[2], runner.graphs[3], runner.graphs[4], runner.graphs[5]]
    for _ in range(3):
        runner.replay()
    torch.cuda.synchronize()
    reps = 10
    tot = [0.0] * len(graphs)
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(graphs) + 1)]
        ev[0].record()
        for i, g in enumerate(graphs):
            g.replay()
            ev[i + 1].record()
        torch.cuda.synchronize()
        for i in range(len(graphs)):
            tot[i] += ev[i].elapsed_time(ev[i + 1])
    for n, t in zip(names, tot):
        print("%-22s %8.3f ms" % (n, t / reps))
    print("%-22s %8.3f ms" % ("sum", sum(tot) / reps))
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        runner.replay()
    torch.cuda.synchronize()
    print("overlapped schedule     %8.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))


if __name__ == "__main__":
    main()
