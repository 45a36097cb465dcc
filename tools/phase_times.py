"""Time the four captured graphs of one iteration separately (D fwd/bwd, D Nadam, G fwd/bwd, G Nadam)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model
from locate_amd.graph import GraphedTrainStep
dev = torch.device("cuda:0")
cfg = NetConfig(image_size=64)
torch.manual_seed(999)
G, GO = get_model(Generator(cfg), cfg.glr, dev)
D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
G.batched_spectral_norm = D.batched_spectral_norm = True
step = TrainStep(G, D, GO, DO)
B, S = 64, 64
lat = torch.randn(B, S, device=dev); real = torch.randn(B, 3, S, S, device=dev).clamp(-1, 1); aug = torch.randn(B, 3, S, S, device=dev).clamp(-1, 1)
r = GraphedTrainStep(step, lat, real, aug, overlap=False)      # the four-graph (in-line) schedule, one phase per graph
for _ in range(3): r.replay()
torch.cuda.synchronize()
names = ["D fwd x3 + G fwd + bwd", "D nadam", "G fwd + D fwd + bwd", "G nadam"]
tot = [0.0] * 4
N = 10
for _ in range(N):
    for i, g in enumerate(r.graphs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize(); tot[i] += time.perf_counter() - t0
for n, t in zip(names, tot): print("%-28s %7.3f ms" % (n, t / N * 1e3))
# finer: eager pieces
def tm(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("G forward (eager, no_grad)   %7.3f ms" % tm(lambda: G(lat)))
    x = G(lat)
    print("D forward (eager, no_grad)   %7.3f ms" % tm(lambda: D(x)))
