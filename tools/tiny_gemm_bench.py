"""Per-launch time of tiny spectral-normalised 1x1 contractions (the style linears / deep discriminator layers) replayed
from a hipGraph chain: separates the fixed per-launch cost from the per-K-step cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from locate_amd import ops

dev = torch.device("cuda:0")
B = 64
print("   K     M    us/launch (graph chain of 200)")
for K, M in ((16, 64), (64, 64), (256, 64), (832, 384), (3072, 768), (832, 64), (12800, 512)):
    w = torch.randn(M, K, 1, 1, device=dev) * 0.05
    u = torch.randn(M, device=dev)
    v = torch.randn(K, device=dev)
    x = torch.randn(B, K, 1, 1, device=dev)
    spec = ops.ConvSpec("conv", 1, 1, 1, 0, 0)
    pre = ops.sn_power_iteration(w, u, v)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            ops.sn_conv(x, w, u, v, None, spec, pre)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                y = ops.sn_conv(x, w, u, v, None, spec, pre)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    e1.synchronize()
    print("%5d %5d   %7.2f" % (K, M, e0.elapsed_time(e1) / 10 / 200 * 1e3))
