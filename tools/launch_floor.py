"""Per-kernel cost of a dependent chain of tiny kernels replayed from a hipGraph (the floor under every small launch)."""
import time
import torch
dev = torch.device("cuda:0")
for n_elem in (64, 1 << 16, 1 << 20):
    x = torch.zeros(n_elem, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(1000):
                x.add_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print("graph chain, %8d elements: %.2f us per kernel" % (n_elem, (time.perf_counter() - t0) / 10 / 1000 * 1e6))
    t0 = time.perf_counter()
    for _ in range(5000):
        x.add_(1.0)
    torch.cuda.synchronize()
    print("eager stream, %8d elements: %.2f us per kernel" % (n_elem, (time.perf_counter() - t0) / 5000 * 1e6))
