"""Cost of the largest-magnitude output of the producing kernels (norm + RootTanh forward, RootTanh backward): the same launch
with and without the absmax words, HIP events over a captured graph of 20 launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd._lib import check, lib  # noqa: E402
from tools.bench_conv import time_it  # noqa: E402

dev = torch.device("cuda:0")
L = lib()
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr() if t is not None else None
for shape in ((64, 768, 4, 4), (64, 384, 8, 8), (64, 192, 16, 16), (64, 96, 32, 32), (64, 48, 64, 64)):
    Bn, C, H, W = shape
    n, hw = Bn * C * H * W, H * W
    x, g, y, gx = (torch.randn(shape, device=dev) for _ in range(4))
    w_, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.empty(2, device=dev)
    ws = torch.empty(max(L.locate_norm_stats_workspace_bytes(), 16), dtype=torch.uint8, device=dev)
    slot = torch.zeros(L.locate_absmax_words(), dtype=torch.int32, device=dev)
    row = []
    for am in (None, slot):
        row.append(time_it(lambda: check(L.locate_norm_fwd(P(x), P(w_), 0, P(b_), P(y), 1, P(stats), Bn, C, hw, 1, P(ws), None, P(am), S())), 20) * 1e3)
        row.append(time_it(lambda: check(L.locate_roottanh_bwd(P(x), P(g), P(gx), n, 0, P(am), S())), 20) * 1e3)
    print("%-18s norm+act fwd %6.1f -> %6.1f us   RootTanh bwd %6.1f -> %6.1f us" % ("x".join(map(str, shape)), row[0], row[2], row[1], row[3]))
