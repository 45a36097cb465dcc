"""Split reductions of the weight gradients of one config-2 step: per geometry the number of slabs, their size and the slab bytes the
batched reduction reads at the end of the pass.  usage (GPU box): python tools/slab_census.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import NetConfig, ops  # noqa: E402
from locate_amd._lib import lib  # noqa: E402
from tools.layer_table import collect  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = NetConfig(image_size=64)
    calls = collect(cfg, 64, dev)
    L = lib()
    rows, total = [], 0
    for (kind, kh, kw, s, ph, pw, xs, ws_), n_calls in calls.items():
        spec = ops.ConvSpec(kind, kh, kw, s, ph, pw)
        geom, _ = spec.geometry(tuple(xs), tuple(ws_))
        garr = ops._geom(geom)
        # forward calls that also run a weight gradient: one per backward pass (the D-step's stacked pass has batch 192)
        wsb = L.locate_conv_wgrad_workspace_bytes(garr)
        n = ws_[0] * ws_[1] * kh * kw
        ns = wsb // (4 * n) if n else 0
        if ns > 1:
            rows.append((wsb, kind, tuple(geom), n, ns))
    rows.sort(reverse=True)
    for ws, kind, geom, n, ns in rows:
        total += ws
        print("%-5s %-52s n %9d  slabs %4d  %7.1f MB" % (kind, str(tuple(geom)), n, ns, ws / 1e6))
    print("total slab bytes read per step (weight gradients with a split): %.1f MB" % (total / 1e6))


if __name__ == "__main__":
    main()
