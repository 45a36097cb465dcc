"""Where do the step's device-to-device copies and element-wise torch kernels come from?  One eager step under torch.profiler,
aten::copy_ / clone / add / fill_ grouped by Python call site.  Usage (GPU box): python tools/copy_sources.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = NetConfig(image_size=64)
    torch.manual_seed(0)
    G, GO = get_model(Generator(cfg), cfg.glr, dev)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    step = TrainStep(G, D, GO, DO, overlap_wgrad=True)
    B, S = 64, 64
    args = (torch.randn(B, S, device=dev), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1))
    for _ in range(3):
        step(*args)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        step(*args)
        torch.cuda.synchronize()
    sites = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::mul", "aten::contiguous"):
            stack = [s for s in (ev.stack or []) if "locate_amd" in s or "autograd" in s]
            where = stack[0] if stack else "(engine / no python frame)"
            sites[(ev.name, where.strip()[:110], str(ev.input_shapes)[:60])] += 1
    for (name, where, shapes), n in sites.most_common(60):
        print("%4d  %-16s %-110s %s" % (n, name, where, shapes))


if __name__ == "__main__":
    main()
