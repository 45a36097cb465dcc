"""Launches ONLY the dominant kernel's four workloads (forward of the generator's ConvTranspose 4x4 s2 stages at config 2:
C = 768, 384, 192, 96, batch 64), `reps` times each, in that order - the command the rocprofv3 --pmc passes behind
bench.py's `roofline.traffic` are taken over (see tools/pmc_summary.py --json and profiles/README).
usage: python3 tools/roofline_stages.py [--reps 5] [--bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import ops  # noqa: E402

STAGES = [(768, 4), (384, 8), (192, 16), (96, 32)]       # (C, input side) of G.b1 .. G.b4 at 64x64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--bf16-pieces", action="store_true", help="the three-piece bf16 form (six MFMAs per slice) instead of the fp16-piece form")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rt = ops.Runtime()
    rt.precision = 1 if args.bf16 else 0
    spec = ops.ConvSpec("convT", 4, 4, 2, 1, 1)
    for c, size in STAGES:
        w = torch.randn(c, c, 4, 4, device=dev) * 0.05
        u, v = torch.randn(c, device=dev), torch.randn(c * 16, device=dev)
        x = torch.randn(args.batch, c, size, size, device=dev)
        if not (args.bf16 or args.bf16_pieces):
            ops.tag_amax(x)          # the producing kernel's job in the step: selects the two-piece fp16 form (precision 2)
        pre = ops.sn_power_iteration(w, u, v)
        with torch.no_grad():
            for _ in range(args.reps + 1):           # the first call also packs the weight panel
                ops.sn_conv(x, w, u, v, None, spec, pre, rt)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
