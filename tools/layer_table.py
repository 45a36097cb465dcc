"""Every dense contraction of one G+D step at config 2 (64x64 RGB, batch 64), each distinct geometry timed in isolation
(forward / input gradient / weight gradient through the C ABI, HIP events) and set against two floors: the matrix pipe
(three fp16 MFMAs per multiply-add, 2500 / 3 TFLOP/s nominal, for layers of >= ops.F16_MIN_FLOPS; six bf16 MFMAs below) and HBM
(activations once each + the weight planes, 8 TB/s).
The table is sorted by the time a geometry costs per step; `x floor` says how far each launch is from the larger floor -
the launches that are far from both are latency- / occupancy-bound and are what the tile and split heuristics can still move.
Usage (GPU box): python tools/layer_table.py [--reps 20] [--top 40]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, ops  # noqa: E402
import tools.bench_conv as bench_conv  # noqa: E402
from tools.bench_conv import bench_shape  # noqa: E402

bench_conv.WIN = "auto"          # the kernels the step itself takes: the window form where the library recommends it


def collect(cfg, batch, dev):
    """(geometry -> [forward calls per step]) by running one eager step with SNConvFn.forward observed."""
    torch.manual_seed(0)
    G, GO = get_model(Generator(cfg), cfg.glr, dev)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    step = TrainStep(G, D, GO, DO)
    S = cfg.image_size
    args = (torch.randn(batch, cfg.input_vector_z, device=dev), torch.randn(batch, 3, S, S, device=dev).clamp(-1, 1),
            torch.randn(batch, 3, S, S, device=dev).clamp(-1, 1))
    step(*args)                      # warm-up (allocations, u/v becoming trainable)
    calls = collections.Counter()
    orig = ops._conv_apply

    def spy(x, w, owner, spec, *rest, **kw):
        if spec.mode == "dense":
            calls[(spec.kind, spec.kh, spec.kw, spec.stride, spec.pad_h, spec.pad_w, tuple(x.shape), tuple(w.shape))] += 1
        return orig(x, w, owner, spec, *rest, **kw)
    ops._conv_apply = spy
    try:
        step(*args)
    finally:
        ops._conv_apply = orig
    torch.cuda.synchronize()
    return calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = NetConfig(image_size=args.image_size)
    calls = collect(cfg, args.batch, dev)
    rows = []
    for (kind, kh, kw, s, ph, pw, xs, ws), n in calls.items():
        B, cin, H, W = xs
        cout = ws[0] if kind == "conv" else ws[1]
        # the form the step runs the layer in: two fp16 pieces (three MFMAs per multiply-add) from ops.F16_MIN_FLOPS of work up -
        # where the operands' largest magnitudes are known, which this table assumes -, three bf16 pieces (six) below
        oh = (H + 2 * ph - kh) // s + 1 if kind == "conv" else H
        ow = (W + 2 * pw - kw) // s + 1 if kind == "conv" else W
        work = 2.0 * B * oh * ow * cin * cout * kh * kw
        f16 = work >= ops.F16_MIN_FLOPS
        ms, flops, nbytes = bench_shape(kind, cin, cout, kh, kw, s, ph, pw, B, H, W, args.reps, 2 if f16 else 0)
        floor = max(flops / (2500e12 / (3 if f16 else 6)), nbytes / 8e12) * 1e3          # ms
        rows.append((n * sum(ms), n, kind, cin, cout, kh, kw, s, B, H, W, ms, flops, floor))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print("# %d distinct geometries, %d forward calls per step; forward + input gradient + weight gradient of every call: %.3f ms"
          % (len(rows), sum(r[1] for r in rows), total))
    print("%-34s %5s %8s | %-22s | %-22s | %-22s | %7s" % ("geometry (kind Cin->Cout k/s B HxW)", "calls", "GFLOP", "fwd us (x floor)",
                                                            "dgrad us (x floor)", "wgrad us (x floor)", "ms/step"))
    for tot, n, kind, cin, cout, kh, kw, s, B, H, W, ms, flops, floor in rows[:args.top]:
        name = "%s %d->%d %dx%d/%d B%d %dx%d" % (kind, cin, cout, kh, kw, s, B, H, W)
        cells = " | ".join("%8.1f (%5.1fx) %5.0fTF" % (m * 1e3, m / floor, flops / m / 1e9) for m in ms)
        print("%-34s %5d %8.2f | %s | %7.3f" % (name, n, flops / 1e9, cells, tot))
    far = sum(n * sum(m for m in ms if m > 4 * floor) for _, n, *_r, ms, _f, floor in rows)
    print("# time in launches more than 4x above their floor: %.3f ms of %.3f" % (far, total))


if __name__ == "__main__":
    main()
