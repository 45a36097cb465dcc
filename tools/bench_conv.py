"""Micro-benchmark of the implicit-GEMM kernels on the shapes of config 2 (64x64, batch 64): forward, data
gradient and weight gradient of every distinct conv stage, timed with HIP events on the launch stream.
Usage (GPU box): python tools/bench_conv.py [--reps 20]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import ops  # noqa: E402
from locate_amd._lib import check, lib  # noqa: E402

# name, kind, Cin, Cout, k, s, p, H (input side), B
SHAPES = [
    ("G.b1 convT4x4 768", "convT", 768, 768, 4, 2, 1, 4, 64),
    ("G.b2 convT4x4 384", "convT", 384, 384, 4, 2, 1, 8, 64),
    ("G.b3 convT4x4 192", "convT", 192, 192, 4, 2, 1, 16, 64),
    ("G.b4 convT4x4 96", "convT", 96, 96, 4, 2, 1, 32, 64),
    ("G.b4 convT1x1 96->48", "convT", 96, 48, 1, 1, 0, 64, 64),
    ("G.b3 convT1x1 192->96", "convT", 192, 96, 1, 1, 0, 32, 64),
    ("G.out conv3x3 48", "conv", 48, 48, 3, 1, 1, 64, 64),
    ("G.sa conv1x1 48 (N=4096)", "conv", 48, 48, 1, 1, 0, 64, 64),
    ("D.stem conv5x5 3", "conv", 3, 3, 5, 2, 2, 64, 64),
    ("D.b0 conv5x5 32", "conv", 32, 32, 5, 2, 2, 32, 64),
    ("D.b1 conv5x5 64", "conv", 64, 64, 5, 2, 2, 16, 64),
    ("D.b2 conv5x5 128", "conv", 128, 128, 5, 2, 2, 8, 64),
    ("D.b3 conv5x5 256", "conv", 256, 256, 5, 2, 2, 4, 64),
    ("D.b4 conv5x5 512", "conv", 512, 512, 5, 2, 2, 2, 64),
    ("D.b3 conv1x1 256->512", "conv", 256, 512, 1, 1, 0, 2, 64),
    ("D.b0 conv5x5 32 B192", "conv", 32, 32, 5, 2, 2, 32, 192),
    ("D.b1 conv5x5 64 B192", "conv", 64, 64, 5, 2, 2, 16, 192),
    ("D.b2 conv5x5 128 B192", "conv", 128, 128, 5, 2, 2, 8, 192),
    ("D.b3 conv5x5 256 B192", "conv", 256, 256, 5, 2, 2, 4, 192),
    ("G.b1 convT1x1 768->384", "convT", 768, 384, 1, 1, 0, 8, 64),
    ("G.b2 convT1x1 384->192", "convT", 384, 192, 1, 1, 0, 16, 64),
    ("G.sa conv1x1 192 (16x16)", "conv", 192, 192, 1, 1, 0, 16, 64),
    ("D.b2 conv1x1 128 B192", "conv", 128, 128, 1, 1, 0, 8, 192),
    ("D.b3 conv1x1 256 B192", "conv", 256, 256, 1, 1, 0, 4, 192),
]


def time_it(fn, reps):
    """Milliseconds per call, the calls replayed from a captured hipGraph: small launches are otherwise paced by the host
    (ctypes + planning: ~10 us per call), which the step's own graph replay does not pay either."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (3 * reps)


CHECK = False          # --check: forward / input gradient against a float64 CPU convolution
ZEROS = False          # --zeros: all-zero operands (the chip holds a higher clock on them: separates clock-bound from issue-bound)


WIN = False            # --win: the window form (csrc/convwin.hip) wherever the geometry has it


def bench_shape(kind, cin, cout, kh, kw, s, ph, pw, B, H, W, reps=20, prec=0, use_cnt=True, dev=None):
    """(ms forward, ms input gradient, ms weight gradient, algorithmic FLOPs, bytes forward) of one layer through the C ABI."""
    dev = dev or torch.device("cuda:0")
    L = lib()
    S = lambda: torch.cuda.current_stream().cuda_stream          # evaluated per call: the graph capture runs on a side stream
    spec = ops.ConvSpec(kind, kh, kw, s, ph, pw)
    wshape = (cout, cin, kh, kw) if kind == "conv" else (cin, cout, kh, kw)
    w = torch.randn(wshape, device=dev) * 0.05
    x = torch.randn(B, cin, H, W, device=dev)
    if ZEROS:
        w.zero_()
        x.zero_()
    geom, out_shape = spec.geometry(tuple(x.shape), tuple(w.shape))
    garr = (ctypes.c_int * 12)(*geom)
    y = torch.empty(out_shape, device=dev)
    gy = torch.randn(out_shape, device=dev)
    if ZEROS:
        gy.zero_()
    gx = torch.empty_like(x)
    gw = torch.empty_like(w)
    one = torch.ones(1, device=dev)
    fmt = 2 if prec == 2 else 0          # panel format bit: fp16-piece planes
    # window form per direction (R forward gathers x for a conv, gy for a transposed conv)
    in0, in1 = (x, gy) if kind == "conv" else (gy, x)
    # WIN: False = gather kernels, True = the window form wherever it exists, "auto" = where the library recommends it (what the
    # training step takes: locate_conv_win_ok == 1); never for fp8 / bf16-operand runs of this tool
    def takes_window(code):
        return bool(WIN) and prec in (0, 2) and (code == 1 or (code == 2 and WIN is True))
    win0 = 4 if takes_window(L.locate_conv_win_ok(garr, 0 | fmt, in0.stride(0), in0.data_ptr())) else 0
    win1 = 4 if takes_window(L.locate_conv_win_ok(garr, 1 | fmt, in1.stride(0), in1.data_ptr())) else 0
    pan0 = torch.empty(max(L.locate_conv_panel_bytes(garr, 0 | fmt | win0), 16), dtype=torch.uint8, device=dev)
    pan1 = torch.empty(max(L.locate_conv_panel_bytes(garr, 1 | fmt | win1), 16), dtype=torch.uint8, device=dev)
    check(L.locate_conv_pack_panel(garr, 0 | fmt | win0, w.data_ptr(), pan0.data_ptr(), S()))
    check(L.locate_conv_pack_panel(garr, 1 | fmt | win1, w.data_ptr(), pan1.data_ptr(), S()))
    nw = L.locate_absmax_words()
    amax = torch.zeros(2 * nw, dtype=torch.int32, device=dev)       # absmax words of x and gy
    check(L.locate_absmax(x.data_ptr(), x.numel(), amax[0:].data_ptr(), S()))
    check(L.locate_absmax(gy.data_ptr(), gy.numel(), amax[nw:].data_ptr(), S()))
    slot = {x.data_ptr(): amax[0:].data_ptr(), gy.data_ptr(): amax[nw:].data_ptr()}
    ws_f = torch.empty(max(L.locate_conv_win_workspace_bytes(garr, 0 | fmt) if win0 else L.locate_conv_fwd_workspace_bytes(garr), 16), dtype=torch.uint8, device=dev)
    ws_d = torch.empty(max(L.locate_conv_win_workspace_bytes(garr, 1 | fmt) if win1 else L.locate_conv_dgrad_workspace_bytes(garr), 16), dtype=torch.uint8, device=dev)
    part = torch.empty(L.locate_conv_wgrad_partials(garr), dtype=torch.float64, device=dev)
    ws_w = torch.empty(max(L.locate_conv_wgrad_workspace_bytes(garr), 16), dtype=torch.uint8, device=dev)
    cnt_f = torch.zeros(L.locate_conv_counter_bytes(), dtype=torch.uint8, device=dev)
    cnt_d = torch.zeros(L.locate_conv_counter_bytes(), dtype=torch.uint8, device=dev)

    def r_fwd(inp, out):      # R forward
        check(L.locate_conv_fwd(garr, inp.data_ptr(), inp.stride(0), pan0.data_ptr(), one.data_ptr(), 0, 0, None, out.data_ptr(),
                                out.stride(0), ws_f.data_ptr(), cnt_f.data_ptr() if use_cnt else None, prec | (16 if win0 else 0), slot[inp.data_ptr()] if prec == 2 else None, None, S()))

    def r_dgrad(inp, out):    # R data adjoint
        check(L.locate_conv_dgrad(garr, inp.data_ptr(), inp.stride(0), pan1.data_ptr(), one.data_ptr(), 0, 0, None, out.data_ptr(),
                                  out.stride(0), ws_d.data_ptr(), cnt_d.data_ptr() if use_cnt else None, prec | (16 if win1 else 0), slot[inp.data_ptr()] if prec == 2 else None, None, S()))

    if kind == "conv":
        fwd, dgr = (lambda: r_fwd(x, y)), (lambda: r_dgrad(gy, gx))
        wgr = lambda: check(L.locate_conv_wgrad(garr, x.data_ptr(), x.stride(0), gy.data_ptr(), gy.stride(0), gw.data_ptr(),
                                                w.data_ptr(), one.data_ptr(), 0, 0, part.data_ptr(), ws_w.data_ptr(), prec,
                                                slot[x.data_ptr()] if prec == 2 else None, slot[gy.data_ptr()] if prec == 2 else None, None, S()))
        flops = 2.0 * B * out_shape[2] * out_shape[3] * cout * cin * kh * kw
    else:
        fwd, dgr = (lambda: r_dgrad(x, y)), (lambda: r_fwd(gy, gx))
        wgr = lambda: check(L.locate_conv_wgrad(garr, gy.data_ptr(), gy.stride(0), x.data_ptr(), x.stride(0), gw.data_ptr(),
                                                w.data_ptr(), one.data_ptr(), 0, 0, part.data_ptr(), ws_w.data_ptr(), prec,
                                                slot[gy.data_ptr()] if prec == 2 else None, slot[x.data_ptr()] if prec == 2 else None, None, S()))
        flops = 2.0 * B * H * W * cout * cin * kh * kw        # every input pixel meets every tap once
    if CHECK:
        import torch.nn.functional as F
        fwd(); dgr()
        torch.cuda.synchronize()
        xd, wd, gd = x.double().cpu(), w.double().cpu(), gy.double().cpu()
        if kind == "conv":
            yr = F.conv2d(xd, wd, None, s, (ph, pw))
            gr = F.conv_transpose2d(gd, wd, None, s, (ph, pw), output_padding=(x.shape[2] - ((gd.shape[2] - 1) * s - 2 * ph + kh), x.shape[3] - ((gd.shape[3] - 1) * s - 2 * pw + kw)))
        else:
            yr = F.conv_transpose2d(xd, wd, None, s, (ph, pw))
            gr = F.conv2d(gd, wd, None, s, (ph, pw))
        wgr()
        torch.cuda.synchronize()
        xr, wr = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
        (F.conv2d(xr, wr, None, s, (ph, pw)) if kind == "conv" else F.conv_transpose2d(xr, wr, None, s, (ph, pw))).backward(gd)
        print("    max |err| / max |ref|:  fwd %.2e   dgrad %.2e   wgrad %.2e" % (
            float((y.double().cpu() - yr).abs().max() / yr.abs().max()), float((gx.double().cpu() - gr).abs().max() / gr.abs().max()),
            float((gw.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max())))
    if WIN:
        print("    window form: R forward %s, R adjoint %s" % (bool(win0), bool(win1)), file=sys.stderr)
    ms = [time_it(f, reps) for f in (fwd, dgr, wgr)]
    nbytes = 4.0 * (x.numel() + y.numel()) + 6.0 * w.numel()      # activations once each + the three bf16 weight planes
    return ms, flops, nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the stage name")
    ap.add_argument("--bf16", action="store_true", help="bf16 operands (precision 1) instead of the fp32-faithful splits")
    ap.add_argument("--no-counters", action="store_true", help="NULL arrival counters: split-K partial tiles summed by the reduction kernel")
    ap.add_argument("--shape", action="append", default=[], help="extra stage: kind,Cin,Cout,k,stride,pad,H,B (replaces the list)")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands")
    ap.add_argument("--f16", action="store_true", help="fp32-faithful with two scaled fp16 pieces per operand (precision 2)")
    ap.add_argument("--check", action="store_true", help="print the forward / input-gradient error against float64")
    ap.add_argument("--win", action="store_true", help="the window form (csrc/convwin.hip) wherever the geometry has it")
    ap.add_argument("--win-auto", action="store_true", help="the window form where the library recommends it (what the training step takes)")
    args = ap.parse_args()
    global ZEROS, CHECK, WIN
    ZEROS = args.zeros
    CHECK = args.check
    WIN = True if args.win else ("auto" if args.win_auto else False)
    print("%-28s %10s %10s %10s   (ms | TFLOP/s)" % ("stage", "fwd", "dgrad", "wgrad"))
    tot = [0.0, 0.0, 0.0]
    shapes = SHAPES
    if args.shape:
        shapes = []
        for t in args.shape:
            f = t.split(",")
            shapes.append((t, f[0]) + tuple(int(v) for v in f[1:]))
    for name, kind, cin, cout, k, s, p, H, B in shapes:
        if args.only and args.only not in name:
            continue
        ms, flops, _ = bench_shape(kind, cin, cout, k, k, s, p, p, B, H, H, args.reps, 2 if args.f16 else (1 if args.bf16 else 0), not args.no_counters)
        for i in range(3):
            tot[i] += ms[i]
        print("%-28s %s" % (name, "  ".join("%6.3f|%6.1f" % (m, flops / m / 1e9) for m in ms)), flush=True)
    print("%-28s %s" % ("sum (ms)", "  ".join("%13.3f" % t for t in tot)))


if __name__ == "__main__":
    main()
