"""Micro-benchmark of the HBM-bound kernels (RootTanh, InPlaceNorm, residual gate, softmax, resampling, FeaturePooling and the
attention gates) on the largest tensors of config 2 (64x64, batch 64): forward and backward through the C ABI
(pre-allocated operands, no host work in the loop), timed with HIP events; prints the ALGORITHMIC bytes (every distinct operand read or written once, fp32) per call and the
rate they imply against the 8 TB/s HBM peak.  Usage (GPU box): python tools/bench_elementwise.py [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd._lib import check, lib  # noqa: E402

PEAK = 8.0e12


def time_it(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64,
                    help="64: the step's own tensors (48-100 MB: they fit the 256 MB Infinity Cache between producer and consumer); 256: "
                         "192-400 MB per operand - every pass really goes to HBM")
    ap.add_argument("--all-shapes", action="store_true", help="every activation shape of the config-2 step instead of the three largest")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    B = args.batch
    print("%-30s %-16s %28s %28s" % ("op", "tensor", "forward  ms | GB/s | of peak", "backward ms | GB/s | of peak"))

    def report(name, shape, fwd, bwd, fe, be):
        tf, tb = time_it(fwd, args.reps), time_it(bwd, args.reps)
        # x model: time against 3 us + bytes at 5 TB/s (what a well-distributed streaming kernel takes on this chip)
        mf, mb = tf / (3e-6 + 4 * fe / 5e12), tb / (3e-6 + 4 * be / 5e12)
        print("%-30s %-16s %9.4f | %6.0f | %4.2f     %9.4f | %6.0f | %4.2f   x model %4.1f %4.1f%s" % (
            name, "x".join(map(str, shape)), tf * 1e3, 4 * fe / tf / 1e9, 4 * fe / tf / PEAK, tb * 1e3, 4 * be / tb / 1e9,
            4 * be / tb / PEAK, mf, mb, "  <<<" if max(mf, mb) > 2.0 else ""), flush=True)

    def P(t):
        return t.data_ptr()

    shapes = ((B, 96, 64, 64), (B, 48, 64, 64), (B, 192, 16, 16))
    if args.all_shapes:
        # every activation shape of the step (generator at batch B, discriminator at the stacked 3 B): the small and mid-sized maps are
        # where a kernel's work distribution, not the memory system, decides its time
        shapes = ((B, 768, 4, 4), (B, 768, 8, 8), (B, 384, 8, 8), (B, 384, 16, 16), (B, 192, 16, 16), (B, 192, 32, 32), (B, 96, 32, 32),
                  (B, 96, 64, 64), (B, 48, 64, 64), (3 * B, 32, 32, 32), (3 * B, 64, 16, 16), (3 * B, 128, 8, 8), (3 * B, 256, 4, 4),
                  (3 * B, 512, 2, 2), (3 * B, 512, 1, 1))
    for shape in shapes:
        Bn, C, H, W = shape
        hw, planes = H * W, Bn * C
        n = planes * hw
        x, g, y, gx = (torch.randn(shape, device=dev) for _ in range(4))
        report("RootTanh", shape, lambda: check(L.locate_roottanh_fwd(P(x), P(y), n, None, st)),
               lambda: check(L.locate_roottanh_bwd(P(x), P(g), P(gx), n, 0, None, st)), 2 * n, 3 * n)
        w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dw, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        stats = torch.empty(2, device=dev)
        ws_f = torch.empty(max(L.locate_norm_stats_workspace_bytes(), 16), dtype=torch.uint8, device=dev)
        ws_b = torch.empty(max(L.locate_norm_bwd_workspace_bytes(Bn, C), 16), dtype=torch.uint8, device=dev)
        for act in (0, 1):
            report("InPlaceNorm" + (" + RootTanh" if act else ""), shape,
                   lambda act=act: check(L.locate_norm_fwd(P(x), P(w), 0, P(b), P(y), act, P(stats), Bn, C, hw, 1, P(ws_f), None, None, st)),
                   lambda act=act: check(L.locate_norm_bwd(P(x), P(g), P(stats), P(w), 0, P(b), act, P(gx), P(dw), P(db), Bn, C, hw, 1,
                                                           P(ws_b), 0, st)), 3 * n, 5 * n)    # statistics pass + apply | plane sums + dx
        a = torch.randn(shape, device=dev)
        ac = torch.randn(Bn, C, device=dev)
        da, dac = torch.empty_like(a), torch.empty_like(ac)
        gam, dgam = torch.full((1,), 2.0, device=dev), torch.empty(1, device=dev)
        ws_g = torch.empty(max(L.locate_gate_bwd_workspace_bytes(planes), 16), dtype=torch.uint8, device=dev)
        report("gate, full attention map", shape, lambda: check(L.locate_gate_fwd(P(x), P(a), 0, P(gam), P(y), planes, hw, st)),
               lambda: check(L.locate_gate_bwd(P(x), P(a), 0, P(gam), P(g), P(gx), P(da), P(dgam), planes, hw, P(ws_g), 0, None, st)), 3 * n, 5 * n)
        report("gate, per-plane attention", shape, lambda: check(L.locate_gate_fwd(P(x), P(ac), 1, P(gam), P(y), planes, hw, st)),
               lambda: check(L.locate_gate_bwd(P(x), P(ac), 1, P(gam), P(g), P(gx), P(dac), P(dgam), planes, hw, P(ws_g), 0, None, st)), 2 * n, 3 * n)
        report("softmax over H*W", shape, lambda: check(L.locate_softmax_fwd(P(x), P(y), planes, hw, st)),
               lambda: check(L.locate_softmax_bwd(P(y), P(g), P(gx), planes, hw, st)), 2 * n, 3 * n)
        if H == 1:
            continue
        if H == 64 or (args.all_shapes and shape[0] != B):
            q = torch.empty(Bn, C, H // 2, W // 2, device=dev)
            gq = torch.randn_like(q)
            report("avgpool 2x2", shape, lambda: check(L.locate_avgpool2_fwd(P(x), P(q), planes, H, W, st)),
                   lambda: check(L.locate_avgpool2_bwd(P(gq), P(gx), planes, H, W, 0, st)), n + n // 4, n + n // 4)
            h2 = torch.empty(Bn, C // 2, H, W, device=dev)
            gh2 = torch.randn_like(h2)
            report("FeaturePooling / 2", shape, lambda: check(L.locate_feature_pool_fwd(P(x), P(h2), n // 2, 2, st)),
                   lambda: check(L.locate_feature_pool_bwd(P(gh2), P(gx), n // 2, 2, 0, st)), n + n // 2, n + n // 2)
        else:
            up = torch.empty(Bn, C, 2 * H, 2 * W, device=dev)
            gup = torch.randn_like(up)
            report("upsample x2 (bilinear)", shape, lambda: check(L.locate_upsample2x_fwd(P(x), P(up), planes, H, W, st)),
                   lambda: check(L.locate_upsample2x_bwd(P(gup), P(gx), planes, H, W, st)), 5 * n, 5 * n)


if __name__ == "__main__":
    main()
