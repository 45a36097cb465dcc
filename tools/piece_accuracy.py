"""CPU study (numpy): representational error of a length-K dot product when both fp32 operands are replaced by sums of
low-precision pieces and only some piece products are kept - the bf16 x 3 scheme of conv.hip (six products) against two fp16
pieces (four or three products), with and without a power-of-two scale per operand tensor.  Accumulation itself is done in
float64 for every scheme: what is compared is what the piece products can represent, not the summation order.
Usage: python tools/piece_accuracy.py"""
import numpy as np


def trunc_bf16(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def bf16x3(x):
    h = trunc_bf16(x)
    r = (x - h).astype(np.float32)
    m = trunc_bf16(r)
    return h, m, (r - m).astype(np.float32)


def fp16x2(x, scale=1.0):
    xs = (x * np.float32(scale)).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32)
    l = (xs - h).astype(np.float16).astype(np.float32)
    return h / np.float32(scale), l / np.float32(scale)


def pow2_scale(x, target=2.0 ** 14):
    return 2.0 ** np.floor(np.log2(target / np.abs(x).max()))


def study(name, a, w, K):
    a64, w64 = a.astype(np.float64), w.astype(np.float64)
    ref = (a64 * w64).reshape(-1, K).sum(1)
    mag = (np.abs(a64 * w64)).reshape(-1, K).sum(1)
    ah, am, al = (t.astype(np.float64) for t in bf16x3(a))
    wh, wm, wl = (t.astype(np.float64) for t in bf16x3(w))
    six = (ah * wh + ah * wm + am * wh + am * wm + ah * wl + al * wh).reshape(-1, K).sum(1)
    rows = [("bf16 x 3 pieces, 6 products", six)]
    for scaled in (False, True):
        sa, sw = (pow2_scale(a), pow2_scale(w)) if scaled else (1.0, 1.0)
        h1, l1 = (t.astype(np.float64) for t in fp16x2(a, sa))
        h2, l2 = (t.astype(np.float64) for t in fp16x2(w, sw))
        tag = "fp16 x 2 pieces%s" % (", 2^k scale per tensor" if scaled else ", unscaled")
        rows.append((tag + ", 4 products", (h1 * h2 + h1 * l2 + l1 * h2 + l1 * l2).reshape(-1, K).sum(1)))
        rows.append((tag + ", 3 products", (h1 * h2 + h1 * l2 + l1 * h2).reshape(-1, K).sum(1)))
    f32 = (a.astype(np.float32) * w.astype(np.float32)).astype(np.float64).reshape(-1, K).sum(1)     # one fp32 rounding per product
    rows.append(("(fp32 products rounded once each)", f32))
    print("%s   (K = %d, %d dot products; error relative to sum |a w|: max / rms)" % (name, K, ref.size))
    for tag, got in rows:
        e = np.abs(got - ref) / mag
        print("    %-52s %.2e / %.2e" % (tag, e.max(), np.sqrt((e ** 2).mean())))


def main():
    rng = np.random.default_rng(0)
    K, n = 3072, 4096
    study("activations ~ N(0,1) x weights ~ N(0, 0.02)", rng.standard_normal(n * K).astype(np.float32),
          (0.02 * rng.standard_normal(n * K)).astype(np.float32), K)
    study("gradients ~ 1e-5 N(0,1) x weights ~ N(0, 0.02)", (1e-5 * rng.standard_normal(n * K)).astype(np.float32),
          (0.02 * rng.standard_normal(n * K)).astype(np.float32), K)
    heavy = (rng.standard_normal(n * K) * np.exp(3.0 * rng.standard_normal(n * K))).astype(np.float32)      # 6 decades of dynamic range
    study("heavy-tailed activations (lognormal spread) x weights", heavy, (0.02 * rng.standard_normal(n * K)).astype(np.float32), K)


if __name__ == "__main__":
    main()
