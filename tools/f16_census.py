"""Contractions of one training iteration by form: which launches of >= F16_MIN_FLOPS take the fp16-piece form and which fall back to the
six-product form because an operand arrives without its largest-magnitude words.  usage (GPU box): python tools/f16_census.py"""
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = NetConfig(image_size=64)
    torch.manual_seed(cfg.seed)
    G, GO = get_model(Generator(cfg), cfg.glr, dev)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    step = TrainStep(G, D, GO, DO)
    B, S = 64, 64
    args = (torch.randn(B, S, device=dev), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1))
    for _ in range(2):
        step(*args)
    log = OrderedDict()
    real = ops._f16_ok

    def spy(spec, geom, precision, *amax):
        ok = real(spec, geom, precision, *amax)
        fl = ops._flops(geom)
        if spec.mode == "dense" and fl >= ops.F16_MIN_FLOPS:
            import traceback
            who = [f.name for f in traceback.extract_stack(limit=6)][:-1]
            kind = "wgrad" if "_raw_weight_grad" in who else ("dgrad" if "_conv_input_grad" in who else "fwd")
            key = (kind, spec.kind, tuple(geom), tuple(a is not None for a in amax), bool(ok))
            log[key] = log.get(key, 0) + 1
        return ok

    ops._f16_ok = spy
    step(*args)
    torch.cuda.synchronize()
    ops._f16_ok = real
    miss = 0.0
    for (kind, sk, geom, have, ok), n in log.items():
        Bn, C, H, W, M, KH, KW, s_, ph, pw, OH, OW = geom
        fl = ops._flops(list(geom)) / 1e9
        if not ok:
            miss += fl * n
        print("%-5s %-5s B%-3d %4d->%-4d %dx%d/%d %3dx%-3d  %6.2f GF x%d  amax %s  %s" % (kind, sk, Bn, C, M, KH, KW, s_, H, W, fl, n, have, "f16" if ok else "SIX-PRODUCT"))
    print("GFLOP per iteration in six-product launches of >= %.1f GF: %.1f" % (ops.F16_MIN_FLOPS / 1e9, miss))


if __name__ == "__main__":
    main()
