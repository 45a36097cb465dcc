"""Which panels every weight of the two networks holds after two config-2 steps: flags (bit 0 adjoint, bit 1 fp16 pieces, bit 2 window,
bit 3 fp8) and bytes - a (layer, direction) that holds both a gather and a window panel is re-packed twice per step.
usage (GPU box): python tools/panel_keys.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model  # noqa: E402


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
    for _ in range(3):
        step(*args)
    for name, net in (("G", G), ("D", D)):
        tot = dup = 0
        for pn, p in net.named_parameters():
            cache = p.__dict__.get("_locate_panels")
            if not cache:
                continue
            by_dir = {}
            for key, (ver, buf, geom) in cache.items():
                by_dir.setdefault(key[0] & 1, []).append((key[0], buf.numel()))
                tot += buf.numel()
            for d, lst in by_dir.items():
                if len(lst) > 1:
                    dup += sum(n for _, n in lst) - max(n for _, n in lst)
                    print("%s %-58s dir %d: %s" % (name, pn, d, ", ".join("flags %d: %.1f MB" % (f, n / 1e6) for f, n in lst)))
        print("%s: %.1f MB of panels, %.1f MB of them second panels of a (layer, direction)" % (name, tot / 1e6, dup / 1e6))


if __name__ == "__main__":
    main()
