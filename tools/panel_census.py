"""Which weight panels a training iteration re-packs: per network, the number of stale panels by (direction, format) and their
bytes against the weights' own bytes.  usage (GPU box): python tools/panel_census.py"""
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
    torch.cuda.synchronize()
    for name, net in (("G", G), ("D", D)):
        tot_w = tot_p = 0
        kinds = {}
        both = 0
        for pn, p in net.named_parameters():
            cache = p.__dict__.get("_locate_panels")
            if not cache:
                continue
            tot_w += p.numel() * 4
            dirs = {}
            for key, (ver, buf, geom) in cache.items():
                tot_p += buf.numel()
                k = ("Rt" if key[0] & 1 else "R") + ("/f16" if key[0] & 2 else "/bf16")
                kinds[k] = kinds.get(k, [0, 0])
                kinds[k][0] += 1
                kinds[k][1] += buf.numel()
                dirs.setdefault(key[0] & 1, set()).add(key[0] & 2)
            both += sum(1 for v in dirs.values() if len(v) > 1)
        print("%s: weights with panels %.1f MB, panels %.1f MB (%.2fx); (layer, direction) pairs packed in BOTH formats: %d" % (name, tot_w / 1e6, tot_p / 1e6, tot_p / max(tot_w, 1), both))
        for k, (n, b) in sorted(kinds.items()):
            print("    %-8s %3d panels %8.1f MB" % (k, n, b / 1e6))


if __name__ == "__main__":
    main()
