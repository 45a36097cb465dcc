"""Worst relative deviations of per-tensor gradient norms from the reference's record for a full architecture
(tests/golden/g11..g14).  usage: python tools/full_arch_errors.py g12_config3_128 128 2 8   [LOCATE_HIP_DEBUG_LIBRARY=1 LOCATE_DISABLE=... to pick kernels]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model

name, S, B, ff = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".npz"))
cfg = NetConfig(image_size=S, base_feature_factor=ff)
torch.manual_seed(cfg.seed)
dev = torch.device("cuda:0")
G, GO = get_model(Generator(cfg), cfg.glr, dev)
D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
G.batched_spectral_norm = D.batched_spectral_norm = True
latent = torch.randn(B, S)
real = torch.randn(B, 3, S, S).clamp(-1, 1)
aug = torch.randn(B, 3, S, S).clamp(-1, 1)
step = TrainStep(G, D, GO, DO)
rec = {}
d_orig, g_orig = DO.step, GO.step
DO.step = lambda: (rec.__setitem__("D", {k: float(p.grad.double().norm()) for k, p in D.named_parameters() if p.grad is not None}), d_orig())[1]
GO.step = lambda: (rec.__setitem__("G", {k: float(p.grad.double().norm()) for k, p in G.named_parameters() if p.grad is not None}), g_orig())[1]
out = step(latent.to(dev), real.to(dev), aug.to(dev))
print("kernels disabled:", os.environ.get("LOCATE_DISABLE", "(none)"))
for k in ("d_error", "penalty", "g_error"):
    print("  %-8s got %.8g want %.8g" % (k, float(out[k]), float(z[k])))
for tag in ("D", "G"):
    want = dict(zip(z[tag + "/grad_keys"].tolist(), z[tag + "/grad_norms"]))
    scale = max(want.values())
    rows = sorted(((abs(rec[tag][k] - w) / max(w, 1e-3 * scale), k, rec[tag][k], w) for k, w in want.items()), reverse=True)
    for r in rows[:6]:
        print("  %s %.2e  %-72s got %.8g want %.8g" % ((tag,) + r))
