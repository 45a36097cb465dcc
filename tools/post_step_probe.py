"""Post-optimizer-step norms and gamma gradients of one step of a full architecture, fp16-piece form on / off, against the
reference's fp32 and fp64 records (tests/golden/*_f64.npz).  usage: python tools/post_step_probe.py g12_config3_128 128 2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, ops  # noqa: E402

name, S, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + "_f64.npz"))
dev = torch.device("cuda:0")
res = {}
for mode in ("f16 pieces", "bf16 x 6"):
    ops.AMAX.enabled = mode == "f16 pieces"
    cfg = NetConfig(image_size=S)
    torch.manual_seed(cfg.seed)
    G, GO = get_model(Generator(cfg), cfg.glr, dev)
    D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
    G.batched_spectral_norm = D.batched_spectral_norm = True
    latent, real, aug = torch.randn(B, S), torch.randn(B, 3, S, S).clamp(-1, 1), torch.randn(B, 3, S, S).clamp(-1, 1)
    step = TrainStep(G, D, GO, DO)
    rec = {}
    g_orig = GO.step

    def g_hook():
        rec["gg"] = {k: float(p.grad.double().sum()) for k, p in G.named_parameters() if p.grad is not None and k.endswith("gamma")}
        return g_orig()
    GO.step = g_hook
    step(latent.to(dev), real.to(dev), aug.to(dev))
    res[mode] = ({k: float(v.double().norm()) for k, v in G.state_dict().items()}, rec["gg"])
k32, n32, n64 = z["f32/G/post_keys"].tolist(), z["f32/G/post_norms"], z["f64/G/post_norms"]
lr = 5e-4
print("post-step norms of G, deviation from the reference's fp64 run in units of lr:   ref32 | f16 pieces | bf16 x 6")
rows = []
for k, a, b in zip(k32, n32, n64):
    rows.append((max(abs(res[m][0][k] - b) for m in res), k, abs(a - b) / lr, abs(res["f16 pieces"][0][k] - b) / lr, abs(res["bf16 x 6"][0][k] - b) / lr))
for _, k, r, f, s in sorted(rows, reverse=True)[:12]:
    print("  %-82s %.4f | %.4f | %.4f" % (k, r, f, s))
print("gamma gradients of G, relative deviation from fp64:   ref32 | f16 pieces | bf16 x 6")
for k, a, b in zip(z["f32/G/gamma_keys"].tolist(), z["f32/G/gamma_grads"], z["f64/G/gamma_grads"]):
    print("  %-50s %.2e | %.2e | %.2e" % (k, abs(a - b) / abs(b), abs(res["f16 pieces"][1][k] - b) / abs(b), abs(res["bf16 x 6"][1][k] - b) / abs(b)))
