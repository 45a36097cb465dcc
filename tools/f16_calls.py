import sys, os
sys.path.insert(0, os.getcwd())
import torch
from locate_amd import Discriminator, Generator, NetConfig, TrainStep, get_model, ops
dev = torch.device("cuda:0")
cfg = NetConfig(image_size=64)
torch.manual_seed(cfg.seed)
G, GO = get_model(Generator(cfg), cfg.glr, dev)
D, DO = get_model(Discriminator(cfg), cfg.dlr, dev)
G.batched_spectral_norm = D.batched_spectral_norm = True
step = TrainStep(G, D, GO, DO, overlap_wgrad=True)
B, S = 64, 64
args = (torch.randn(B, S, device=dev), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1), torch.randn(B, 3, S, S, device=dev).clamp(-1, 1))
for i in range(3):
    before = dict(ops.F16_CALLS)
    step(*args)
    torch.cuda.synchronize()
    print({k: ops.F16_CALLS[k] - before[k] for k in before})
