import os, sys
sys.path.insert(0, "/root/repo")
os.environ["LOCATE_WINDOW"] = "all"
import numpy as np, torch
from locate_amd import Generator, Discriminator, NetConfig, ops
z = np.load("/root/repo/tests/golden/g8_tiny_e2e.npz", allow_pickle=False)
T = torch.as_tensor
dev = torch.device("cuda:0")
cfg = NetConfig(image_size=32, base_feature_factor=1)
G = Generator(cfg)
G.load_state_dict({k[len("G/sd0/"):]: T(z[k]) for k in z.files if k.startswith("G/sd0/")})
G.noise = T(z["G/noise"]).clone()
G = G.to(dev)
real = ops._contract
def spy(forward_of_r, x, w, owner, spec, geom, garr, sigma, bias, y, precision=0, amax=None, epilogue=None):
    out = real(forward_of_r, x, w, owner, spec, geom, garr, sigma, bias, y, precision, amax, epilogue)
    torch.cuda.synchronize()
    bad = not torch.isfinite(out).all().item()
    win = ops._win_ok(geom, garr, (0 if forward_of_r else 1) | (2 if amax is not None else 0), x) if spec.mode == "dense" else 0
    print(("NaN " if bad else "ok  "), spec.kind, "fwdR" if forward_of_r else "adj", tuple(geom), "win", win, "amax", amax is not None, "x finite", torch.isfinite(x).all().item(), flush=True)
    return out
ops._contract = spy
with torch.no_grad():
    out = G(T(z["step1/latent"]).to(dev))
print("final finite", torch.isfinite(out).all().item())
