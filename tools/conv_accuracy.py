"""Error of the conv forward / data gradient / weight gradient against a float64 CPU reference (normalised max error).
Run once per kernel flavour: LOCATE_HIP_DEBUG_LIBRARY=1 LOCATE_DISABLE=bx6,wbx6 (fp32 MFMA, debug library only) vs default (bf16 x 6)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from locate_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [("conv", 48, 64, 5, 2, 2, 8, 32, 32), ("conv", 5, 3, 3, 1, 1, 2, 6, 6), ("conv", 96, 96, 3, 1, 1, 16, 32, 32),
         ("convT", 192, 192, 4, 2, 1, 16, 8, 8), ("conv", 64, 64, 1, 1, 0, 16, 16, 16)]
print("flavour:", os.environ.get("LOCATE_DISABLE", "(default)"))
for kind, cin, cout, k, s, p, B, H, W in cases:
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = torch.randn(wshape) * 0.1
    x = torch.randn(B, cin, H, W)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, s, p) if kind == "conv" else F.conv_transpose2d(xd, wd, None, s, p)
    g = torch.randn_like(yd)
    yd.backward(g)
    wg = w.to(dev).requires_grad_(True)
    xg = x.to(dev).requires_grad_(True)
    h = wshape[0]
    u = torch.randn(h, device=dev)
    v = torch.randn(w.numel() // h, device=dev)
    sigma = torch.tensor([1.0, 1.0], device=dev)
    wv = torch.zeros(h, device=dev)
    spec = ops.ConvSpec(kind, k, k, s, p, p)
    y = ops.SNConvFn.apply(xg, wg, u, v, None, sigma, wv, spec)
    y.backward(g.float().to(dev))

    def err(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max())
    # the rank-1 spectral-norm term is added to gw in place: remove it again (dsigma * u v^T with sigma = 1)
    gw = wg.grad.double().cpu().reshape(h, -1)
    inner = float((wd.grad.reshape(h, -1) * wd.detach().reshape(h, -1)).sum())
    gw = gw + inner * torch.outer(u.double().cpu(), v.double().cpu())
    print("%-6s C%d->%d k%d s%d B%d %dx%d:  y %.2e  dx %.2e  dw %.2e" % (kind, cin, cout, k, s, B, H, W, err(y, yd.detach()),
          err(xg.grad, xd.grad), float((gw - wd.grad.reshape(h, -1)).abs().max() / wd.grad.abs().max())))
