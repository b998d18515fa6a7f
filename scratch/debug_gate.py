import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import paramgen
from test_model_gpu import build_precond, SMALL_CFG, load, T, DEV
from edm2.loss import EDM2Loss
from autoregressive_diffusion_amd import ops
z = load("g8_unet"); tag, mode = "small", "3d"
images, labels = T(z[tag + "_images"]).to(DEV), T(z[tag + "_labels"]).to(DEV)
net = build_precond(SMALL_CFG, int(z[tag + "_seed"]), 1.0).train()
sigma, eps = T(z[f"{tag}_{mode}_sigma"]).to(DEV), T(z[f"{tag}_{mode}_eps"]).to(DEV)
# monkeypatch to capture S1,S2 check
orig = ops._GatedConvFn.backward
def patched(ctx, dout):
    x, out, y3, ca, cb = ctx.saved_tensors
    res = orig(ctx, dout)
    B, Tt = ctx.dims
    N = x.shape[0]
    d = dout.float(); o = out.float()
    S1 = (d * o).sum((1, 2, 3))
    y3e = y3.float().reshape(B, 1, Tt, *y3.shape[1:]).expand(B, 2, Tt, *y3.shape[1:]).reshape(N, *y3.shape[1:])
    S2 = (d * y3e).sum((1, 2, 3))
    dca = (S1 - cb * S2) / ca
    mag = (d.abs() * o.abs()).sum((1, 2, 3))
    print("  layer Cout", out.shape[-1], "dca kern vs torch relerr", ((res[1] - dca).norm() / dca.norm()).item(),
          " |S1|/sum|terms| (cancellation)", (S1.abs() / mag).mean().item())
    return res
ops._GatedConvFn.backward = staticmethod(patched)
loss, unw = EDM2Loss(sigma_data=1.0)(net, images, labels, sigma=sigma, just_2d=False, noise=eps)
loss.backward()
prm = dict(net.named_parameters())
for k in z.files:
    pre = f"{tag}_{mode}_g_"
    if k.startswith(pre) and "gating" in k:
        n = k[len(pre):]
        print(n, np.round(prm[n].grad.flatten().cpu().numpy(), 6), "ref", np.round(z[k].flatten(), 6))
