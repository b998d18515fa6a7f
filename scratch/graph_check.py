import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edm2.networks_edm2 import UNet, Precond
from edm2.loss import EDM2Loss
from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
from autoregressive_diffusion_amd.graphs import GraphedStep
from bench import GYM_CFG
mode = sys.argv[1]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = UNet(**GYM_CFG).to(dev)
for m in unet.modules():
    if hasattr(m, "emb_gain"): torch.nn.init.constant_(m.emb_gain, 0.3)
torch.nn.init.constant_(unet.out_gain, 1.0)
flat = FlatParams(unet)
net = Precond(unet, sigma_data=1.0).to(dev).train()
opt = FlatAdamW(flat, lr=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2, eps=1e-8)
loss_fn = EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5)
g = torch.Generator(device=dev).manual_seed(1234)
TT = int(sys.argv[3]) if len(sys.argv) > 3 else 16
latents = torch.randn(2, TT, 8, 64, 64, device=dev, generator=g)
actions = torch.randint(0, 4, (2, TT), device=dev, generator=g)
def fwd_bwd(j2d):
    opt.zero_grad()
    loss, _ = loss_fn(net, latents, actions, just_2d=j2d, sync=False)
    loss.backward()
    return loss
graphed = {False: GraphedStep(lambda: fwd_bwd(False)), True: GraphedStep(lambda: fwd_bwd(True))}
out = []
for i in range(48):
    j2d = (i % 4 == 0)
    loss = graphed[j2d]() if mode == "graph" else fwd_bwd(j2d)
    opt.step()
    if i % 4 == 1: out.append(loss.detach().clone())
out = [round(float(o.item()), 3) for o in out]
print(mode, out, "grad finite", bool(torch.isfinite(flat.grad).all()), "param absmax", float(flat.flat.abs().max()))
