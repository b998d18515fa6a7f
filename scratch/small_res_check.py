"""The reference's test UNet (resolution 16: 2x2 bottom level): does the HIP path serve it, or refuse loudly?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autoregressive_diffusion_amd  # noqa: F401
from edm2.networks_edm2 import UNet
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oniris_oracle as O
torch.manual_seed(0)
kw = dict(img_resolution=16, img_channels=16, label_dim=0, model_channels=32, channel_mult=[1, 2, 2, 4], num_blocks=3,
          video_attn_resolutions=[16, 8])
net = UNet(**kw).cuda().train()
with torch.no_grad():
    net.out_gain.fill_(1.0)
x = torch.randn(2, 16, 16, 16, 16, device="cuda")
nz = torch.randn(2, 16, device="cuda") * 0.3
try:
    with torch.no_grad():
        y, _ = net(x, nz, None)
    print("ran: finite", bool(torch.isfinite(y).all()), "std", y.std().item())
    ref = O.UNet(**kw) if hasattr(O, "UNet") else None
    print("oracle UNet available:", ref is not None)
except Exception as e:
    print("raised:", type(e).__name__, str(e)[:300])
