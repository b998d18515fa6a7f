"""Is a forward of the UNet bit-reproducible?  eval mode (weights untouched) vs training mode (forced weight normalisation
rewrites the parameters on every call, conv.py:16-18)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autoregressive_diffusion_amd  # noqa: F401  (puts the edm2 shim on the path)
from edm2.networks_edm2 import UNet
torch.manual_seed(0)
net = UNet(img_resolution=32, img_channels=16, label_dim=0, model_channels=32, channel_mult=[1, 2, 2, 4], num_blocks=3,
           video_attn_resolutions=[16, 8]).cuda()
with torch.no_grad():
    net.out_gain.fill_(1.0)
x = torch.randn(2, 16, 16, 32, 32, device="cuda")
nz = torch.zeros(2, 16, device="cuda")
with torch.no_grad():
    for mode in ("eval", "train"):
        getattr(net, mode)()
        ys = [net(x, nz, None)[0].clone() for _ in range(5)]
        w = [p.detach().clone() for p in net.parameters()]
        print(mode, "std(y_i - y_0):", [f"{(y - ys[0]).float().std().item():.2e}" for y in ys[1:]], "std(y)", f"{ys[0].std().item():.3f}",
              "| std(y_i - y_i-1):", [f"{(ys[i] - ys[i - 1]).float().std().item():.2e}" for i in range(2, 5)])
    net.train()
    w0 = [p.detach().clone() for p in net.parameters()]
    net(x, nz, None)
    d = max(((p.detach() - q).abs().max() / (q.abs().max() + 1e-12)).item() for p, q in zip(net.parameters(), w0))
    print("largest relative parameter change by one more training forward:", f"{d:.2e}")
