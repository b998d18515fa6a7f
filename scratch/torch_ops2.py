"""Which Python call sites launch the ATen kernels of one 3-D training step (torch.profiler with stacks)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
from torch.profiler import profile, ProfilerActivity
import bench as B
dev = torch.device("cuda:0")
from edm2.networks_edm2 import UNet, Precond
from edm2.loss import EDM2Loss
from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW, FlatEMA
torch.manual_seed(0)
unet = UNet(**B.GYM_CFG).to(dev)
for m in unet.modules():
    if hasattr(m, 'emb_gain'): torch.nn.init.constant_(m.emb_gain, 0.3)
torch.nn.init.constant_(unet.out_gain, 1.0)
flat = FlatParams(unet, lazy_small=True)
net = Precond(unet, use_fp16=True, sigma_data=1.0).to(dev).train()
opt = FlatAdamW(flat, lr=1e-2, eps=1e-8)
ema = FlatEMA(flat)
loss_fn = EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5)
lat = torch.randn(2, 64, 8, 64, 64, device=dev); act = torch.randint(0, 4, (2, 64), device=dev)
J2D = len(sys.argv) > 1 and sys.argv[1] == "2d"
def step():
    opt.zero_grad(); loss, _ = loss_fn(net, lat, act, just_2d=J2D, sync=False); loss.backward(); opt.step(max_norm=0.1, ema=ema.weights(1000, 2))
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = prof.events()
agg = collections.defaultdict(lambda: [0, 0.0])
nk = 0
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if not e.name.startswith("aten::"):
        continue
    nk += len(e.kernels)
    site = "?"
    for fr in (e.stack or []):
        if "/repo/" in fr and "torch/" not in fr:
            site = fr.split("/repo/")[-1]
            break
    if site == "?" and e.stack:
        site = "(autograd engine) " + next((fr for fr in e.stack if "Backward" in fr or "autograd" in fr), e.stack[0])[-70:]
    a = agg[(e.name, site)]
    a[0] += len(e.kernels); a[1] += sum(k.duration for k in e.kernels)
print("aten kernels in one step:", nk)
for (name, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{name:28s} n={n:4d} {us:9.1f} us  {site}")
