"""Which torch ops launch the small elementwise kernels of a training step (torch.profiler, one 3-D step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
import importlib.util
from torch.profiler import profile, ProfilerActivity
import bench as B
dev = torch.device("cuda:0")
from edm2.networks_edm2 import UNet, Precond
from edm2.loss import EDM2Loss
from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
torch.manual_seed(0)
unet = UNet(**B.GYM_CFG).to(dev)
for m in unet.modules():
    if hasattr(m, 'emb_gain'): torch.nn.init.constant_(m.emb_gain, 0.3)
torch.nn.init.constant_(unet.out_gain, 1.0)
flat = FlatParams(unet, lazy_small=True)
net = Precond(unet, use_fp16=True, sigma_data=1.0).to(dev).train()
opt = FlatAdamW(flat, lr=1e-2, eps=1e-8)
loss_fn = EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5)
lat = torch.randn(2, 64, 8, 64, 64, device=dev); act = torch.randint(0, 4, (2, 64), device=dev)
def step():
    opt.zero_grad(); loss, _ = loss_fn(net, lat, act, just_2d=False, sync=False); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") or "Backward" in e.key]
rows.sort(key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.key[:60]:60s} n={e.count:5d} cuda_ms={getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0))/1e3:8.2f} cpu_ms={e.cpu_time_total/1e3:8.2f}")
