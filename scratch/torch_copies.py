"""Which torch ops copy memory in one 3-D training step (torch.profiler, shapes recorded)."""
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
def step():
    opt.zero_grad(); loss, _ = loss_fn(net, lat, act, just_2d=False, sync=False); loss.backward(); opt.step(max_norm=0.1, ema=ema.weights(1000, 2))
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
names = ("aten::copy_", "aten::cat", "aten::stack", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_",
         "aten::add_", "aten::add", "aten::index_select", "aten::index_add_", "aten::mul", "aten::sum", "aten::zeros", "aten::zeros_like", "aten::empty_like")
for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count):
    if e.key in names:
        print(f"{e.key:20s} n={e.count:4d} dev_us={getattr(e, 'device_time_total', 0):9.1f} shapes={str(e.input_shapes)[:110]}")
mem = collections.Counter()
for e in prof.events():
    if "emcpy" in e.name or "emset" in e.name:
        mem[e.name] += 1
print(dict(mem))
