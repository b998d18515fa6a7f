"""Which python lines issue the small torch kernels of a training step (forward part; backward ops are attributed to
their autograd node names)."""
import sys, collections, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import bench
from edm2.networks_edm2 import UNet, Precond
from edm2.loss import EDM2Loss
from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = UNet(**bench.GYM_CFG).to(dev)
flat = FlatParams(unet, lazy_small=True)
net = Precond(unet, use_fp16=True, sigma_data=1.0).to(dev).train()
opt = FlatAdamW(flat, lr=1e-2)
loss_fn = EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5)
lat = torch.randn(2, 64, 8, 64, 64, device=dev); act = torch.randint(0, 4, (2, 64), device=dev)
def step(i):
    opt.zero_grad()
    loss, _ = loss_fn(net, lat, act, just_2d=False, sync=False)
    loss.backward()
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(1); torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"): continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"): continue
    dt = sum(k.duration for k in e.kernels) if hasattr(e, "kernels") else 0
    if dt == 0: continue
    st = [s for s in (e.stack or []) if "/root/repo" in s or "repo/" in s]
    par = e.cpu_parent.name if e.cpu_parent is not None else ""
    key = (e.name, (st[0].split("/")[-1][:60] if st else par[:60]))
    cnt[key] += 1; tim[key] += dt
tot = sum(tim.values())
print("torch op GPU time per step (us):", tot)
for k, v in tim.most_common(40):
    print(f"{v:8.0f} us {cnt[k]:4d}x {k[0]:24s} {k[1]}")
