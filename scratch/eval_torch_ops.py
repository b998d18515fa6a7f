"""ATen ops (with shapes) inside ONE cached one-frame UNet evaluation of the rollout (eager)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
from torch.profiler import profile, ProfilerActivity
import bench as B
from edm2.networks_edm2 import UNet, Precond
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = UNet(**B.GYM_CFG).to(dev)
torch.nn.init.constant_(unet.out_gain, 1.0)
net = Precond(unet, sigma_data=1.0).to(dev).eval()
with torch.no_grad():
    ctx = torch.randn(1, 8, 8, 64, 64, device=dev); lab = torch.randint(0, 4, (1, 8), device=dev)
    _, cache = net(ctx, torch.ones(1, 8, device=dev) * 0.05, lab, update_cache=True)
    x = torch.randn(1, 1, 8, 64, 64, device=dev); t = torch.ones(1, 1, device=dev) * 3.0
    for _ in range(3):
        net(x, t, lab[:, :1], cache=cache, update_cache=False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        net(x, t, lab[:, :1], cache=cache, update_cache=False); torch.cuda.synchronize()
n = 0
for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count):
    if e.key.startswith("aten::") and getattr(e, "device_time_total", 0) > 0:
        n += e.count
        print(f"{e.key:24s} n={e.count:4d} dev_us={e.device_time_total:8.1f} shapes={str(e.input_shapes)[:100]}")
print("aten ops with device time:", n)
