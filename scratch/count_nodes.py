"""Kernel launches of one cached one-frame evaluation (eager, B = 1), by name."""
import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoregressive_diffusion_amd import edm2 as _e, _lib  # noqa
from edm2.networks_edm2 import UNet, Precond
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = UNet(**bench.GYM_CFG).to(dev)
net = Precond(unet, sigma_data=1.0).to(dev).eval()
with torch.no_grad():
    ctx = torch.randn(1, 8, 8, 64, 64, device=dev); lab = torch.randint(0, 4, (1, 8), device=dev)
    _, cache = net(ctx, torch.ones(1, 8, device=dev) * 0.05, lab, update_cache=True)
    unet.prewarm_eval(cache)
    x = torch.randn(1, 1, 8, 64, 64, device=dev)
    net(x, torch.ones(1, 1, device=dev), lab[:, :1], cache=cache, update_cache=False)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net(x, torch.ones(1, 1, device=dev), lab[:, :1], cache=cache, update_cache=False)
        torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith("CUDA"):
        c[e.name[:60]] += 1
print("launches:", sum(c.values()))
for k, v in c.most_common(30):
    print("%4d %s" % (v, k))
