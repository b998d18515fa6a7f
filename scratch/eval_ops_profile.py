"""Which python lines issue the device copies / tiny torch kernels of ONE cached single-frame UNet evaluation."""
import sys, collections, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import bench
from edm2.networks_edm2 import UNet, Precond
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = Precond(UNet(**bench.GYM_CFG).to(dev), sigma_data=1.0).to(dev).eval()
with torch.no_grad():
    ctx = torch.randn(1, 8, 8, 64, 64, device=dev); lab = torch.randint(0, 4, (1, 8), device=dev)
    _, cache = net(ctx, torch.ones(1, 8, device=dev) * 0.05, lab, update_cache=True)
    x = torch.randn(1, 1, 8, 64, 64, device=dev); t = torch.ones(1, 1, device=dev)
    for _ in range(3):
        net(x, t, lab[:, :1], cache=cache, update_cache=False)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        net(x, t, lab[:, :1], cache=cache, update_cache=False)
        torch.cuda.synchronize()
cnt = collections.Counter(); kcnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::") or (e.cpu_parent is None and e.name.startswith("aten::")):
        st = [s for s in (e.stack or []) if "autoregressive_diffusion_amd" in s or "edm2" in s]
        cnt[(e.name, st[0].split("/")[-1] if st else "?")] += 1
for (n, s), c in cnt.most_common(60):
    print(f"{c:4d} {n:28s} {s}")
