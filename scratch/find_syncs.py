"""Host synchronisations inside the steady-state frame loop of the rollout (torch sync debug mode + timing of host phases)."""
import sys, os, time, torch, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoregressive_diffusion_amd import edm2 as _e  # noqa
from edm2.networks_edm2 import UNet, Precond
import edm2.sampler as S
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = UNet(**bench.GYM_CFG).to(dev)
torch.nn.init.constant_(unet.out_gain, 1.0)
net = Precond(unet, sigma_data=1.0).to(dev).eval()
B = 1
with torch.no_grad():
    ctx = torch.randn(B, 8, 8, 64, 64, device=dev)
    lab = torch.randint(0, 4, (B, 8), device=dev)
    _, cache = net(ctx, torch.ones(B, 8, device=dev) * 0.05, lab, update_cache=True)
    for i in range(3):
        _, _, _, cache = S.edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    host = []
    t00 = time.perf_counter()
    for i in range(6):
        t0 = time.perf_counter()
        x, _, _, cache = S.edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
        host.append((time.perf_counter() - t0) * 1e3)
    t1 = time.perf_counter()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("host ms per frame:", [round(h, 1) for h in host], "queue drain after loop %.1f ms" % ((t2 - t1) * 1e3), "total/frame %.1f" % ((t2 - t00) / 6 * 1e3))
