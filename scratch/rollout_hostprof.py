"""cProfile of the rollout's host side at a deep context (where does the wall time per frame go?)."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
import bench as B
from edm2.networks_edm2 import UNet, Precond
from edm2.sampler import edm_sampler_with_mse
dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = UNet(**B.GYM_CFG).to(dev)
torch.nn.init.constant_(unet.out_gain, 1.0)
net = Precond(unet, sigma_data=1.0).to(dev).eval()
ctxf = int(sys.argv[1]) if len(sys.argv) > 1 else 248
with torch.no_grad():
    ctx = torch.randn(1, ctxf, 8, 64, 64, device=dev); lab = torch.randint(0, 4, (1, ctxf), device=dev)
    _, cache = net(ctx, torch.ones(1, ctxf, device=dev) * 0.05, lab, update_cache=True)
    for i in range(2):
        _, _, _, cache = edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    for i in range(3):
        _, _, _, cache = edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
    torch.cuda.synchronize()
    pr.disable(); dt = time.perf_counter() - t0
print("ms per frame", dt / 3 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
