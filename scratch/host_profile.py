"""cProfile of the host side of training steps (enqueue only; the GPU runs behind)."""
import sys, os, cProfile, pstats, torch, time
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
    loss, _ = loss_fn(net, lat, act, just_2d=(i % 4 == 0), sync=False)
    loss.backward()
    opt.step()
for i in range(8): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(8): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {(t1-t0)/8*1e3:.2f} ms/step, with GPU drain {(t2-t0)/8*1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(8): step(i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
