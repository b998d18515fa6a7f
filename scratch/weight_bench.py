"""Time weight_prep / weight_bwd of the gym net in isolation (after one real 3-D step filled the slabs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoregressive_diffusion_amd import edm2 as _e  # noqa
from edm2.networks_edm2 import UNet, Precond
from edm2.loss import EDM2Loss
from autoregressive_diffusion_amd.parallel import FlatParams
from autoregressive_diffusion_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
unet = UNet(**bench.GYM_CFG).to(dev)
flat = FlatParams(unet, lazy_small=True)
net = Precond(unet, use_fp16=True, sigma_data=1.0).to(dev).train()
loss_fn = EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5)
B, T = 2, 64
lat = torch.randn(B, T, 8, 64, 64, device=dev)
act = torch.randint(0, 4, (B, T), device=dev)
loss, _ = loss_fn(net, lat, act, just_2d=False, sync=False)
bank = next(m for m in unet.modules() if hasattr(m, "weight") and hasattr(m.weight, "pw")).weight.pw.bank
saved = {}
orig = bank.backward
def spy():
    saved["ns"] = bank.nsplit_all.clone()
    orig()
bank.backward = spy
loss.backward()
torch.cuda.synchronize()
ns = saved["ns"]
nsl = ns.tolist()
slab_bytes = sum(nsl[i] * w.taps * w.CoutP * w.CinP * 2 for i, (w, _) in enumerate(bank.items) if getattr(w, "group", None) is None)
par = sum(w.param.numel() for w, _ in bank.items)
print("slab bytes %.1f MB, params %.1f M" % (slab_bytes / 1e6, par / 1e6))
print("nsplit: mean %.1f max %d" % (ns.float().mean().item(), ns.max().item()))
g0 = flat.grad.clone() if hasattr(flat, "grad") else None

def timeit(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

# weight_bwd consumes nsplit: restore it before each call (tiny copy)
def bwd():
    bank.nsplit_all.copy_(ns)
    ops.check(ops.lib.oniris_weight_bwd(ops._p(bank._dev_table), len(bank.items), bank.total_rows, ops._stream()), "weight_bwd")
def prep():
    ops.check(ops.lib.oniris_weight_prep(ops._p(bank._dev_table), len(bank.items), bank.total_rows, bank.total_tiles, 1, ops._stream()), "weight_prep")
print("weight_bwd  %.1f us" % timeit(bwd))
print("weight_prep %.1f us (prep + wb)" % timeit(prep))
