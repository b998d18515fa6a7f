"""Where the time of one generated frame goes (rollout, B=1): prewarm, graph capture, replays, the eager last evaluation."""
import sys, os, time, torch
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
acc = {}
def timed(name, f):
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return g
with torch.no_grad():
    ctx = torch.randn(B, 8, 8, 64, 64, device=dev)
    lab = torch.randint(0, 4, (B, 8), device=dev)
    _, cache = net(ctx, torch.ones(B, 8, device=dev) * 0.05, lab, update_cache=True)
    for i in range(2):
        _, _, _, cache = S.edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
    unet.prewarm_eval = timed("prewarm", unet.prewarm_eval)
    orig_run = S._GraphedDenoiser.run
    def run(self):
        name = "capture+1st replay" if self.graph is None else "replay"
        return timed(name, orig_run)(self)
    S._GraphedDenoiser.run = run
    net_fwd = net.forward
    def fwd(*a, **k):
        if k.get("update_cache") and not torch.cuda.is_current_stream_capturing():
            return timed("eager last eval", net_fwd)(*a, **k)
        return net_fwd(*a, **k)
    net.forward = fwd
    n = 6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        x, _, _, cache = S.edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("per frame %.2f ms (with syncs)" % (tot / n * 1e3))
for k, v in acc.items():
    print("  %-22s %.2f ms/frame" % (k, v / n * 1e3))
print("  other                  %.2f ms/frame" % ((tot - sum(acc.values())) / n * 1e3))
