"""HOST time of one generated frame of the rollout (B = 1), without any synchronisation: what the Python side spends in
prewarm_eval, the graph capture, the 30 other replays, the update launches and finish_cache, against the frame time.  The GPU idles
at a frame boundary for whatever of this exceeds the lead the host built up while the GPU ran the frame's evaluations."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoregressive_diffusion_amd import edm2 as _e, ops  # noqa
from edm2.networks_edm2 import UNet, Precond
import edm2.sampler as S

dev = torch.device("cuda", 0)
torch.manual_seed(0)
unet = UNet(**bench.GYM_CFG).to(dev)
torch.nn.init.constant_(unet.out_gain, 1.0)
net = Precond(unet, sigma_data=1.0).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
acc, cnt = {}, {}
def timed(name, f):
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; cnt[name] = cnt.get(name, 0) + 1
        return r
    return g
with torch.no_grad():
    ctx = torch.randn(B, 8, 8, 64, 64, device=dev)
    lab = torch.randint(0, 4, (B, 8), device=dev)
    _, cache = net(ctx, torch.ones(B, 8, device=dev) * 0.05, lab, update_cache=True)
    for i in range(2):
        _, _, _, cache = S.edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
    unet.prewarm_eval = timed("prewarm_eval", unet.prewarm_eval)
    orig_run = S._GraphedDenoiser.run
    marks = []
    def run(self):
        first = self.graph is None
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        if not first:
            e0.record()
        r = timed("run: capture + replay" if first else "run: replay", orig_run)(self)
        if first:
            e0 = None
        e1.record()
        marks.append((first, e0, e1))
        return r
    S._GraphedDenoiser.run = run
    S._GraphedDenoiser.finish_cache = timed("finish_cache", S._GraphedDenoiser.finish_cache)
    ops.sampler_update = timed("sampler_update", ops.sampler_update)
    for meth in ("capture_begin", "capture_end", "replay"):
        setattr(torch.cuda.CUDAGraph, meth, timed("  CUDAGraph." + meth, getattr(torch.cuda.CUDAGraph, meth)))
    net.forward = timed("  net.forward (inside the capture)", net.forward)
    n = 8
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        x, _, _, cache = S.edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01, sigma_max=80, rho=2)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
# GPU-side: time between the end of a frame's last replay and the end of the next frame's first replay, and replay durations
ends = [m[2] for m in marks]
firsts = [i for i, m in enumerate(marks) if m[0]]
bound = [ends[i - 1].elapsed_time(ends[i]) for i in firsts[1:]]
reps = [m[1].elapsed_time(m[2]) for m in marks if not m[0]]
print("GPU: last replay of a frame -> end of the next frame's first replay: %.2f ms (median of %d); a replay: %.3f ms median, %.3f mean" %
      (sorted(bound)[len(bound) // 2], len(bound), sorted(reps)[len(reps) // 2], sum(reps) / len(reps)))
print("B = %d: frame %.2f ms; host busy %.2f ms per frame (enqueue of all %d frames took %.1f of %.1f ms)" % (B, tot / n * 1e3, t_host / n * 1e3, n, t_host * 1e3, tot * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-24s %7.2f ms/frame  (%d calls/frame, %.3f ms each)" % (k, v / n * 1e3, cnt[k] // n, v / cnt[k] * 1e3))
print("  %-24s %7.2f ms/frame" % ("other host work", (t_host - sum(acc.values())) / n * 1e3))
