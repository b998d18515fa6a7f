"""Per-call host time of the 31 graph replays of a frame, the sampler_update calls and the gaps between them (default and
ONIRIS_SAMPLER_KEEP_GRAPHS=1)."""
import sys, time, types, torch
sys.path.insert(0, ".")
import bench
from autoregressive_diffusion_amd import ops
import edm2.sampler as S
log = []
_replay = torch.cuda.CUDAGraph.replay
def replay(self):
    t = time.perf_counter(); _replay(self); log.append(("replay", t, time.perf_counter()))
torch.cuda.CUDAGraph.replay = replay
_ce = torch.cuda.CUDAGraph.capture_end
def ce(self):
    t = time.perf_counter(); _ce(self); log.append(("capture_end", t, time.perf_counter()))
torch.cuda.CUDAGraph.capture_end = ce
_cb = torch.cuda.CUDAGraph.capture_begin
def cb(self, *a, **k):
    t = time.perf_counter(); _cb(self, *a, **k); log.append(("capture_begin", t, time.perf_counter()))
torch.cuda.CUDAGraph.capture_begin = cb
_fin = S._GraphedDenoiser.finish_cache
def fin(self):
    t = time.perf_counter(); r = _fin(self); log.append(("finish_cache", t, time.perf_counter())); return r
S._GraphedDenoiser.finish_cache = fin
from edm2.networks_edm2 import UNet
_pw = UNet.prewarm_eval
def pw(self, cache):
    t = time.perf_counter(); r = _pw(self, cache); log.append(("prewarm", t, time.perf_counter())); return r
UNet.prewarm_eval = pw
out = bench.rollout(types.SimpleNamespace(batch=1, ctx_frames=8, gen_frames=8), quiet=True)
print(f"{out['value']:.2f} frames/s")
# the last full frame: from the last-but-one capture_begin on
cbs = [i for i, e in enumerate(log) if e[0] == "capture_begin"]
a, b = cbs[-2], cbs[-1]
t0 = log[a][1]
for name, s, e in log[a - 2:b + 1]:
    print(f"  {(s - t0) * 1e3:8.2f} ms  +{(e - s) * 1e3:7.3f} ms  {name}")
