"""Is a hipGraph replay asynchronous on the host?  Host time spent inside CUDAGraph.replay(), in ops.sampler_update and in the rest of a
generated frame (bench.py's rollout settings), against the frame's wall time."""
import sys, time, types, torch
sys.path.insert(0, ".")
import bench
from autoregressive_diffusion_amd import ops
acc = {"replay": 0.0, "n": 0, "update": 0.0}
_replay = torch.cuda.CUDAGraph.replay
def replay(self):
    t = time.perf_counter(); _replay(self); acc["replay"] += time.perf_counter() - t; acc["n"] += 1
torch.cuda.CUDAGraph.replay = replay
_upd = ops.sampler_update
def upd(*a, **k):
    t = time.perf_counter(); r = _upd(*a, **k); acc["update"] += time.perf_counter() - t; return r
ops.sampler_update = upd
import edm2.sampler as S
_run = S._GraphedDenoiser.run
cap = {"t": 0.0}
def run(self):
    first = self.graph is None
    t = time.perf_counter(); r = _run(self)
    if first: cap["t"] += time.perf_counter() - t
    return r
S._GraphedDenoiser.run = run
_pre = None
from edm2.networks_edm2 import UNet
_pw = UNet.prewarm_eval
pw = {"t": 0.0}
def prewarm(self, cache):
    t = time.perf_counter(); r = _pw(self, cache); pw["t"] += time.perf_counter() - t; return r
UNet.prewarm_eval = prewarm
out = bench.rollout(types.SimpleNamespace(batch=1, ctx_frames=8, gen_frames=16), quiet=True)
nf = 18
print(f"{out['value']:.2f} frames/s = {1e3 / out['value']:.2f} ms per frame")
print(f"host per frame: {acc['n'] / nf:.1f} replays, {acc['replay'] / nf * 1e3:.2f} ms inside replay() ({acc['replay'] / acc['n'] * 1e6:.0f} us each), "
      f"{acc['update'] / nf * 1e3:.2f} ms in sampler_update, first run() of a frame incl. capture {cap['t'] / nf * 1e3:.2f} ms, prewarm_eval {pw['t'] / nf * 1e3:.2f} ms")
# ---- second pass: which part of the first run() blocks
sub = {"begin": 0.0, "end": 0.0, "fwd": 0.0, "keeper": 0.0, "n": 0}
_cb, _ce = torch.cuda.CUDAGraph.capture_begin, torch.cuda.CUDAGraph.capture_end
def cb(self, *a, **k):
    t = time.perf_counter(); r = _cb(self, *a, **k); sub["begin"] += time.perf_counter() - t; sub["t0"] = time.perf_counter(); return r
def ce(self):
    sub["fwd"] += time.perf_counter() - sub["t0"]
    t = time.perf_counter(); r = _ce(self); sub["end"] += time.perf_counter() - t; sub["n"] += 1; return r
torch.cuda.CUDAGraph.capture_begin, torch.cuda.CUDAGraph.capture_end = cb, ce
cap["t"] = 0.0
out = bench.rollout(types.SimpleNamespace(batch=1, ctx_frames=8, gen_frames=16), quiet=True)
n = sub["n"]
print(f"per captured frame: capture_begin {sub['begin'] / n * 1e3:.2f} ms, forward under capture {sub['fwd'] / n * 1e3:.2f} ms, capture_end {sub['end'] / n * 1e3:.2f} ms; "
      f"whole first run() {cap['t'] / n * 1e3:.2f} ms")
# ---- third pass: never let go of a graph (is the destruction of an old graph what blocks?)
keep = []
_init = torch.cuda.CUDAGraph.__init__
class KeepGraph(torch.cuda.CUDAGraph):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k); keep.append(g); return g
torch.cuda.CUDAGraph = KeepGraph
cap["t"] = 0.0; sub.update(begin=0.0, end=0.0, fwd=0.0, n=0); acc.update(replay=0.0, n=0, update=0.0); pw["t"] = 0.0
_fin = S._GraphedDenoiser.finish_cache
fin = {"t": 0.0}
def finish(self):
    t = time.perf_counter(); r = _fin(self); fin["t"] += time.perf_counter() - t; return r
S._GraphedDenoiser.finish_cache = finish
_samp = S.edm_sampler_with_mse
tot = {"t": 0.0}
def samp(*a, **k):
    t = time.perf_counter(); r = _samp(*a, **k); tot["t"] += time.perf_counter() - t; return r
bench_mod = sys.modules["edm2.sampler"]; bench_mod.edm_sampler_with_mse = samp
out = bench.rollout(types.SimpleNamespace(batch=1, ctx_frames=8, gen_frames=16), quiet=True)
print(f"   host per frame: replay {acc['replay'] / nf * 1e3:.2f} ms ({acc['replay'] / max(1, acc['n']) * 1e6:.0f} us each), update {acc['update'] / nf * 1e3:.2f}, prewarm {pw['t'] / nf * 1e3:.2f}, finish_cache {fin['t'] / nf * 1e3:.2f}, whole sampler call {tot['t'] / nf * 1e3:.2f} ms")
print(f"graphs never destroyed: {out['value']:.2f} frames/s; whole first run() {cap['t'] / max(1, sub['n']) * 1e3:.2f} ms (capture part {(sub['begin'] + sub['fwd'] + sub['end']) / max(1, sub['n']) * 1e3:.2f} ms)")
