"""EDM / Heun sampler that generates ONE next frame against a cache (reference edm2/sampler.py:12-85):
rho-schedule, optional churn, Euler + 2nd-order correction, cache updated on the last Euler evaluation only."""
import os
import numpy as np
import torch


SAMPLER_GRAPH = int(os.environ.get("ONIRIS_SAMPLER_GRAPH", "1"))
FUSED_FRAME = int(os.environ.get("ONIRIS_SAMPLER_FUSED", "1"))      # 0: the tensor-expression loop below (A/B aid)
# 1: never let go of a frame's graph while the process lives (experiment: destroying a finished hipGraph exec blocks the host until the
# device is idle -- 23 ms per frame inside CUDAGraph's destructor, scratch/r06_replay_host.py)
KEEP_GRAPHS = int(os.environ.get("ONIRIS_SAMPLER_KEEP_GRAPHS", "0"))
REPLAY_SAME_STREAM = int(os.environ.get("ONIRIS_SAMPLER_SAME_STREAM", "1"))      # 0: replays on the capture stream (rounds 1-5; A/B)
_graph_pool = None
_pool_keeper = []          # [(graph, event recorded behind its last replay)]


class _GraphedDenoiser:
    """The cache-reading UNet evaluations of ONE generated frame (30 of its 31: same cache, same shapes, only x and
    sigma change) replayed from a hipGraph: the per-frame-count tables (RoPE, gate counters) are built first
    (UNet.prewarm_eval), the first evaluation is captured, the rest are replays -- ~400 kernel launches per
    evaluation without their host cost.  The last evaluation of the frame updates the cache and stays eager.  A new graph per
    frame: the KV length and the cache tensors change with every frame."""

    def __init__(self, net, cache, conditioning, B, dtype, device):
        self.net, self.cache, self.cond = net, cache, conditioning
        self.t = torch.ones(B, 1, device=device, dtype=dtype)
        self.x, self.out, self.graph, self.calls = None, None, None, 0

    def __call__(self, x, t):
        global _graph_pool
        self.calls += 1
        if self.calls == 1:
            unet = getattr(self.net, "unet", None)
            if hasattr(unet, "prewarm_eval"):                      # tables for this frame count, built outside the capture
                unet.prewarm_eval(self.cache)
            else:                                                  # unknown net: one eager evaluation builds them
                Dx, _ = self.net(x, self.t * t, self.cond, cache=self.cache, update_cache=False, just_2d=False)
                return Dx
        self.t.fill_(1.0).mul_(t)
        if self.graph is None:
            self.x = x.clone()
            cur = torch.cuda.current_stream()
            if _graph_pool is None:
                _graph_pool = (torch.cuda.graph_pool_handle(), torch.cuda.Stream())
            pool, side = _graph_pool
            side.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=side):
                self.out, _ = self.net(self.x, self.t, self.cond, cache=self.cache, update_cache=False, just_2d=False)
            self.graph, self.side = g, side
        else:
            self.x.copy_(x)
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self.graph.replay()
        cur.wait_stream(self.side)
        return self.out.clone()

    # -- protocol of the fused frame loop (_fused_frame): the caller keeps x and sigma in the graph's own input buffers
    # (written by ops.sampler_update), so an evaluation is a replay and nothing else.  The graph is captured with
    # update_cache=True on a SHADOW of the cache (same tensors, copied dicts): every replay then also leaves behind what
    # the cache update of that evaluation would be -- the shifted activation pairs as fresh tensors, the new key / value
    # frame behind the committed ones of the KV rings -- without touching anything the evaluations read.  After the last
    # replay (the frame's last Euler evaluation, the one the reference updates the cache on, sampler.py:63) the shadow IS
    # the updated cache: no separate eager evaluation (4.2 ms against 1.4 ms for a replay at B = 1).  The caller's cache
    # keeps the old tensors alive for as long as the graph reads them.
    def prepare(self, x0, t0):
        self.x = x0.clone()
        self.t.fill_(t0)
        self.new_cache = None

    @staticmethod
    def _shadow(c):
        return {k: _GraphedDenoiser._shadow(v) for k, v in c.items()} if isinstance(c, dict) else c

    def finish_cache(self):
        """The updated cache after the frame's last replay: every gated conv's (old pair, input of the last evaluation)
        becomes the shifted pair (reference conv.py:83-86), as fresh tensors outside the graph's memory."""
        def walk(c):
            if not isinstance(c, dict):
                return
            fr = c.pop("_pending_frame", None)
            if fr is not None:
                c["activations"] = torch.cat([c["activations"][:, 1:], fr], dim=1)
            for v in c.values():
                walk(v)
        walk(self.new_cache)
        return self.new_cache

    def run(self):
        global _graph_pool
        cur = torch.cuda.current_stream()
        if self.graph is None:
            if _graph_pool is None:
                _graph_pool = (torch.cuda.graph_pool_handle(), torch.cuda.Stream())
            pool, side = _graph_pool
            side.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            shadow = self._shadow(self.cache)
            from .conv import DEFER_CACHE_SHIFT
            DEFER_CACHE_SHIFT[0] = True
            # capture_begin / capture_end by hand: the `torch.cuda.graph` context manager opens with a device
            # synchronisation, a gc.collect() and an empty_cache() -- with the previous frame's 31 replays still queued that
            # is where the host waited for the GPU, and the GPU then idled through the ~4 ms of host work of this capture
            try:
                with torch.cuda.stream(side):
                    g.capture_begin(pool=pool)
                    try:
                        self.out, self.new_cache = self.net(self.x, self.t, self.cond, cache=shadow, update_cache=True,
                                                            just_2d=False)
                    finally:
                        g.capture_end()
            finally:
                DEFER_CACHE_SHIFT[0] = False
            self.graph, self.side = g, side
            # the updated cache lives in the graphs' memory pool and outlives this graph: keep the pool in use (the caching
            # allocator refuses a capture into a pool whose last graph is gone while tensors of it are alive) by letting go
            # of the previous frame's graph only now
            # ... and only once its own last replay has run (the host is a frame ahead of the GPU)
            global _pool_keeper
            if KEEP_GRAPHS:                  # (experiment, see the knob)
                _pool_keeper = _pool_keeper + [(g, torch.cuda.Event())]
            else:
                _pool_keeper = [e for e in _pool_keeper[:-1] if not e[1].query()] + _pool_keeper[-1:] + [(g, torch.cuda.Event())]
        if REPLAY_SAME_STREAM:
            # the replay goes out on the CALLER's stream, like the update kernels around it: with the capture stream as the launch
            # stream every evaluation cost two cross-stream event waits (two hardware queues: ~25 us of idle GPU per replay and
            # milliseconds at the frame boundary, profiles/r06_rollout_gaps.txt); a capture needs a side stream, a replay does not
            self.graph.replay()
        else:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.graph.replay()
            cur.wait_stream(self.side)
        _pool_keeper[-1][1].record(cur)    # (this graph is the newest entry: see the capture above)
        return self.out                    # valid until the next run(): consumed by the update kernel that follows


_t_steps_cache = {}


def _t_steps(num_steps, sigma_min, sigma_max, rho, dtype, device):
    """The rho-schedule (reference sampler.py:40-44), evaluated on the device as there; kept per argument set together with
    its host copy (the fused frame loop passes the sigmas as kernel arguments: reading them back every frame would wait for
    the previous frame's evaluations)."""
    key = (num_steps, float(sigma_min), float(sigma_max), float(rho), dtype, str(device))
    hit = _t_steps_cache.get(key)
    if hit is None:
        i = torch.arange(num_steps, dtype=dtype, device=device)
        t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
        t = torch.cat([t, torch.zeros_like(t[:1])])
        if len(_t_steps_cache) > 16:
            _t_steps_cache.clear()
        hit = _t_steps_cache[key] = (t, t.tolist())
    return hit[0]


def _fused_frame(net, graphed, cache, conditioning, t_steps, x0, B, num_steps):
    """The frame loop of edm_sampler_with_mse for the common case (no churn, no guidance, no target, fp32, CUDA): same
    arithmetic in the same order (reference sampler.py:56-76), but every evaluation is a graph replay whose x / sigma inputs
    were written in place by the previous update kernel -- per evaluation ONE replay + ONE small launch instead of ~10 torch
    launches (fill, mul, copy, clone, sub, div, mul, add ... on a 128 KB tensor, each ~5 us of a 1.6 ms evaluation)."""
    from .. import ops
    ts = next((h[1] for h in _t_steps_cache.values() if h[0] is t_steps), None) or t_steps.tolist()
    xh = x0.contiguous().clone()
    d = torch.empty_like(xh)
    graphed.prepare(xh, ts[0])
    for k in range(num_steps):
        t_hat, t_next = ts[k], ts[k + 1]
        if k == num_steps - 1:             # the evaluation whose cache update is kept (see _GraphedDenoiser.prepare)
            x_pred = graphed.run()
            ops.sampler_update(0, xh, x_pred, d, None, xh, t_hat, t_next - t_hat)
            new = graphed.finish_cache()
            cache.clear()                  # like the reference, the caller's dict itself carries the update (the replays that
            cache.update(new)              # read the old tensors are ordered before anything queued from here on)
            break
        x_pred = graphed.run()
        ops.sampler_update(0, xh, x_pred, d, None, graphed.x, t_hat, t_next - t_hat, graphed.t, t_next)
        x_pred = graphed.run()
        ops.sampler_update(1, xh, x_pred, d, graphed.x, graphed.x, t_next, t_next - t_hat)
    return xh, cache


@torch.no_grad()
def edm_sampler_with_mse(net, cache, target=None, gnet=None, conditioning=None, num_steps=32, sigma_min=0.002,
                         sigma_max=80, rho=7, guidance=1, S_churn=0, S_min=0, S_max=float("inf"), S_noise=1,
                         dtype=torch.float32, noise=None, churn_noise=None):
    """Reference signature (edm2/sampler.py:12-18) + two optional inputs for reproducible runs: `noise` (B,1,C,H,W) replaces
    the initial torch.randn draw, `churn_noise[k]` the torch.randn_like draw of step k's noise injection (S_churn > 0)."""
    was_training = net.training
    net.eval()
    B, _, C, H, W = cache.get("shape", (None,) * 5)
    device = net.device

    # per-frame-count preparation shared by all 31 evaluations of this frame (graphed or eager): RoPE tables, gate counters,
    # room in the KV rings, and the cached keys rotated ONCE for the new key count (ops.KVRing.rotate_committed)
    unet = getattr(net, "unet", None)
    if hasattr(unet, "prewarm_eval") and torch.device(device).type == "cuda":
        unet.prewarm_eval(cache)

    graphed = None
    if (SAMPLER_GRAPH and guidance == 1 and torch.device(device).type == "cuda" and dtype == torch.float32
            and num_steps >= 4):
        graphed = _GraphedDenoiser(net, cache, conditioning, B, dtype, device)

    def denoise(x, t, cache, update_cache):
        if graphed is not None and not update_cache:
            return graphed(x, t), cache
        t = torch.ones(B, 1, device=device, dtype=dtype) * t
        Dx, cache = net(x, t, conditioning, cache=cache, update_cache=update_cache, just_2d=False)
        if guidance == 1:
            return Dx, cache
        ref, _ = net(x, t, conditioning, just_2d=True)
        return ref.lerp(Dx, guidance), cache

    t_steps = _t_steps(num_steps, sigma_min, sigma_max, rho, dtype, device)
    if noise is None:
        noise = torch.randn(B, 1, C, H, W, device=device)
    x_next = noise * t_steps[0]
    mse_values, mse_pred_values = [], []
    if (graphed is not None and FUSED_FRAME and S_churn == 0 and target is None and x_next.dtype == torch.float32
            and hasattr(unet, "prewarm_eval")):
        x_next, cache = _fused_frame(net, graphed, cache, conditioning, t_steps, x_next, B, num_steps)
        if was_training:
            net.train()
        return x_next, mse_values, mse_pred_values, cache
    if target is not None:
        target = target.to(dtype)
        x_next = x_next + target
    for k, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
        x_cur = x_next
        if S_churn > 0 and S_min <= t_cur <= S_max:
            gamma = min(S_churn / num_steps, np.sqrt(2) - 1)
            t_hat = t_cur + gamma * t_cur
            x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise * (churn_noise[k] if churn_noise is not None else
                                                                          torch.randn_like(x_cur))
        else:
            t_hat, x_hat = t_cur, x_cur
        x_pred, cache = denoise(x_hat, t_hat, cache, update_cache=(k == num_steps - 1 and target is None))
        d_cur = (x_hat - x_pred) / t_hat
        x_next = x_hat + (t_next - t_hat) * d_cur
        if k < num_steps - 1:
            x_pred, _ = denoise(x_next, t_next, cache, update_cache=False)
            d_prime = (x_next - x_pred) / t_next
            x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
        if target is not None:
            mse_pred_values.append(torch.mean((x_pred - target) ** 2).item())
            mse_values.append(torch.mean((x_next - target) ** 2).item())
    if was_training:
        net.train()
    return x_next, mse_values, mse_pred_values, cache
