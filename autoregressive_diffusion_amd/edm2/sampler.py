"""EDM / Heun sampler that generates ONE next frame against a cache (reference edm2/sampler.py:12-85):
rho-schedule, optional churn, Euler + 2nd-order correction, cache updated on the last Euler evaluation only."""
import os
import numpy as np
import torch


SAMPLER_GRAPH = int(os.environ.get("ONIRIS_SAMPLER_GRAPH", "1"))
_graph_pool = None


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


@torch.no_grad()
def edm_sampler_with_mse(net, cache, target=None, gnet=None, conditioning=None, num_steps=32, sigma_min=0.002,
                         sigma_max=80, rho=7, guidance=1, S_churn=0, S_min=0, S_max=float("inf"), S_noise=1,
                         dtype=torch.float32, noise=None):
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

    i = torch.arange(num_steps, dtype=dtype, device=device)
    t_steps = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    t_steps = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])
    if noise is None:
        noise = torch.randn(B, 1, C, H, W, device=device)
    x_next = noise * t_steps[0]
    mse_values, mse_pred_values = [], []
    if target is not None:
        target = target.to(dtype)
        x_next = x_next + target
    for k, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
        x_cur = x_next
        if S_churn > 0 and S_min <= t_cur <= S_max:
            gamma = min(S_churn / num_steps, np.sqrt(2) - 1)
            t_hat = t_cur + gamma * t_cur
            x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise * torch.randn_like(x_cur)
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
