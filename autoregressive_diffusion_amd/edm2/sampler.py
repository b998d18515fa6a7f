"""EDM / Heun sampler that generates ONE next frame against a cache (reference edm2/sampler.py:12-85):
rho-schedule, optional churn, Euler + 2nd-order correction, cache updated on the last Euler evaluation only."""
import numpy as np
import torch


@torch.no_grad()
def edm_sampler_with_mse(net, cache, target=None, gnet=None, conditioning=None, num_steps=32, sigma_min=0.002,
                         sigma_max=80, rho=7, guidance=1, S_churn=0, S_min=0, S_max=float("inf"), S_noise=1,
                         dtype=torch.float32, noise=None):
    was_training = net.training
    net.eval()
    B, _, C, H, W = cache.get("shape", (None,) * 5)
    device = net.device

    def denoise(x, t, cache, update_cache):
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
