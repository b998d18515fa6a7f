"""Optimizer side of the training step (gym_train.py:105-108): gradient-norm clipping + AdamW + power-function EMA.
CPU: the flat-buffer host logic against torch's own clip_grad_norm_ / AdamW / lerp_ and the reference's EMA
coefficients (fixture G11); GPU: the fused HIP pass against the same host math."""
import os
import numpy as np
import pytest
import torch
from torch import nn

import cpu_reference_optimizer      # (the product's FlatAdamW has no CPU arithmetic: the CPU tests install their own)
cpu_reference_optimizer.install()

HERE = os.path.dirname(os.path.abspath(__file__))


def test_power_function_coefficients_match_reference():
    from autoregressive_diffusion_amd.parallel import power_function_exponent, power_function_beta
    z = np.load(os.path.join(HERE, "golden", "g11_phema.npz"))
    for i, s in enumerate(z["stds"]):
        assert abs(power_function_exponent(s) - z["exps"][i]) <= 1e-9 * z["exps"][i]
        for j, (a, b) in enumerate(zip(z["t_next"], z["t_delta"])):
            assert abs(power_function_beta(s, a, b) - z["betas"][i, j]) <= 1e-12
    with pytest.raises(ValueError):
        power_function_exponent(0.5)


def _nets():
    torch.manual_seed(3)
    a = nn.Sequential(nn.Linear(7, 9), nn.Tanh(), nn.Linear(9, 5))
    b = nn.Sequential(nn.Linear(7, 9), nn.Tanh(), nn.Linear(9, 5))
    b.load_state_dict(a.state_dict())
    return a, b


def test_flat_clip_adamw_ema_matches_torch():
    """FlatAdamW.step(max_norm, ema) == clip_grad_norm_ + torch.optim.AdamW.step + PowerFunctionEMA.update."""
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW, FlatEMA, power_function_beta
    import copy
    a, b = _nets()
    flat = FlatParams(a)
    opt_a = FlatAdamW(flat, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    ema_a = FlatEMA(flat, stds=(0.05, 0.10))
    opt_b = torch.optim.AdamW(b.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    emas_b = [copy.deepcopy(b), copy.deepcopy(b)]
    g = torch.Generator().manual_seed(4)
    bs = 8
    for i in range(1, 6):
        x = torch.randn(16, 7, generator=g) * 3
        for net in (a, b):
            net.zero_grad(set_to_none=False) if net is b else opt_a.zero_grad()
            net(x).pow(2).sum().backward()
        total = torch.nn.utils.clip_grad_norm_(b.parameters(), 0.1)
        assert float(total) > 0.1                       # the clip is active in this test
        opt_b.step()
        for std, e in zip((0.05, 0.10), emas_b):
            beta = power_function_beta(std, i * bs, bs)
            for pn, pe in zip(b.parameters(), e.parameters()):
                pe.data.lerp_(pn.data, 1 - beta)
        opt_a.step(max_norm=0.1, ema=ema_a.weights(i * bs, bs))
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, atol=1e-6), (pa - pb).abs().max()
    for k, e in enumerate(emas_b):
        for pa, pe in zip(a.parameters(), e.parameters()):
            assert torch.allclose(ema_a.view(k, pa), pe, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,clip,nema", [(1000003, True, 2), (4096, False, 1), (777, True, 0), (50000, False, 0)])
def test_fused_optimizer_kernel(n, clip, nema):
    from autoregressive_diffusion_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01
    m, v = torch.randn(n, generator=g) * 0.01, torch.rand(n, generator=g) * 1e-4
    emas = [torch.randn(n, generator=g) for _ in range(nema)]
    ws = [0.13, 0.06][:nema]
    lr, b1, b2, eps, wd, step, gs = 1e-2, 0.9, 0.99, 1e-8, 0.01, 3, 0.5
    # host math
    G = gr.double() * gs
    if clip:
        G = G * min(1.0, 0.1 / (float(G.norm()) + 1e-6))
    M = b1 * m.double() + (1 - b1) * G
    V = b2 * v.double() + (1 - b2) * G * G
    P = p.double() * (1 - lr * wd) - lr * (M / (1 - b1 ** step)) / ((V / (1 - b2 ** step)).sqrt() + eps)
    E = [e.double() + w * (P - e.double()) for e, w in zip(emas, ws)]
    # device
    dp, dg, dm, dv = (t.to(dev) for t in (p, gr, m, v))
    de = [e.to(dev) for e in emas]
    buf = torch.zeros(1 + ops.SQNORM_WS, device=dev)
    ops.adamw_(dp, dg, dm, dv, lr, b1, b2, eps, wd, step, gs, 0.1 if clip else None, buf, list(zip(de, ws)))
    torch.cuda.synchronize()
    if clip:
        assert abs(float(buf[0]) - float((gr.double() ** 2).sum())) <= 1e-5 * float((gr.double() ** 2).sum())
    rel = lambda a, b: float((a.double().cpu() - b).norm() / (b.norm() + 1e-30))
    assert rel(dp, P) < 1e-6 and rel(dm, M) < 1e-6 and rel(dv, V) < 1e-5
    for a, b in zip(de, E):
        assert rel(a, b) < 1e-6
