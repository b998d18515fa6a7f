"""Plain-torch AdamW / clip / EMA update for FlatAdamW on CPU tensors: TEST INFRASTRUCTURE (the product path has no CPU
arithmetic: parallel.FlatAdamW refuses CPU tensors unless this is installed).  The world-size-2 gloo tests exercise the
host-side logic of the data-parallel classes with it.  install() is idempotent; spawned workers call it themselves."""
import torch


def cpu_update(opt, runs, grad_scale, max_norm, ema, owned, norm_reduce):
    f = opt.flat
    grad = getattr(opt, 'grad_src', f.grad)      # (mesh exchange: the reduced chunks live in f.grad_reduced)
    b1, b2 = opt.betas
    coef = 1.0
    if max_norm is not None:
        if owned is None:
            sq = (grad * grad_scale).pow(2).sum().reshape(1)
        else:
            sq = norm_reduce(sum((grad[a:b] * grad_scale).pow(2).sum() for a, b in owned).reshape(1))
        coef = min(1.0, max_norm / (float(sq.sqrt()) + 1e-6))
    for lo, hi, st in runs:
        if st:
            g = grad[lo:hi] * (grad_scale * coef)
            opt.m[lo:hi].mul_(b1).add_(g, alpha=1 - b1)
            opt.v[lo:hi].mul_(b2).addcmul_(g, g, value=1 - b2)
            mh, vh = opt.m[lo:hi] / (1 - b1 ** st), opt.v[lo:hi] / (1 - b2 ** st)
            f.flat[lo:hi].mul_(1 - opt.lr * opt.weight_decay).sub_(opt.lr * mh / (vh.sqrt() + opt.eps))
        for e, w in ema:
            e[lo:hi].lerp_(f.flat[lo:hi], w)


def install():
    from autoregressive_diffusion_amd.parallel import FlatAdamW
    FlatAdamW.cpu_update = staticmethod(cpu_update)
