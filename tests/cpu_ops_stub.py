"""CPU stand-ins for the HIP ops of the training step: TEST INFRASTRUCTURE for the multi-rank host logic.

`install()` replaces the kernel-backed entry points of `autoregressive_diffusion_amd.ops` that UNet.forward reaches in
training mode by cheap differentiable torch expressions with the SAME interface, shapes and autograd topology (which tensors
and which weights every op depends on; kernel-owned weights still report `touched` and ask the WeightBank for its
end-of-backward callback) -- NOT the same arithmetic.  With it the real module tree (all 449 parameters of the gym net, its
overlap stages, its 2-D / 3-D parameter classes) runs on CPU under OnirisDDP with gloo ranks: stage-hook order, staged
exchanges, no_sync(), the active-bitmap check and the flat layout are exercised on the layout the 8-GPU run will use
(tests/test_ddp_cpu.py).  The product never imports this file; numerics are the GPU parity tests' business."""
import torch

BF16 = torch.bfloat16


def _weights_of(pw):
    return [m.param for m in pw.members] if pw.members is not None else [pw.param]


class _Owned(torch.autograd.Function):
    """Identity on y; its backward does what a conv backward does for the bookkeeping of kernel-owned weights."""

    @staticmethod
    def forward(ctx, y, pws):
        ctx.pws = pws
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        for pw in ctx.pws:
            pw.touched = pw.hit = True
            if pw.members is not None:
                for m in pw.members:
                    m.touched = m.hit = True
        ctx.pws[0].bank.request_finish()
        return g, None


def _mix(x, cout, pws, extra=()):
    """(N,H,W,cout) bf16 that depends on every element of x (per pixel), on every weight of `pws` and on `extra` tensors."""
    from autoregressive_diffusion_amd.ops import roundup
    Co = roundup(cout, 8)
    s = x.float().mean(dim=-1, keepdim=True)
    w = sum(p.float().mean() for pw in pws for p in _weights_of(pw))
    y = s * (1.0 + 0.05 * torch.tanh(w)) + 0.01 * w
    for e in extra:
        y = y + 0.01 * e.float().mean()
    ramp = torch.linspace(0.5, 1.5, Co)
    return _Owned.apply((y * ramp).to(BF16), tuple(pws))


_saved = {}


def uninstall():
    """Put the product's entry points back (the parent process of the DDP tests runs other tests afterwards)."""
    from autoregressive_diffusion_amd import ops
    for (obj, name), val in _saved.items():
        setattr(obj, name, val)
    _saved.clear()
    ops.prelude_reference = None


def install():
    from autoregressive_diffusion_amd import ops
    if _saved:
        return
    for obj, names in ((ops.WeightBank, ("prepare", "backward")),
                       (ops, ("conv", "gated_conv_train", "act", "resample", "attention_train", "FUSED_PRELUDE", "GRAD_SLOTS"))):
        for n in names:
            _saved[(obj, n)] = getattr(obj, n)

    def prepare(self, training):
        ops.GradSlot.live = []

    def backward(self):
        pass
    ops.WeightBank.prepare = prepare
    ops.WeightBank.backward = backward

    def conv(x, pw, res=None, ta=0.0, tb=0.0, clip=0.0, cscale=None, in_slot=None, res_slot=None, grad_private=False, res_alias=False):
        y = _mix(x, pw.cout, [pw])
        if res is not None:
            y = (ta * res.float() + tb * y.float()).to(BF16)
        if cscale is not None:
            y = torch.nn.functional.silu(y.float() * cscale.float()[:, None, None, :]).to(BF16)
        return y

    def gated_conv_train(x, gate, pw2, pw3, B, T, coefs=None, res=None, ta=0.0, tb=0.0, clip=0.0, cscale=None, grad_private=False, res_slot=None, res_alias=False):
        ca, cb = coefs if coefs is not None else ops.gate_coefs(gate)
        N = x.shape[0]
        y = _mix(x, pw2.cout, [pw2, pw3]).float() * (ca.float() + cb.float()).reshape(N, 1, 1, 1)
        if res is not None:
            y = ta * res.float() + tb * y
        if cscale is not None:
            y = torch.nn.functional.silu(y * cscale.float()[:, None, None, :])
        return y.to(BF16)

    def act(x, skip=None, w1=1.0, w2=1.0, norm=False, want_xo=False, in_slot=None, skip_slot=None, resample="keep", xo_slot=None):
        x = ops_resample(x, resample)
        v = torch.cat([w1 * x.float(), w2 * skip.float()], dim=-1) if skip is not None else x.float()
        if norm:
            v = v / (1e-4 + v.norm(dim=-1, keepdim=True) / v.shape[-1] ** 0.5)
        a = (torch.nn.functional.silu(v) / 0.596).to(BF16)
        return (v.to(BF16), a) if (want_xo or norm) else a

    def ops_resample(x, mode, in_slot=None):
        if mode == "keep":
            return x
        N, H, W, C = x.shape
        if mode == "down":
            return x.reshape(N, H // 2, 2, W // 2, 2, C).float().mean(dim=(2, 4)).to(x.dtype)
        return x[:, :, None, :, None, :].expand(N, H, 2, W, 2, C).reshape(N, 2 * H, 2 * W, C)

    def attention_train(qkv, kind, B, T, heads, rope_bufs=None):
        N, P, C3 = qkv.shape
        q, k, v = qkv.float().split(C3 // 3, dim=-1)
        return (v + 0.1 * torch.tanh(q * k).mean(dim=1, keepdim=True)).to(BF16)

    ops.conv, ops.gated_conv_train, ops.act, ops.resample, ops.attention_train = conv, gated_conv_train, act, ops_resample, attention_train
    ops.FUSED_PRELUDE = 0
    ops.GRAD_SLOTS = 0
    import torch_prelude
    ops.prelude_reference = torch_prelude
