"""fp32 verification path: `Precond(use_fp16=False)` / `Precond.forward(force_fp32=True)` (reference networks_edm2.py:285,294 --
the switch that picks the arithmetic type of the whole net; the reference's modules then compute in the dtype of their input,
conv.py:37-46).

What runs here: the SAME modules and parameters as the bf16 product path, evaluated in fp32 end to end.  Every contraction --
1x1 / 3x3 convolutions with their data and weight gradients, attention scores / values with their gradients -- is a HIP
kernel on fp32 operands (csrc/fp32.hip: v_mfma_f32_32x32x2_f32, fp32 accumulation; include/oniris.h `oniris_conv_f32`,
`oniris_wgrad_f32`, `oniris_attn_f32_fwd/_bwd`).  The per-element glue between them (magnitude-preserving sums, SiLU, pixel
norms, gates, rotary rotation) is written with torch's fp32 elementwise ops under autograd: this is a VERIFICATION mode -- it
exists so that the reference's own criterion std(diff) <= 3e-4 (edm2/consistency_test.py:23-32) can be held against the
reference-generated fp32 fixtures (tests/test_fp32_gpu.py) -- not the timed path, and bench.py never enters it.
Like the product path it has no CPU form: the kernels refuse host tensors.

Also served from here: attention heads WIDER than 64 channels in the bf16 path (networks_edm2.py:28,39 accepts any
`channels_per_head`; the product kernels are written for <= 64): ops.attention_* hands such layers to `attention()` below.

`fp32_arithmetic()` is the switch for code that calls the modules directly (tests; the reference's module-level API has no
precision argument: there the input dtype decides)."""
import contextlib
import ctypes
import math
import threading

import torch

from . import _lib
from ._lib import lib

_tls = threading.local()


def active():
    return getattr(_tls, "depth", 0) > 0


@contextlib.contextmanager
def fp32_arithmetic():
    """Inside this block every module of the package (MPConv, MPCausal3DGatedConv, Video/FrameAttention, Block, UNet, Precond)
    computes in fp32 on fp32 activations.  Caches written inside it are fp32 (NCHW) and only valid inside it."""
    _tls.depth = getattr(_tls, "depth", 0) + 1
    try:
        yield
    finally:
        _tls.depth -= 1


def _check(rc, who):
    if rc != 0:
        raise RuntimeError(f"{who}: {lib.oniris_last_error().decode()}")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the fp32 path runs on HIP kernels like the product path: tensors must live on the MI355X "
                               "(there is no CPU fallback)")


# ----------------------------------------------------------------------------------------------------------------------
# contractions

class _ConvF32(torch.autograd.Function):
    """x (N, H, W, Cin) fp32, w (taps, Cout, Cin) fp32 -> (N, H, W, Cout); taps 1 or 9 (3x3, zero padding)."""

    @staticmethod
    def forward(ctx, x, w):
        _need_gpu(x, w)
        x, w = x.contiguous(), w.contiguous()
        N, H, W, Cin = x.shape
        taps, Cout, _ = w.shape
        out = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device)
        _check(lib.oniris_conv_f32(_p(x), _p(w), _p(out), N, H, W, Cin, Cout, taps, _stream()), "conv_f32")
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        N, H, W, Cin = x.shape
        taps, Cout, _ = w.shape
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = w.flip(0).transpose(1, 2).contiguous()              # [8 - tap][ci][co]: the adjoint convolution
            dx = torch.empty_like(x)
            _check(lib.oniris_conv_f32(_p(g), _p(wt), _p(dx), N, H, W, Cout, Cin, taps, _stream()), "conv_f32 (dgrad)")
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            _check(lib.oniris_wgrad_f32(_p(x), _p(g), _p(dw), N, H, W, Cin, Cout, taps, _stream()), "wgrad_f32")
        return dx, dw


def conv2d(x, w):
    """F.conv2d(x, w, padding=k // 2) of MPConv.forward (conv.py:41-46) on NCHW fp32: w (Cout, Cin, k, k), k = 1 or 3."""
    Cout, Cin, kh, kw = w.shape
    if (kh, kw) not in ((1, 1), (3, 3)):
        raise NotImplementedError(f"fp32 path: {kh}x{kw} convolution (the reference's nets use 1x1 and 3x3)")
    wp = w.reshape(Cout, Cin, kh * kw).permute(2, 0, 1)
    y = _ConvF32.apply(x.permute(0, 2, 3, 1), wp)
    return y.permute(0, 3, 1, 2)


def linear(x, w):
    """x @ w.t() (conv.py:38-39) through the same kernel: a 1x1 convolution over one position per row."""
    y = _ConvF32.apply(x[:, None, None, :], w[None])
    return y[:, 0, 0, :]


class _AttnF32(torch.autograd.Function):
    """q (BH, Lq, D), k / v (BH, Lk, D) fp32 -> softmax(scale q k^T + mask) v; masks: include/oniris.h OnirisAttnF32Args."""

    @staticmethod
    def forward(ctx, q, k, v, mask_mode, P, T, q_frame_off, scale):
        _need_gpu(q, k, v)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        BH, Lq, D = q.shape
        Lk = k.shape[1]
        out = torch.empty_like(q)
        lse = torch.empty((BH, Lq), dtype=torch.float32, device=q.device)
        a = _lib.AttnF32Args()
        a.q, a.k, a.v, a.out, a.lse = _p(q), _p(k), _p(v), _p(out), _p(lse)
        a.BH, a.Lq, a.Lk, a.D, a.mask_mode, a.P, a.T, a.q_frame_off, a.scale = BH, Lq, Lk, D, mask_mode, P, T, q_frame_off, scale
        _check(lib.oniris_attn_f32_fwd(ctypes.byref(a), _stream()), "attn_f32_fwd")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.meta = (mask_mode, P, T, q_frame_off, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        mask_mode, P, T, q_frame_off, scale = ctx.meta
        g = g.contiguous()
        delta = (g * out).sum(-1).contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        a = _lib.AttnF32Args()
        a.q, a.k, a.v, a.out, a.lse = _p(q), _p(k), _p(v), _p(out), _p(lse)
        a.dout, a.delta, a.dq, a.dk, a.dv = _p(g), _p(delta), _p(dq), _p(dk), _p(dv)
        a.BH, a.Lq, a.Lk, a.D = q.shape[0], q.shape[1], k.shape[1], q.shape[2]
        a.mask_mode, a.P, a.T, a.q_frame_off, a.scale = mask_mode, P, T, q_frame_off, scale
        _check(lib.oniris_attn_f32_bwd(ctypes.byref(a), _stream()), "attn_f32_bwd")
        return dq, dk, dv, None, None, None, None, None


def attention(q, k, v, mask="dense", P=1, T=1, q_frame_off=0):
    """q (B, m, Lq, d), k / v (B, m, Lk, d) -> (B, m, Lq, d).  mask: 'dense' (F.scaled_dot_product_attention,
    attention_modules.py:42,70,115), 'causal' (frames of P tokens, key frame <= query frame + q_frame_off: make_infer_mask,
    attention_masking.py:64-90), 'train' (make_train_mask as the compiled FlexAttention evaluates it, :27-53)."""
    B, m, Lq, d = q.shape
    mode = {"dense": 0, "causal": 1, "train": 2}[mask]
    out = _AttnF32.apply(q.reshape(B * m, Lq, d).float(), k.reshape(B * m, -1, d).float(), v.reshape(B * m, -1, d).float(),
                         mode, int(P), int(T), int(q_frame_off), 1.0 / math.sqrt(d))
    return out.reshape(B, m, Lq, d)


def rope(q, k, inv_freq, scale_vec, training, scale_base=64):
    """RotaryEmbedding.forward (RoPe.py:21-57) from its two buffers: q, k (b, m, frames, hw, c) -> rotated (b, m, frames * hw, c).
    Angles and xPos scales are rounded to fp16 as in the reference (tables of fp16 values, fp16 cos / sin)."""
    nk = k.shape[-3] // 2 if training else k.shape[-3]
    t = torch.arange(nk, device=inv_freq.device).type_as(inv_freq)
    ang = torch.outer(t, inv_freq)
    ang = torch.cat((ang, ang), dim=-1).to(torch.float16)
    sc = scale_vec[None, :] ** ((t - (nk // 2)) / scale_base)[:, None]
    sc = torch.cat((sc, sc), dim=-1).to(torch.float16).unsqueeze(1)
    cos, sin = ang.cos().unsqueeze(1), ang.sin().unsqueeze(1)
    if training:
        cos, sin, sc = (torch.cat((z, z), dim=0) for z in (cos, sin, sc))

    def rot(x):
        a, b = x.chunk(2, dim=-1)
        return torch.cat((-b, a), dim=-1)
    k = (k * cos + rot(k) * sin) / sc
    nq = q.shape[-3]
    q = (q * cos[-nq:] + rot(q) * sin[-nq:]) * sc[-nq:]
    return q.flatten(-3, -2), k.flatten(-3, -2)


def wide_heads_train(qkv, kind, B, T, heads, rope_bufs):
    """The attention core of the bf16 path for heads WIDER than 64 channels (the product kernels are written for <= 64;
    networks_edm2.py:28,39 accepts any channels_per_head): qkv (N, P, 3C) bf16 with channel = s * C + head * d + c (the packed
    attn_qkv order) -> (N, P, C) bf16.  Per-head normalisation and rotation as fp32 torch ops under autograd, scores / values /
    their gradients through the generic fp32 attention kernel (any width up to 256) -- correct and slow; no BASELINE
    configuration comes here."""
    u = _utils()
    N, P, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    x = qkv.float().reshape(N, P, 3, heads, d)
    q, k, v = (u.normalize(x[:, :, s], dim=-1) for s in range(3))               # (N, P, m, d)
    if kind == "video":
        q, k, v = (z.reshape(B, 2 * T, P, heads, d).permute(0, 3, 1, 2, 4) for z in (q, k, v))     # (b, m, 2T, P, d)
        qr, kr = rope(q, k, rope_bufs[0], rope_bufs[1], True)
        o = attention(qr, kr, v.reshape(B, heads, 2 * T * P, d), "train", P=P, T=T)
        o = o.reshape(B, heads, 2 * T, P, d).permute(0, 2, 3, 1, 4).reshape(N, P, C)
    else:
        q, k, v = (z.permute(0, 2, 1, 3) for z in (q, k, v))                    # (N, m, P, d)
        o = attention(q, k, v, "dense").permute(0, 2, 1, 3).reshape(N, P, C)
    return o.to(qkv.dtype)


@torch.no_grad()
def wide_heads_eval(qkv, B, heads, rope_bufs, kv_cache, update_cache, P):
    """Eval-mode VideoAttention core for wide heads (attention_modules.py:51-77): the cache is the reference's -- normalised,
    UN-rotated k and v (b, m, frames, P, d), here fp32 --, every call rotates all keys for the grown key count."""
    u = _utils()
    N, _, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    t = N // B
    x = qkv.float().reshape(N, P, 3, heads, d)
    q, k, v = (u.normalize(x[:, :, s], dim=-1).reshape(B, t, P, heads, d).permute(0, 3, 1, 2, 4) for s in range(3))
    if kv_cache is not None:
        k, v = torch.cat((kv_cache[0], k), dim=2), torch.cat((kv_cache[1], v), dim=2)
    new_cache = (k, v) if update_cache else kv_cache
    qr, kr = rope(q, k, rope_bufs[0], rope_bufs[1], False)
    vf = v.reshape(B, heads, -1, d)
    if t == 1:
        o = attention(qr, kr, vf, "dense")
    else:
        o = attention(qr, kr, vf, "causal", P=P, q_frame_off=k.shape[2] - t)
    return o.reshape(B, heads, t, P, d).permute(0, 2, 3, 1, 4).reshape(N, P, C).to(qkv.dtype), new_cache


# ----------------------------------------------------------------------------------------------------------------------
# per-element glue (the reference's edm2/utils.py formulas on NCHW fp32)

def _utils():
    from .edm2 import utils
    return utils


def resample(x, f, mode):
    """utils.py:94-107 (depthwise strided / transposed filter pass with the separable filter f / sum(f)) as shifted slices."""
    if mode == "keep":
        return x
    taps = [float(v) for v in f]
    L = len(taps)
    if L % 2 != 0 or L < 2:
        raise ValueError("resample: the filter needs an even number of taps")
    s = sum(taps)
    taps = [t / s for t in taps]
    pad = (L - 1) // 2
    N, C, H, W = x.shape
    if mode == "down":
        xp = torch.nn.functional.pad(x, (pad, pad, pad, pad))
        Ho, Wo = (H + 2 * pad - L) // 2 + 1, (W + 2 * pad - L) // 2 + 1
        out = None
        for i, fi in enumerate(taps):
            for j, fj in enumerate(taps):
                term = xp[:, :, i:i + 2 * Ho - 1:2, j:j + 2 * Wo - 1:2] * (fi * fj)
                out = term if out is None else out + term
        return out
    assert mode == "up"
    # conv_transpose2d(stride 2, padding pad, kernel 4 f f^T): zero-stuffed input correlated with the flipped kernel
    z = x.new_zeros(N, C, 2 * H - 1, 2 * W - 1)
    z[:, :, ::2, ::2] = x
    lo, Ho, Wo = L - 1 - pad, 2 * H - 2 - 2 * pad + L, 2 * W - 2 - 2 * pad + L
    zp = torch.nn.functional.pad(z, (lo, Wo + L - 1 - lo - (2 * W - 1), lo, Ho + L - 1 - lo - (2 * H - 1)))
    out = None
    for i in range(L):
        for j in range(L):
            term = zp[:, :, i:i + Ho, j:j + Wo] * (4.0 * taps[L - 1 - i] * taps[L - 1 - j])
            out = term if out is None else out + term
    return out


# ----------------------------------------------------------------------------------------------------------------------
# modules (reference file:line in each docstring); x is NCHW fp32 as in the reference's public signatures

def mpconv(mod, x, gain=1):
    """MPConv.forward (conv.py:34-46): the weight is force-normalised in place in training mode, normalised again with
    gradient, scaled by gain / sqrt(fan_in)."""
    w = mod.weight(gain)                          # NormalizedWeight.forward: conv.py:14-21 (plain torch fp32)
    x = x.float()
    if w.ndim == 2:
        return linear(x, w)
    return conv2d(x, w)


def gated_conv(mod, x, emb, batch_size, c_noise, cache=None, update_cache=False, just_2d=False):
    """MPCausal3DGatedConv.forward (conv.py:59-95).  The (2,3,3) temporal kernel over [frame t-2, frame t-1] is two 3x3
    convolutions summed; temporal padding is ONES (:65), spatial padding zeros."""
    u = _utils()
    if just_2d:
        return mpconv(mod.last_frame_conv, x), cache
    if cache is None:
        cache = {}
    x = x.float()
    w = mod.weight()                              # (Cout, Cin, 2, 3, 3)
    B = batch_size
    N, C, H, W = x.shape
    pad = cache.get("activations")
    if pad is None:
        pad = torch.ones(B, C, 2, H, W, device=x.device, dtype=x.dtype)
    gate, n_new = mod.gating(c_noise.float(), cache.get("n_context_frames", 0))
    if update_cache:
        cache["n_context_frames"] = n_new
    y2 = mpconv(mod.last_frame_conv, x)
    if mod.training:
        T = N // (2 * B)
        clean = x.reshape(B, 2, T, C, H, W)[:, 0]                        # '(b s t) c h w': the context is the clean half
    else:
        T = N // B
        clean = x.reshape(B, T, C, H, W)
    ctx = torch.cat([pad.permute(0, 2, 1, 3, 4), clean], dim=1)         # (B, 2 + T, C, H, W)
    if update_cache:
        cache["activations"] = ctx[:, -2:].permute(0, 2, 1, 3, 4).clone().detach()
    y3 = (conv2d(ctx[:, 0:T].reshape(B * T, C, H, W), w[:, :, 0]) + conv2d(ctx[:, 1:T + 1].reshape(B * T, C, H, W), w[:, :, 1]))
    if mod.training:
        y3 = y3.reshape(B, 1, T, -1, H, W).expand(B, 2, T, y3.shape[1], H, W).reshape(N, -1, H, W)
    return u.mp_sum(y2, y3, gate.flatten()), cache


def _split_heads(y, m):
    """'n (m c s) h w -> s n m (h w) c' + normalize(dim=-1) (attention_modules.py:37-38,48-49)."""
    u = _utils()
    N, C3, H, W = y.shape
    d = C3 // (3 * m)
    y = y.reshape(N, m, d, 3, H * W).permute(3, 0, 1, 4, 2)            # (s, n, m, hw, c)
    return u.normalize(y, dim=-1).unbind(0)


def frame_attention(mod, x):
    """FrameAttention.forward (attention_modules.py:105-119) / VideoAttention's just_2d branch (:36-45)."""
    u = _utils()
    if mod.num_heads == 0:
        return x
    N, C, H, W = x.shape
    y = mpconv(mod.attn_qkv, x)
    q, k, v = _split_heads(y, mod.num_heads)
    o = attention(q, k, v, "dense")                                    # (n, m, hw, c)
    o = o.permute(0, 1, 3, 2).reshape(N, C, H, W)
    return u.mp_sum(x.float(), mpconv(mod.attn_proj, o), t=mod.attn_balance)


def video_attention(mod, x, batch_size, cache=None, update_cache=False, just_2d=False):
    """VideoAttention.forward (attention_modules.py:30-82): rotary embedding over the frame index (RoPe.py), the DART training
    mask / causal prefill / one new frame against the cache.  The cache holds normalised UN-rotated k and v (:51-57)."""
    u = _utils()
    if mod.num_heads == 0:
        return x, None
    if just_2d:
        return frame_attention(mod, x), cache
    N, C, H, W = x.shape
    B, m, P = batch_size, mod.num_heads, H * W
    t = N // B
    y = mpconv(mod.attn_qkv, x)
    q, k, v = _split_heads(y, m)                                        # (n, m, hw, c)
    q, k, v = (z.reshape(B, t, m, P, -1).permute(0, 2, 1, 3, 4) for z in (q, k, v))      # (b, m, t, hw, c)
    if not mod.training:
        if cache is not None:
            k, v = torch.cat((cache[0], k), dim=2), torch.cat((cache[1], v), dim=2)
        if update_cache:
            cache = (k, v)
    rope = mod.rope
    was = rope.training
    rope.train(mod.training)                                           # (the rotary module reads ITS OWN flag for the clean | noised layout)
    qr, kr = rope(q, k)                                                # RoPe.py:43-68: (b, m, frames * hw, c)
    rope.train(was)
    vf = v.reshape(B, m, -1, v.shape[-1])
    if mod.training:
        o = attention(qr, kr, vf, "train", P=P, T=t // 2)
    elif t == 1:
        o = attention(qr, kr, vf, "dense")
    else:
        o = attention(qr, kr, vf, "causal", P=P, q_frame_off=k.shape[2] - t)
    o = o.reshape(B, m, t, H, W, -1).permute(0, 2, 1, 5, 3, 4).reshape(N, C, H, W)       # 'b m (t h w) c -> (b t) (m c) h w'
    return u.mp_sum(x.float(), mpconv(mod.attn_proj, o), t=mod.attn_balance), cache


def block(mod, x, emb, batch_size, c_noise, cache=None, update_cache=False, just_2d=False):
    """Block.forward (networks_edm2.py:62-94)."""
    u = _utils()
    if cache is None:
        cache = {}
    x = resample(x.float(), mod.resample_filter, mod.resample_mode)
    if mod.flavor == "enc":
        if mod.conv_skip is not None:
            x = mpconv(mod.conv_skip, x)
        x = u.normalize(x, dim=1)
    y, cache["conv_res0"] = gated_conv(mod.conv_res0, u.mp_silu(x), emb, batch_size, c_noise, cache.get("conv_res0"), update_cache, just_2d)
    c = mpconv(mod.emb_linear, emb, gain=mod.emb_gain) + 1
    y = u.mp_silu(u.bmult(y, c))
    if mod.training and mod.dropout != 0:
        y = torch.nn.functional.dropout(y, p=mod.dropout)
    y, cache["conv_res1"] = gated_conv(mod.conv_res1, y, emb, batch_size, c_noise, cache.get("conv_res1"), update_cache, just_2d)
    if mod.flavor == "dec" and mod.conv_skip is not None:
        x = mpconv(mod.conv_skip, x)
    x = u.mp_sum(x, y, t=mod.res_balance)
    from .edm2.attention import VideoAttention
    if isinstance(mod.attn, VideoAttention):
        x, cache["attn"] = video_attention(mod.attn, x, batch_size, cache.get("attn"), update_cache, just_2d)
    else:
        x, cache["attn"] = frame_attention(mod.attn, x), None
    if mod.clip_act is not None:
        x = x.clip(-mod.clip_act, mod.clip_act)
    return x, cache


def unet(mod, x, c_noise, conditioning=None, cache=None, update_cache=False, just_2d=False):
    """UNet.forward (networks_edm2.py:191-236)."""
    u = _utils()
    if cache is None:
        cache = {}
    B, tt = x.shape[:2]
    n_ctx = cache.get("n_context_frames", 0)
    _, n_new = mod.out_res(c_noise.float(), n_ctx, just_2d)
    if update_cache:
        cache["n_context_frames"] = n_new
    x = x.float().reshape(B * tt, *x.shape[2:])
    cn = c_noise.float().reshape(-1)
    labels = torch.arange(tt, device=x.device).repeat(B) + n_ctx
    labels = labels.log1p().to(cn.dtype) / 4
    mpconv(mod.emb_time, mod.emb_fourier_time(labels))                 # (:206: evaluated and unused -- its forced weight-norm still happens)
    emb = mpconv(mod.emb_noise, mod.emb_fourier_sigma(cn))
    if mod.emb_label is not None and conditioning is not None:
        onehot = torch.nn.functional.one_hot(conditioning.reshape(-1), num_classes=mod.label_dim).to(cn.dtype) * mod.label_dim ** 0.5
        emb = u.mp_sum(emb, mpconv(mod.emb_label, onehot), t=1 / 3)
    emb = u.mp_silu(emb)
    c_noise = cn.reshape(B, tt)
    x = torch.cat([x, torch.ones_like(x[:, :1])], dim=1)
    from .edm2.networks_edm2 import Block
    skips = []
    for name, blk in mod.enc.items():
        if isinstance(blk, Block):
            x, cache["enc", name] = block(blk, x, emb, B, c_noise, cache.get(("enc", name)), update_cache, just_2d)
        else:
            x, cache["enc", name] = gated_conv(blk, x, emb, B, c_noise, cache.get(("enc", name)), update_cache, just_2d)
        skips.append(x)
    for name, blk in mod.dec.items():
        if "block" in name:
            x = u.mp_cat(x, skips.pop(), t=mod.concat_balance)
        x, cache["dec", name] = block(blk, x, emb, B, c_noise, cache.get(("dec", name)), update_cache, just_2d)
    x, cache["out_conv"] = gated_conv(mod.out_conv, x, emb, B, c_noise, cache.get("out_conv"), update_cache, just_2d)
    return x.reshape(B, tt, *x.shape[1:]) * mod.out_gain, cache
