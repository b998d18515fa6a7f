"""CPU ORACLE for the Oniris denoiser step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-PyTorch fp32 restatement (written from scratch, in the fused algebraic form the HIP kernels use) of the
reference hot path of Francesco215/autoregressive_diffusion.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this file; the product package `autoregressive_diffusion_amd`
never does (it fails loudly when the HIP library is missing).

Parity pin: every function here is checked against golden vectors produced by importing the real reference in
the build container (tests/golden/make_golden.py -> tests/golden/*.npz; test: tests/test_oracle_golden.py).

Reference citations (relative to the reference repository root):
  normalize / mp_silu / mp_sum / mp_cat / resample / MPFourier ... edm2/utils.py:83-158
  NormalizedWeight / MPConv / MPCausal3DGatedConv / Gating ......... edm2/conv.py:8-127
  RotaryEmbedding .................................................. edm2/attention/RoPe.py:5-74
  make_train_mask / make_infer_mask ................................ edm2/attention/attention_masking.py:8-90
  VideoAttention / FrameAttention .................................. edm2/attention/attention_modules.py:15-119
  Block / UNet / Precond ........................................... edm2/networks_edm2.py:19-297
  EDM2Loss ......................................................... edm2/loss.py:9-47
  edm_sampler_with_mse ............................................. edm2/sampler.py:12-85
All state is passed explicitly: `params` is a dict with the reference's state_dict key names.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-4
SILU_DIV = 0.596
FLEX_BLOCK = 128  # torch.nn.attention.flex_attention._DEFAULT_SPARSE_BLOCK_SIZE


# ----------------------------------------------------------------------------------------------------------------
# mask tables (edm2/attention/attention_masking.py:27-53, 64-90)  -- integer work, bit-exact target

def train_table(n_frames, image_size):
    """kv_num_blocks (2nb,), kv_indices (2nb,2nb) int32 of make_train_mask for one (batch, head); None if the
    reference returns None (T*P not a multiple of 128 while P < 128).  Also returns the block size."""
    T, P = int(n_frames), int(image_size)
    if P < FLEX_BLOCK:
        if (T * P) % FLEX_BLOCK != 0:
            return None
        nb, blk = T * P // FLEX_BLOCK, FLEX_BLOCK
    else:
        nb, blk = T, P
    num = np.tile(np.arange(1, nb + 1, dtype=np.int32), 2)
    idx = np.zeros((2 * nb, 2 * nb), dtype=np.int32)
    for i in range(nb):
        idx[i, : i + 1] = np.arange(i + 1)          # clean row i   -> clean blocks 0..i
        idx[nb + i, :i] = np.arange(i)              # noisy row i   -> clean blocks 0..i-1
        idx[nb + i, i] = nb + i                     #                 + its own noisy block
    return num, idx, blk


def infer_table(n_frames, image_size):
    """make_infer_mask: returns (kind, num, idx, blk).  kind: 'score_mod' (t*P < 128: pure frame-causal mask),
    'dense' (t*P not a multiple of 128: create_block_mask on the frame-causal mask_mod), 'table'."""
    t, P = int(n_frames), int(image_size)
    if t * P < FLEX_BLOCK:
        return "score_mod", None, None, None
    if P < FLEX_BLOCK:
        if (t * P) % FLEX_BLOCK != 0:
            return "dense", None, None, None
        nb, blk = t * P // FLEX_BLOCK, FLEX_BLOCK
    else:
        nb, blk = t, P
    num = np.arange(1, nb + 1, dtype=np.int32)
    idx = np.zeros((nb, nb), dtype=np.int32)
    for i in range(nb):
        idx[i, : i + 1] = np.arange(i + 1)
    return "table", num, idx, blk


def train_mask_mod(qf, kf, T):
    """TrainingMask.__call__ on FRAME indices in [0,2T) (attention_masking.py:15-24), simplified:
    clean q sees clean kf<=qf; noisy q (frame f=qf-T) sees clean kf<f and itself."""
    clean_q = qf < T
    return np.where(clean_q, (kf < T) & (kf <= qf), ((kf < T) & (kf < qf - T)) | (kf == qf))


def train_allowed_tokens(T, P):
    """Dense boolean (2TP,2TP): tile listed in the table AND mask_mod  (what compiled FlexAttention computes, F2)."""
    tab = train_table(T, P)
    assert tab is not None
    num, idx, blk = tab
    L = 2 * T * P
    nblk = L // blk
    listed = np.zeros((nblk, nblk), dtype=bool)
    for i in range(nblk):
        listed[i, idx[i, : num[i]]] = True
    tok = np.arange(L)
    allowed = listed[tok[:, None] // blk, tok[None, :] // blk]
    return allowed & train_mask_mod(tok[:, None] // P, tok[None, :] // P, T)


def infer_allowed_tokens(t, P):
    tok = np.arange(t * P)
    return (tok[:, None] // P) >= (tok[None, :] // P)


# ----------------------------------------------------------------------------------------------------------------
# magnitude-preserving primitives (edm2/utils.py:83-158)

def normalize(x, dim=None):
    if dim is None:
        dim = list(range(1, x.ndim))
    n = torch.linalg.vector_norm(x, dim=dim, keepdim=True)
    return x / (EPS + n * math.sqrt(n.numel() / x.numel()))


def mp_silu(x):
    return F.silu(x) / SILU_DIV


def mp_sum(a, b, t):
    if isinstance(t, float):
        return (a + (b - a) * t) / math.sqrt((1 - t) ** 2 + t ** 2)
    t = t.reshape(-1, *([1] * (a.ndim - 1)))
    return (a + (b - a) * t) * ((1 - t) ** 2 + t ** 2) ** -0.5


def mp_cat(a, b, t=0.5):
    Na, Nb = a.shape[1], b.shape[1]
    C = math.sqrt((Na + Nb) / ((1 - t) ** 2 + t ** 2))
    return torch.cat([a * (C / math.sqrt(Na) * (1 - t)), b * (C / math.sqrt(Nb) * t)], dim=1)


def resample(x, mode, f=(1, 1)):
    """utils.py:94-107: 'down' = depthwise conv2d with outer(f, f) / sum^2, stride 2, padding (len - 1) // 2; 'up' = depthwise
    conv_transpose2d with 4 x that, stride 2."""
    if mode == "keep":
        return x
    if tuple(float(v) for v in f) == (1.0, 1.0):
        if mode == "down":                               # depthwise [1,1]x[1,1]/4, stride 2  == 2x2 mean
            return F.avg_pool2d(x, 2)
        return x.repeat_interleave(2, -1).repeat_interleave(2, -2)   # transposed conv with 4*[.25] == nearest x2
    t = torch.tensor([float(v) for v in f], dtype=x.dtype)
    t = t / t.sum()
    k2 = torch.outer(t, t)[None, None].repeat(x.shape[1], 1, 1, 1)
    pad = (len(f) - 1) // 2
    if mode == "down":
        return F.conv2d(x, k2, groups=x.shape[1], stride=2, padding=pad)
    return F.conv_transpose2d(x, 4 * k2, groups=x.shape[1], stride=2, padding=pad)


def mp_fourier(x, freqs, phases):
    return torch.cos(torch.outer(x, freqs) + phases) * math.sqrt(2)


# ----------------------------------------------------------------------------------------------------------------
# weights (edm2/conv.py:8-46)

def weight_forced(w):
    """The value the stored parameter is overwritten with in training mode (conv.py:16-18)."""
    return normalize(w)


def weight_effective(w, gain=1.0, training=True):
    """NormalizedWeight.forward: returns (effective weight, new stored weight)."""
    if training:
        w = _forced_inplace(w)
    fan_in = w[0].numel()
    return normalize(w) * (gain / math.sqrt(fan_in)), w


def _forced_inplace(w):
    # the reference overwrites the parameter under no_grad and then differentiates through the SECOND normalise
    # only; numerically: w_hat = normalize(w) is treated as the leaf.
    with torch.no_grad():
        w_hat = normalize(w)
    if w.requires_grad:
        w_hat = w_hat + (w - w.detach())     # value w_hat, gradient flows to w one-to-one (leaf identity)
    return w_hat


def mpconv(x, w_eff):
    if w_eff.ndim == 2:
        return x @ w_eff.t()
    return F.conv2d(x, w_eff, padding=w_eff.shape[-1] // 2)


# ----------------------------------------------------------------------------------------------------------------
# Gating (edm2/conv.py:104-127)

def gating(p, prefix, c_noise, n_ctx, training, just_2d=False):
    """c_noise (B, t or 2T). returns g (B, t), updated frame counter."""
    B, tt = c_noise.shape
    T = tt // 2 if training else tt
    if just_2d:
        pos = torch.zeros_like(c_noise)
    else:
        pos = (torch.arange(B * tt) % T).reshape(B, tt) + n_ctx
        pos = torch.log1p(pos.to(c_noise.dtype))
    mult, off = p[prefix + "mult"], p[prefix + "offset"]
    sv = c_noise * mult[0] + off[0] + pos * mult[1] + off[1]
    lo, hi = torch.sigmoid(p[prefix + "min_gating"]), torch.sigmoid(p[prefix + "max_gating"])
    return lo + (1 - lo) * hi * torch.sigmoid(sv), n_ctx + T


# ----------------------------------------------------------------------------------------------------------------
# gated causal conv (edm2/conv.py:59-95), fused form of SURVEY section 9

def gated_conv(p, prefix, x, B, c_noise, cache, update_cache, just_2d, training, new_p=None):
    """x (N,C,H,W), N = B*2T (train, (b s t) order) or B*t (eval).  returns y, cache."""
    w2, w2_new = weight_effective(p[prefix + "last_frame_conv.weight.weight"], 1.0, training)
    if new_p is not None:
        new_p[prefix + "last_frame_conv.weight.weight"] = w2_new.detach()
    y2 = F.conv2d(x, w2, padding=1)
    if just_2d:
        return y2, cache
    if cache is None:
        cache = {}
    w3, w3_new = weight_effective(p[prefix + "weight.weight"], 1.0, training)
    if new_p is not None:
        new_p[prefix + "weight.weight"] = w3_new.detach()
    N, C, H, W = x.shape
    g, n_new = gating(p, prefix + "gating.", c_noise, cache.get("n_context_frames", 0), training)
    pad = cache["activations"] if "activations" in cache else torch.ones(B, C, 2, H, W, dtype=x.dtype)
    if update_cache:
        cache["n_context_frames"] = n_new
    if training:
        T = N // (2 * B)
        clean = x.reshape(B, 2, T, C, H, W)[:, 0]                    # (B,T,C,H,W)
    else:
        T = N // B
        clean = x.reshape(B, T, C, H, W)
    ctx = torch.cat([pad.permute(0, 2, 1, 3, 4), clean], dim=1)      # (B,T+2,C,H,W) frames -2..T-1
    if update_cache:
        cache["activations"] = ctx[:, -2:].permute(0, 2, 1, 3, 4).detach().clone()
    a0 = ctx[:, 0:T].reshape(B * T, C, H, W)                         # frame t-2
    a1 = ctx[:, 1:T + 1].reshape(B * T, C, H, W)                     # frame t-1
    y3 = F.conv2d(a0, w3[:, :, 0], padding=1) + F.conv2d(a1, w3[:, :, 1], padding=1)
    if training:
        y3 = y3.reshape(B, 1, T, -1, H, W).expand(B, 2, T, y3.shape[1], H, W).reshape(N, -1, H, W)
    return mp_sum(y2, y3, g.reshape(-1)), cache


# ----------------------------------------------------------------------------------------------------------------
# rotary tables (edm2/attention/RoPe.py:21-32): angles and xPos scale are ROUNDED TO FP16 (part of the spec)

def rope_tables(inv_freq, scale_base_vec, seq_len, scale_base=64):
    t = torch.arange(seq_len, dtype=inv_freq.dtype)
    freqs = torch.outer(t, inv_freq)
    freqs = torch.cat([freqs, freqs], -1).to(torch.float16)
    power = (t - (seq_len // 2)) / scale_base
    scale = scale_base_vec[None, :] ** power[:, None]
    scale = torch.cat([scale, scale], -1).to(torch.float16)
    return freqs, scale          # (seq,d) fp16 each


def rot_half(x):
    a, b = x.chunk(2, dim=-1)
    return torch.cat([-b, a], dim=-1)


def rope_apply(q, k, inv_freq, scale_vec, training):
    """q,k (B,m,frames,P,d).  train: frames = 2T laid out (clean T, noisy T), both halves use positions 0..T-1."""
    nk = k.shape[2] // 2 if training else k.shape[2]
    ang, sc = rope_tables(inv_freq, scale_vec, nk)
    cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]       # fp16 math like the reference, then promoted
    sc = sc[:, None, :]
    if training:
        cos, sin, sc = (torch.cat([z, z], 0) for z in (cos, sin, sc))
    k = (k * cos + rot_half(k) * sin) / sc
    nq = q.shape[2]
    cq, sq, scq = cos[-nq:], sin[-nq:], sc[-nq:]
    q = (q * cq + rot_half(q) * sq) * scq
    return q, k


# ----------------------------------------------------------------------------------------------------------------
# attention (edm2/attention/attention_modules.py)

def _split_qkv(y, m):
    """y (N, 3C, H, W) with channel = (head*d + c)*3 + s  ->  q,k,v (N, m, P, d), each normalised over d."""
    N, C3, H, W = y.shape
    d = C3 // (3 * m)
    y = y.reshape(N, m, d, 3, H * W).permute(3, 0, 1, 4, 2)         # s N m P d
    y = normalize(y, dim=-1)
    return y[0], y[1], y[2]


def video_attention(p, prefix, x, B, m, cache, update_cache, just_2d, training, balance=0.3, new_p=None):
    if m == 0:
        return x, None
    N, C, H, W = x.shape
    P = H * W
    wq, wq_new = weight_effective(p[prefix + "attn_qkv.weight.weight"], 1.0, training)
    wp, wp_new = weight_effective(p[prefix + "attn_proj.weight.weight"], 1.0, training)
    if new_p is not None:
        new_p[prefix + "attn_qkv.weight.weight"] = wq_new.detach()
        new_p[prefix + "attn_proj.weight.weight"] = wp_new.detach()
    q, k, v = _split_qkv(F.conv2d(x, wq), m)
    d = q.shape[-1]
    if just_2d:
        o = F.scaled_dot_product_attention(q, k, v)                 # per frame, dense
        o = o.permute(0, 1, 3, 2).reshape(N, C, H, W)
        return mp_sum(x, F.conv2d(o, wp), balance), cache
    fr = N // B
    q, k, v = (z.reshape(B, fr, m, P, d).permute(0, 2, 1, 3, 4) for z in (q, k, v))   # B m fr P d
    if not training:
        if cache is not None:
            k = torch.cat([cache[0], k], dim=2)
            v = torch.cat([cache[1], v], dim=2)
        if update_cache:
            cache = (k, v)
    q, k = rope_apply(q, k, p[prefix + "rope.inv_freq"], p[prefix + "rope.scale"], training)
    q, k, v = (z.reshape(B, m, -1, d) for z in (q, k, v))
    if training:
        T = fr // 2
        allowed = torch.from_numpy(train_allowed_tokens(T, P))
    elif q.shape[2] == P:
        allowed = None
    elif q.shape[2] == k.shape[2]:
        allowed = torch.from_numpy(infer_allowed_tokens(fr, P))
    else:
        raise NotImplementedError("The inference mask is not implemented for this case")
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=allowed)
    o = o.reshape(B, m, fr, P, d).permute(0, 2, 1, 4, 3).reshape(N, C, H, W)
    return mp_sum(x, F.conv2d(o, wp), balance), cache


def frame_attention(p, prefix, x, m, training, balance=0.3, new_p=None):
    if m == 0:
        return x
    N, C, H, W = x.shape
    wq, wq_new = weight_effective(p[prefix + "attn_qkv.weight.weight"], 1.0, training)
    wp, wp_new = weight_effective(p[prefix + "attn_proj.weight.weight"], 1.0, training)
    if new_p is not None:
        new_p[prefix + "attn_qkv.weight.weight"] = wq_new.detach()
        new_p[prefix + "attn_proj.weight.weight"] = wp_new.detach()
    q, k, v = _split_qkv(F.conv2d(x, wq), m)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.permute(0, 1, 3, 2).reshape(N, C, H, W)
    return mp_sum(x, F.conv2d(o, wp), balance)


# ----------------------------------------------------------------------------------------------------------------
# network topology (edm2/networks_edm2.py:118-189)

def unet_layout(img_resolution, img_channels, label_dim, model_channels, channel_mult=(1, 2, 2, 4),
                channel_mult_noise=None, channel_mult_emb=None, num_blocks=3, video_attn_resolutions=(8,),
                frame_attn_resolutions=(16,), channels_per_head=64, **_):
    cblock = [model_channels * m for m in channel_mult]
    cnoise = model_channels * channel_mult_noise if channel_mult_noise is not None else cblock[0]
    cemb = model_channels * channel_mult_emb if channel_mult_emb is not None else max(cblock)

    def attn_kind(res):
        return "video" if res in video_attn_resolutions else "frame" if res in frame_attn_resolutions else None

    def blk(name, cin, cout, flavor, mode="keep", attention=None):
        heads = cout // channels_per_head if attention else 0
        return dict(kind="block", name=name, cin=cin, cout=cout, flavor=flavor, mode=mode,
                    attention=attention, heads=heads)

    enc, cout = [], img_channels + 1
    for level, ch in enumerate(cblock):
        res = img_resolution >> level
        if level == 0:
            enc.append(dict(kind="conv", name=f"{res}x{res}_conv", cin=cout, cout=ch))
            cout = ch
        else:
            enc.append(blk(f"{res}x{res}_down", cout, cout, "enc", "down"))
        for i in range(num_blocks):
            enc.append(blk(f"{res}x{res}_block{i}", cout, ch, "enc", attention=attn_kind(res)))
            cout = ch
    skips = [e["cout"] for e in enc]
    dec = []
    for level, ch in reversed(list(enumerate(cblock))):
        res = img_resolution >> level
        if level == len(cblock) - 1:
            dec.append(blk(f"{res}x{res}_in0", cout, cout, "dec", attention="video"))
            dec.append(blk(f"{res}x{res}_in1", cout, cout, "dec"))
        else:
            dec.append(blk(f"{res}x{res}_up", cout, cout, "dec", "up"))
        for i in range(num_blocks + 1):
            dec.append(blk(f"{res}x{res}_block{i}", cout + skips.pop(), ch, "dec", attention=attn_kind(res)))
            cout = ch
    return dict(enc=enc, dec=dec, cnoise=cnoise, cemb=cemb, cout=cout, label_dim=label_dim,
                img_channels=img_channels)


def block_forward(p, prefix, e, x, emb, B, c_noise, cache, update_cache, just_2d, training,
                  res_balance=0.3, clip_act=256, new_p=None):
    if cache is None:
        cache = {}
    x = resample(x, e["mode"], e.get("filter", (1, 1)))
    skipw = prefix + "conv_skip.weight.weight"
    if e["flavor"] == "enc":
        if e["cin"] != e["cout"]:
            w, wn = weight_effective(p[skipw], 1.0, training)
            if new_p is not None:
                new_p[skipw] = wn.detach()
            x = F.conv2d(x, w)
        x = normalize(x, dim=1)
    y, cache["conv_res0"] = gated_conv(p, prefix + "conv_res0.", mp_silu(x), B, c_noise, cache.get("conv_res0"),
                                       update_cache, just_2d, training, new_p)
    wl, wln = weight_effective(p[prefix + "emb_linear.weight.weight"], p[prefix + "emb_gain"], training)
    if new_p is not None:
        new_p[prefix + "emb_linear.weight.weight"] = wln.detach()
    c = emb @ wl.t() + 1
    y = mp_silu(y * c[:, :, None, None])
    y, cache["conv_res1"] = gated_conv(p, prefix + "conv_res1.", y, B, c_noise, cache.get("conv_res1"),
                                       update_cache, just_2d, training, new_p)
    if e["flavor"] == "dec" and e["cin"] != e["cout"]:
        w, wn = weight_effective(p[skipw], 1.0, training)
        if new_p is not None:
            new_p[skipw] = wn.detach()
        x = F.conv2d(x, w)
    x = mp_sum(x, y, res_balance)
    if e["attention"] == "video":
        x, cache["attn"] = video_attention(p, prefix + "attn.", x, B, e["heads"], cache.get("attn"), update_cache,
                                           just_2d, training, new_p=new_p)
    else:
        x = frame_attention(p, prefix + "attn.", x, e["heads"], training, new_p=new_p)
        cache["attn"] = None
    if clip_act is not None:
        x = x.clamp(-clip_act, clip_act)
    return x, cache


def unet_forward(p, cfg, x, c_noise, conditioning=None, cache=None, update_cache=False, just_2d=False,
                 training=True, label_balance=0.5, concat_balance=0.5, new_p=None, prefix=""):
    """x (B,t,C,H,W) -> (B,t,C,H,W), cache.   `new_p` (dict) receives the force-normalised weights (training)."""
    lay = unet_layout(**cfg)
    if cache is None:
        cache = {}
    B, tt = x.shape[:2]
    n_ctx = cache.get("n_context_frames", 0)
    if update_cache:
        T = tt // 2 if training else tt
        cache["n_context_frames"] = n_ctx + T           # Gating(out_res) counter (networks_edm2.py:197-198)
    x = x.reshape(B * tt, *x.shape[2:])
    cn = c_noise.reshape(-1)
    w, wn = weight_effective(p[prefix + "emb_noise.weight.weight"], 1.0, training)
    if new_p is not None:
        new_p[prefix + "emb_noise.weight.weight"] = wn.detach()
        # emb_time is evaluated (and force-normalised) by the reference although its output is unused (:207)
        new_p[prefix + "emb_time.weight.weight"] = weight_forced(p[prefix + "emb_time.weight.weight"]).detach() \
            if training else p[prefix + "emb_time.weight.weight"]
    emb = mp_fourier(cn, p[prefix + "emb_fourier_sigma.freqs"], p[prefix + "emb_fourier_sigma.phases"]) @ w.t()
    if lay["label_dim"] != 0 and conditioning is not None:
        wl, wln = weight_effective(p[prefix + "emb_label.weight.weight"], 1.0, training)
        if new_p is not None:
            new_p[prefix + "emb_label.weight.weight"] = wln.detach()
        oh = F.one_hot(conditioning.reshape(-1), lay["label_dim"]).to(emb.dtype) * math.sqrt(lay["label_dim"])
        emb = mp_sum(emb, oh @ wl.t(), 1 / 3)
    emb = mp_silu(emb)
    x = torch.cat([x, torch.ones_like(x[:, :1])], dim=1)
    skips = []
    for e in lay["enc"]:
        key, pre = ("enc", e["name"]), f"{prefix}enc.{e['name']}."
        if e["kind"] == "conv":
            x, cache[key] = gated_conv(p, pre, x, B, c_noise, cache.get(key), update_cache, just_2d, training, new_p)
        else:
            x, cache[key] = block_forward(p, pre, e, x, emb, B, c_noise, cache.get(key), update_cache, just_2d,
                                          training, new_p=new_p)
        skips.append(x)
    for e in lay["dec"]:
        key, pre = ("dec", e["name"]), f"{prefix}dec.{e['name']}."
        if "block" in e["name"]:
            x = mp_cat(x, skips.pop(), concat_balance)
        x, cache[key] = block_forward(p, pre, e, x, emb, B, c_noise, cache.get(key), update_cache, just_2d,
                                      training, new_p=new_p)
    x, cache["out_conv"] = gated_conv(p, prefix + "out_conv.", x, B, c_noise, cache.get("out_conv"), update_cache,
                                      just_2d, training, new_p)
    x = x.reshape(B, tt, *x.shape[1:]) * p[prefix + "out_gain"]
    return x, cache


def precond_forward(p, cfg, x, sigma, conditioning=None, cache=None, update_cache=False, just_2d=False,
                    training=True, sigma_data=0.5, new_p=None, prefix="unet."):
    """Precond.forward (networks_edm2.py:278-297), fp32."""
    if cache is None:
        cache = {}
    cache["shape"] = tuple(x.shape)
    x = x.float()
    s = sigma.float()[:, :, None, None, None]
    c_skip = sigma_data ** 2 / (s ** 2 + sigma_data ** 2)
    c_out = s * sigma_data / (s ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + s ** 2).sqrt()
    c_noise = sigma.float().log() / 4
    Fx, cache = unet_forward(p, cfg, c_in * x, c_noise, conditioning, cache, update_cache, just_2d, training,
                             new_p=new_p, prefix=prefix)
    return c_skip * x + c_out * Fx, cache


# ----------------------------------------------------------------------------------------------------------------
# loss (edm2/loss.py:17-47) with sigma and noise supplied (no RNG parity needed)

def fourier_mean_loss(coeff, sigma, num_terms=4):
    xl = torch.log10(sigma.reshape(-1))
    basis = [torch.full_like(xl, 0.5)]
    for n in range(1, num_terms):
        basis += [torch.cos(n * xl), torch.sin(n * xl)]
    return (10 ** (torch.stack(basis, -1) @ coeff)).reshape(sigma.shape)


def edm2_loss(p, cfg, images, sigma, eps, conditioning=None, just_2d=False, sigma_data=1.0, new_p=None):
    """images (B,T,C,H,W); sigma (B,2T) [(B,T) if just_2d]; eps like the concatenated input.
    returns (loss, un_weighted mean, D_x)."""
    B, T = images.shape[:2]
    cat = images if just_2d else torch.cat([images, images], 1)
    if conditioning is not None and not just_2d:
        conditioning = torch.cat([conditioning, conditioning], 1)
    x = cat + sigma[:, :, None, None, None] * eps
    D, _ = precond_forward(p, cfg, x, sigma, conditioning, just_2d=just_2d, training=True, sigma_data=sigma_data,
                           new_p=new_p)
    losses = ((D[:, -T:] - images) ** 2).mean(dim=(-1, -2, -3))
    sg = sigma[:, -T:]
    losses = losses * (sg ** 2 + sigma_data ** 2) / (sg * sigma_data) ** 2
    unweighted = losses.mean().detach()
    with torch.no_grad():
        mean_loss = fourier_mean_loss(p["noise_weight.fourier_approximator.coefficients"], sg)
    return (losses / mean_loss).mean(), unweighted, D


# ----------------------------------------------------------------------------------------------------------------
# sampler (edm2/sampler.py:12-85), S_churn = 0 / guidance = 1 path with the initial noise supplied

def edm_t_steps(num_steps, sigma_min, sigma_max, rho):
    i = torch.arange(num_steps, dtype=torch.float32)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros(1)])


def edm_sample_frame(p, cfg, cache, noise, conditioning=None, num_steps=32, sigma_min=0.002, sigma_max=80.0,
                     rho=7, sigma_data=0.5, guidance=1.0, S_churn=0.0, S_min=0.0, S_max=float("inf"), S_noise=1.0,
                     churn_noise=None, target=None):
    """One autoregressive frame (reference edm2/sampler.py:12-85): noise (B,1,C,H,W) ~ N(0,1).  Heun; cache updated on the
    last Euler evaluation only, and not at all when `target` is given (:63).  guidance != 1 (:25-32): every evaluation is
    lerp(D_2d, D, guidance) with D_2d an eval-mode `just_2d` evaluation WITHOUT cache.  S_churn (:52-59): churn_noise[i] is the
    N(0,1) draw of step i.  target (:46-48, 78-83): x starts at target + noise * t0.
    Returns (x, cache) -- or (x, cache, mse, mse_pred) when target is given."""
    B = noise.shape[0]
    t_steps = edm_t_steps(num_steps, sigma_min, sigma_max, rho)
    x_next = noise * t_steps[0]
    if target is not None:
        x_next = x_next + target

    def den(x, t, cache, upd):
        sig = torch.ones(B, 1) * t
        D, cache = precond_forward(p, cfg, x, sig, conditioning, cache=cache, update_cache=upd,
                                   training=False, sigma_data=sigma_data)
        if guidance != 1:
            ref, _ = precond_forward(p, cfg, x, sig, conditioning, cache=None, update_cache=False, just_2d=True,
                                     training=False, sigma_data=sigma_data)
            D = ref.lerp(D, guidance)
        return D, cache

    mse, mse_pred = [], []
    with torch.no_grad():
        for i in range(num_steps):
            t_cur, t_next = t_steps[i], t_steps[i + 1]
            if S_churn > 0 and S_min <= t_cur <= S_max:
                gamma = min(S_churn / num_steps, 2 ** 0.5 - 1)
                t_hat = t_cur + gamma * t_cur
                x_hat = x_next + (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise * churn_noise[i]
            else:
                t_hat, x_hat = t_cur, x_next
            x_pred, cache = den(x_hat, t_hat, cache, i == num_steps - 1 and target is None)
            d_cur = (x_hat - x_pred) / t_hat
            x_next = x_hat + (t_next - t_hat) * d_cur
            if i < num_steps - 1:
                x_pred, _ = den(x_next, t_next, cache, False)
                d_prime = (x_next - x_pred) / t_next
                x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
            if target is not None:
                mse_pred.append(torch.mean((x_pred - target) ** 2).item())
                mse.append(torch.mean((x_next - target) ** 2).item())
    if target is not None:
        return x_next, cache, mse, mse_pred
    return x_next, cache
