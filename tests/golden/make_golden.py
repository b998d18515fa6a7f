"""Generate the golden fixtures (tests/golden/*.npz) by IMPORTING the reference (read-only, /root/reference) on CPU
in the build container.  Run from the repo root:   python tests/golden/make_golden.py

Fixtures are data (inputs + expected outputs, fp32 / int32).  Parameters are regenerated at test time by
tests/golden/paramgen.py from a seed (and were loaded into the reference modules here with strict=True), so the
files stay small.  Fixture IDs follow SURVEY.md section 8(c) (G1..G9).
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refshim  # noqa: E402
import paramgen  # noqa: E402

edm2 = _refshim.install()
from edm2.networks_edm2 import UNet, Precond, Block  # noqa: E402
from edm2.conv import MPConv, MPCausal3DGatedConv, Gating, NormalizedWeight  # noqa: E402
from edm2.attention import VideoAttention, FrameAttention  # noqa: E402
from edm2.attention.attention_masking import make_train_mask, make_infer_mask  # noqa: E402
from edm2.attention.RoPe import RotaryEmbedding  # noqa: E402
from edm2.loss import EDM2Loss  # noqa: E402
from edm2.sampler import edm_sampler_with_mse  # noqa: E402

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def load_sub(module, params, prefix):
    sd = {k[len(prefix):]: v.clone() for k, v in params.items() if k.startswith(prefix)}
    module.load_state_dict(sd, strict=True)


# ------------------------------------------------------------------ G1 mask tables
def g1():
    out = {}
    for (B, m, T, P) in [(2, 4, 64, 64), (2, 8, 32, 16), (2, 8, 64, 16), (2, 2, 8, 64), (1, 1, 4, 256), (1, 1, 3, 128)]:
        bm = make_train_mask(B, m, T, P)
        tag = f"train_{T}_{P}"
        out[tag + "_num"] = bm.kv_num_blocks.to(torch.int32)
        out[tag + "_idx"] = bm.kv_indices.to(torch.int32)
        out[tag + "_blk"] = np.int32(bm.BLOCK_SIZE[0])
        L = 2 * T * P
        if L <= 4096:   # dense token-level truth = listed tiles AND mask_mod
            q = torch.arange(L)[:, None]
            k = torch.arange(L)[None, :]
            dense = bm.to_dense()[0, 0].bool().repeat_interleave(bm.BLOCK_SIZE[0], 0).repeat_interleave(bm.BLOCK_SIZE[1], 1)
            allowed = dense & bm.mask_mod(0, 0, q, k)
            out[tag + "_allowed_packed"] = np.packbits(npy(allowed), axis=1)
        # frame-granularity view of mask_mod alone
        qf = torch.arange(2 * T)[:, None] * P
        kf = torch.arange(2 * T)[None, :] * P
        out[tag + "_maskmod_frames"] = bm.mask_mod(0, 0, qf, kf)
    assert make_train_mask(1, 1, 3, 64) is None
    for (t, P) in [(4, 64), (8, 16), (5, 64), (1, 64), (6, 256)]:
        score_mod, bm = make_infer_mask(2, 2, t, P)
        tag = f"infer_{t}_{P}"
        L = t * P
        q = torch.arange(L)[:, None]
        k = torch.arange(L)[None, :]
        if bm is None:
            out[tag + "_kind"] = "score_mod"
            allowed = torch.isfinite(score_mod(torch.zeros(L, L), 0, 0, q, k))
        else:
            dense = bm.to_dense()[0, 0].bool().repeat_interleave(bm.BLOCK_SIZE[0], 0).repeat_interleave(bm.BLOCK_SIZE[1], 1)[:L, :L]
            allowed = dense & bm.mask_mod(0, 0, q, k)
            if (t * P) % 128 != 0 and P < 128:
                out[tag + "_kind"] = "dense"
            else:
                out[tag + "_kind"] = "table"
                out[tag + "_num"] = bm.kv_num_blocks.to(torch.int32)
                out[tag + "_idx"] = bm.kv_indices.to(torch.int32)
        out[tag + "_allowed_packed"] = np.packbits(npy(allowed), axis=1)
    save("g1_masks", **out)


# ------------------------------------------------------------------ G2 weights / MPConv
def g2():
    out = {}
    g = torch.Generator().manual_seed(2)
    for tag, cin, cout, kernel, xshape in [("lin", 24, 16, [], (5, 24)), ("c1", 16, 24, [1, 1], (3, 16, 6, 6)),
                                           ("c3", 8, 16, [3, 3], (3, 8, 6, 6))]:
        mod = MPConv(cin, cout, kernel)
        w_in = torch.randn(mod.weight.weight.shape, generator=g) * 1.7
        x = torch.randn(xshape, generator=g, requires_grad=True)
        with torch.no_grad():
            mod.weight.weight.copy_(w_in)
        mod.train()
        y = mod(x, gain=0.8)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        with torch.no_grad():
            w_after = mod.weight.weight.clone()
        mod.eval()
        w_eff_eval = mod.weight(0.8)
        out.update({f"{tag}_w_in": w_in, f"{tag}_x": x, f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_w_after": w_after,
                    f"{tag}_gx": x.grad, f"{tag}_gw": mod.weight.weight.grad, f"{tag}_w_eff_eval_after": w_eff_eval})
    save("g2_weights", **out)


def gate_params(g):
    return {"gating.offset": 0.3 * torch.randn(2, generator=g),
            "gating.mult": torch.tensor([1.5, -0.5]) + 0.2 * torch.randn(2, generator=g),
            "gating.max_gating": -1.0 + 0.3 * torch.randn((), generator=g),
            "gating.min_gating": -3.0 + 0.3 * torch.randn((), generator=g)}


# ------------------------------------------------------------------ G3 gated causal conv, G4 gating
def g3():
    out = {}
    g = torch.Generator().manual_seed(3)
    B, T, Ci, Co, H = 2, 4, 8, 8, 8
    conv = MPCausal3DGatedConv(Ci, Co, [3, 3, 3])
    sd = {"last_frame_conv.weight.weight": torch.randn(Co, Ci, 3, 3, generator=g),
          "weight.weight": torch.randn(Co, Ci, 2, 3, 3, generator=g)}
    sd.update(gate_params(g))
    for k in list(sd):
        if k.endswith("weight.weight"):
            sd[k] = paramgen.O.normalize(paramgen.O.normalize(sd[k]))   # fixed point of the forced normalisation
    conv.load_state_dict(sd, strict=True)
    out.update({"p_" + k: v for k, v in sd.items()})
    # training
    x = torch.randn(B * 2 * T, Ci, H, H, generator=g, requires_grad=True)
    cn = torch.randn(B, 2 * T, generator=g) * 0.5
    conv.train()
    y, _ = conv(x, None, B, cn)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out.update(train_x=x, train_cn=cn, train_y=y, train_gy=gy, train_gx=x.grad)
    for n, prm in conv.named_parameters():
        out["train_g_" + n] = prm.grad
    out["train_w2_after"] = conv.last_frame_conv.weight.weight
    y2d, _ = conv(x.detach(), None, B, cn, just_2d=True)
    out["train_y_just2d"] = y2d
    # eval, uncached, then cached 1-step and 2-step
    conv.eval()
    t_all = 6
    xe = torch.randn(B * t_all, Ci, H, H, generator=g)
    cne = torch.randn(B, t_all, generator=g) * 0.5
    with torch.no_grad():
        ye, _ = conv(xe, None, B, cne)
        xs = xe.reshape(B, t_all, Ci, H, H)
        y4, cache = conv(xs[:, :4].reshape(-1, Ci, H, H), None, B, cne[:, :4], cache=None, update_cache=True)
        c_act4, c_n4 = cache["activations"].clone(), cache["n_context_frames"]
        y5, cache = conv(xs[:, 4:5].reshape(-1, Ci, H, H), None, B, cne[:, 4:5], cache=cache, update_cache=True)
        c_act5, c_n5 = cache["activations"].clone(), cache["n_context_frames"]
        y6, cache = conv(xs[:, 5:6].reshape(-1, Ci, H, H), None, B, cne[:, 5:6], cache=cache, update_cache=False)
    out.update(eval_x=xe, eval_cn=cne, eval_y=ye, eval_y4=y4, eval_y5=y5, eval_y6=y6, eval_cache_act4=c_act4,
               eval_cache_n4=np.int64(c_n4), eval_cache_act5=c_act5, eval_cache_n5=np.int64(c_n5))
    save("g3_gated_conv", **out)

    # G4 gating
    gt = Gating()
    gsd = {k[len("gating."):]: v for k, v in gate_params(g).items()}
    gt.load_state_dict(gsd)
    cn = torch.randn(3, 8, generator=g)
    gt.train()
    a, na = gt(cn, 0)
    gt.eval()
    b, nb_ = gt(cn, 0)
    c, nc = gt(cn, 5)
    d, nd = gt(cn, 0, just_2d=True)
    save("g4_gating", **{"p_" + k: v for k, v in gsd.items()}, cn=cn, train=a, train_n=np.int64(na), eval=b,
         eval_n=np.int64(nb_), eval_ctx5=c, eval_ctx5_n=np.int64(nc), just2d=d, just2d_n=np.int64(nd))


# ------------------------------------------------------------------ G5 RoPE
def g5():
    out = {}
    g = torch.Generator().manual_seed(5)
    rope = RotaryEmbedding(64)
    for T in (4, 8, 64, 256):
        fr, sc = rope.make_rotary_embedding(T)
        out[f"T{T}_freqs_f16"] = fr.squeeze(1)
        out[f"T{T}_scale_f16"] = sc.squeeze(1)
        out[f"T{T}_cos"] = fr.squeeze(1).cos().float()
        out[f"T{T}_sin"] = fr.squeeze(1).sin().float()
    q = torch.randn(1, 2, 8, 3, 64, generator=g)
    k = torch.randn(1, 2, 8, 3, 64, generator=g)
    rope.train()
    qt, kt = rope(q, k)
    rope.eval()
    qe, ke = rope(q[:, :, -2:], k)
    out.update(q=q, k=k, q_train=qt, k_train=kt, q_eval=qe, k_eval=ke)
    save("g5_rope", **out)


def compiled_flex_check(att, x, B, y_dense, what):
    """The same training-mode call through the reference's REAL compiled FlexAttention (forward only): must equal the
    dense `table AND mask_mod` stand-in the gradients were taken through, and must differ from `mask_mod` alone whenever a
    frame has fewer than 128 tokens (the F2 quirk: eager flex_attention, which ignores the block table, gives that answer)."""
    with torch.no_grad(), _refshim.reference_compiled_flex():
        y_c, _ = att(x.detach(), B)
    d = (y_c - y_dense.detach()).abs().max().item()
    # ... and what `mask_mod` alone -- the answer of the un-compiled flex_attention, which ignores the block table -- would have
    # given: the F2 quirk (nonzero whenever a frame has fewer than 128 tokens)
    from edm2.attention import attention_modules as am

    def mask_mod_only(q, k, v, score_mod=None, block_mask=None):
        qi, ki = torch.arange(q.shape[-2])[:, None], torch.arange(k.shape[-2])[None, :]
        return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=block_mask.mask_mod(0, 0, qi, ki))
    saved, am.compiled_flex_attention = am.compiled_flex_attention, mask_mod_only
    try:
        with torch.no_grad():
            y_e, _ = att(x.detach(), B)
    finally:
        am.compiled_flex_attention = saved
    q = (y_e - y_c).abs().max().item()
    print(f"  compiled FlexAttention vs dense(table AND mask_mod) [{what}]: max |diff| {d:.2e};  vs mask_mod alone (not what the reference computes): {q:.2e}")
    assert d <= 1e-6, (what, d)
    return y_c


def attn_params(C, g, video=True):
    p = {"attn_qkv.weight.weight": torch.randn(3 * C, C, 1, 1, generator=g),
         "attn_proj.weight.weight": torch.randn(C, C, 1, 1, generator=g)}
    p = {k: paramgen.O.normalize(paramgen.O.normalize(v)) for k, v in p.items()}
    if video:
        p["rope.inv_freq"] = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
        p["rope.scale"] = (torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)
    return p


# ------------------------------------------------------------------ G6 attention modules
def g6():
    out = {}
    g = torch.Generator().manual_seed(6)
    for tag, T, H, C, m, B in [("a", 4, 8, 64, 1, 2), ("b", 2, 16, 64, 1, 1), ("c", 8, 4, 128, 2, 1)]:
        att = VideoAttention(C, m)
        p = attn_params(C, g)
        att.load_state_dict(p, strict=True)
        out.update({f"{tag}_p_{k}": v for k, v in p.items()})
        x = torch.randn(B * 2 * T, C, H, H, generator=g, requires_grad=True)
        att.train()
        y, _ = att(x, B)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_gx": x.grad,
                    f"{tag}_g_qkv": att.attn_qkv.weight.weight.grad, f"{tag}_g_proj": att.attn_proj.weight.weight.grad})
        out[f"{tag}_y_compiledflex"] = compiled_flex_check(att, x, B, y, f"G6 {tag}: T={T} P={H * H} heads={m}")
        y2d, _ = att(x.detach(), B, just_2d=True)
        out[f"{tag}_y_just2d"] = y2d
        if tag == "a":
            att.eval()
            t_all = 6
            xe = torch.randn(B * t_all, C, H, H, generator=g)
            xs = xe.reshape(B, t_all, C, H, H)
            with torch.no_grad():
                ye, _ = att(xe, B)                                                        # causal prefill, 6 frames
                y4, cache = att(xs[:, :4].reshape(-1, C, H, H), B, None, update_cache=True)
                y5, cache = att(xs[:, 4:5].reshape(-1, C, H, H), B, cache, update_cache=True)
                k5 = cache[0].clone()
                y6, cache2 = att(xs[:, 5:6].reshape(-1, C, H, H), B, cache, update_cache=False)
                with _refshim.reference_compiled_flex():                                  # the causal prefill, compiled
                    ye_c, _ = att(xe, B)
            d = (ye_c - ye).abs().max().item()
            print(f"  compiled FlexAttention vs dense, causal prefill of 6 frames: max |diff| {d:.2e}")
            assert d <= 1e-6, d
            out.update(a_eval_x=xe, a_eval_y=ye, a_eval_y4=y4, a_eval_y5=y5, a_eval_y6=y6, a_eval_k5=k5,
                       a_eval_v5=cache[1], a_eval_y_compiledflex=ye_c)
    fa = FrameAttention(64, 1)
    p = attn_params(64, g, video=False)
    fa.load_state_dict(p, strict=True)
    x = torch.randn(6, 64, 8, 8, generator=g, requires_grad=True)
    fa.train()
    y, _ = fa(x)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out.update({"f_p_" + k: v for k, v in p.items()})
    out.update(f_x=x, f_y=y, f_gy=gy, f_gx=x.grad, f_g_qkv=fa.attn_qkv.weight.weight.grad)
    save("g6_attention", **out)


# ------------------------------------------------------------------ G6b attention modules with 16- / 32-channel heads
def g6b():
    """VideoAttention / FrameAttention with heads of 16 and 32 channels (Block(channels_per_head=...), networks_edm2.py:28;
    the reference's own tests use 16: consistency_test.py:39,61): training forward + backward, just_2d, and -- for the
    16-channel case -- causal prefill and two cached one-frame steps."""
    out = {}
    g = torch.Generator().manual_seed(66)

    def params(C, d, video=True):
        p = {"attn_qkv.weight.weight": torch.randn(3 * C, C, 1, 1, generator=g),
             "attn_proj.weight.weight": torch.randn(C, C, 1, 1, generator=g)}
        p = {k: paramgen.O.normalize(paramgen.O.normalize(v)) for k, v in p.items()}
        if video:
            p["rope.inv_freq"] = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
            p["rope.scale"] = (torch.arange(0, d, 2) + 0.4 * d) / (1.4 * d)
        return p
    for tag, T, H, C, m, B in [("h16", 4, 8, 64, 4, 2), ("h32", 2, 8, 64, 2, 1)]:
        att = VideoAttention(C, m)
        p = params(C, C // m)
        att.load_state_dict(p, strict=True)
        out.update({f"{tag}_p_{k}": v for k, v in p.items()})
        x = torch.randn(B * 2 * T, C, H, H, generator=g, requires_grad=True)
        att.train()
        y, _ = att(x, B)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_gx": x.grad,
                    f"{tag}_g_qkv": att.attn_qkv.weight.weight.grad, f"{tag}_g_proj": att.attn_proj.weight.weight.grad})
        out[f"{tag}_y_compiledflex"] = compiled_flex_check(att, x, B, y, f"G6b {tag}: T={T} P={H * H} heads={m}")
        y2d, _ = att(x.detach(), B, just_2d=True)
        out[f"{tag}_y_just2d"] = y2d
        if tag == "h16":
            att.eval()
            t_all = 6
            xe = torch.randn(B * t_all, C, H, H, generator=g)
            xs = xe.reshape(B, t_all, C, H, H)
            with torch.no_grad():
                ye, _ = att(xe, B)
                y4, cache = att(xs[:, :4].reshape(-1, C, H, H), B, None, update_cache=True)
                y5, cache = att(xs[:, 4:5].reshape(-1, C, H, H), B, cache, update_cache=True)
                y6, _ = att(xs[:, 5:6].reshape(-1, C, H, H), B, cache, update_cache=False)
            out.update(h16_eval_x=xe, h16_eval_y=ye, h16_eval_y4=y4, h16_eval_y5=y5, h16_eval_y6=y6)
    fa = FrameAttention(32, 2)
    p = params(32, 16, video=False)
    fa.load_state_dict(p, strict=True)
    x = torch.randn(6, 32, 8, 8, generator=g, requires_grad=True)
    fa.train()
    y, _ = fa(x)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out.update({"f_p_" + k: v for k, v in p.items()})
    out.update(f_x=x, f_y=y, f_gy=gy, f_gx=x.grad, f_g_qkv=fa.attn_qkv.weight.weight.grad)
    save("g6b_attention_heads", **out)


# ------------------------------------------------------------------ G7 blocks
def g7():
    out = {}
    cemb = 32
    for tag, kw, cin, cout, H, T, B in [
            ("enc", dict(flavor="enc", resample_mode="down", attention="frame"), 32, 64, 16, 2, 1),
            ("dec", dict(flavor="dec", resample_mode="up", attention="video"), 96, 64, 4, 4, 1)]:
        e = dict(kind="block", name="blk", cin=cin, cout=cout, flavor=kw["flavor"], mode=kw["resample_mode"],
                 attention=kw["attention"], heads=cout // 64)
        shapes = {}
        pre = ""
        shapes[pre + "emb_gain"] = ()
        shapes[pre + "emb_linear.weight.weight"] = (cout, cemb)
        c0 = cout if kw["flavor"] == "enc" else cin
        shapes.update(paramgen._conv_keys(pre + "conv_res0.", c0, cout))
        shapes.update(paramgen._conv_keys(pre + "conv_res1.", cout, cout))
        shapes[pre + "conv_skip.weight.weight"] = (cout, cin, 1, 1)
        shapes[pre + "attn.attn_qkv.weight.weight"] = (3 * cout, cout, 1, 1)
        shapes[pre + "attn.attn_proj.weight.weight"] = (cout, cout, 1, 1)
        if kw["attention"] == "video":
            shapes[pre + "attn.rope.inv_freq"] = (32,)
            shapes[pre + "attn.rope.scale"] = (32,)
        seed = 70 if tag == "enc" else 71
        p = paramgen.prenormalise(paramgen.fill(shapes, seed))
        blk = Block(cin, cout, cemb, **kw)
        blk.load_state_dict({k: v.clone() for k, v in p.items()}, strict=True)
        g = torch.Generator().manual_seed(seed + 100)
        N = B * 2 * T
        x = torch.randn(N, cin, H, H, generator=g, requires_grad=True)
        emb = torch.randn(N, cemb, generator=g, requires_grad=True)
        cn = torch.randn(B, 2 * T, generator=g) * 0.5
        blk.train()
        y, _ = blk(x, emb, B, cn)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out.update({f"{tag}_seed": np.int64(seed), f"{tag}_x": x, f"{tag}_emb": emb, f"{tag}_cn": cn, f"{tag}_y": y,
                    f"{tag}_gy": gy, f"{tag}_gx": x.grad, f"{tag}_gemb": emb.grad})
        for n, prm in blk.named_parameters():
            if prm.grad is not None and (prm.numel() <= 4096 or n.endswith("conv_skip.weight.weight")):
                out[f"{tag}_g_{n}"] = prm.grad
            if prm.grad is not None:
                out[f"{tag}_gn_{n}"] = prm.grad.norm()
    save("g7_blocks", **out)


# ------------------------------------------------------------------ G13 resampling filters other than [1, 1]
def g13():
    """`resample(x, f, mode)` of the reference (edm2/utils.py:94-107) for f = [1, 3, 3, 1] and [1, 2, 3, 3, 2, 1] (values and input
    gradients), and one encoder Block with resample_mode='down', resample_filter=[1, 3, 3, 1] (networks_edm2.py:26,66): output
    and input gradient.  No BASELINE configuration uses such a filter; the fixture pins the generalisation."""
    from edm2.utils import resample
    out = {}
    g = torch.Generator().manual_seed(130)
    for tag, f in (("f4", [1, 3, 3, 1]), ("f6", [1, 2, 3, 3, 2, 1])):
        out[f"{tag}_f"] = np.float32(f)
        for mode, shape in (("down", (3, 8, 12, 16)), ("up", (3, 8, 6, 4))):
            x = torch.randn(*shape, generator=g, requires_grad=True)
            y = resample(x, f=f, mode=mode)
            gy = torch.randn(y.shape, generator=g)
            (y * gy).sum().backward()
            out.update({f"{tag}_{mode}_x": x, f"{tag}_{mode}_y": y, f"{tag}_{mode}_gy": gy, f"{tag}_{mode}_gx": x.grad})
    cemb, cin, cout, H, T, B = 32, 32, 32, 16, 2, 1
    shapes = {"emb_gain": (), "emb_linear.weight.weight": (cout, cemb)}
    shapes.update(paramgen._conv_keys("conv_res0.", cout, cout))
    shapes.update(paramgen._conv_keys("conv_res1.", cout, cout))
    p = paramgen.prenormalise(paramgen.fill(shapes, 131))
    blk = Block(cin, cout, cemb, flavor="enc", resample_mode="down", resample_filter=[1, 3, 3, 1])
    blk.load_state_dict({k: v.clone() for k, v in p.items()}, strict=True)
    N = B * 2 * T
    x = torch.randn(N, cin, H, H, generator=g, requires_grad=True)
    emb = torch.randn(N, cemb, generator=g, requires_grad=True)
    cn = torch.randn(B, 2 * T, generator=g) * 0.5
    blk.train()
    y, _ = blk(x, emb, B, cn)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out.update(blk_seed=np.int64(131), blk_x=x, blk_emb=emb, blk_cn=cn, blk_y=y, blk_gy=gy, blk_gx=x.grad, blk_gemb=emb.grad)
    save("g13_resample_filter", **out)


SMALL_CFG = dict(img_resolution=32, img_channels=4, label_dim=4, model_channels=16, channel_mult=[1, 4, 4],
                 num_blocks=1, video_attn_resolutions=[8], frame_attn_resolutions=[16])
C1_CFG = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=16, channel_mult=[1, 2, 4, 8],
              num_blocks=1, video_attn_resolutions=[8], frame_attn_resolutions=[16])


def build_precond(cfg, seed, sigma_data):
    p = paramgen.prenormalise(paramgen.precond_params(cfg, seed))
    unet = UNet(**cfg)
    net = Precond(unet, use_fp16=False, sigma_data=sigma_data)
    missing = net.load_state_dict({k: v.clone() for k, v in p.items()}, strict=True)
    return net, p


# ------------------------------------------------------------------ G8 UNet / Precond / loss
def g8():
    out = {}
    for tag, cfg, seed, B, T in [("small", SMALL_CFG, 80, 1, 4), ("c1", C1_CFG, 81, 1, 2)]:
        net, p = build_precond(cfg, seed, 1.0)
        net.train()
        C, R = cfg["img_channels"], cfg["img_resolution"]
        g = torch.Generator().manual_seed(seed + 100)
        images = torch.randn(B, T, C, R, R, generator=g)
        labels = torch.randint(0, 4, (B, T), generator=g)
        for mode in ("3d", "2d"):
            just_2d = mode == "2d"
            nt = T if just_2d else 2 * T
            sigma = (torch.randn(B, nt, generator=g) * 1.0 + 0.4).exp()
            if not just_2d:
                sigma[:, :T] = torch.rand(B, 1, generator=g) * 0.5
            eps = torch.randn(B, nt, C, R, R, generator=g)
            # replicate EDM2Loss.__call__ with sigma given and the noise we supply (loss.py:30-46)
            cat = images if just_2d else torch.cat([images, images], 1)
            cond = labels if just_2d else torch.cat([labels, labels], 1)
            net.zero_grad()
            x_in = cat + sigma[:, :, None, None, None] * eps
            Dx, _ = net(x_in, sigma, cond, just_2d=just_2d)
            losses = ((Dx[:, -T:] - images) ** 2).mean(dim=(-1, -2, -3))
            sg = sigma[:, -T:]
            losses = losses * (sg ** 2 + 1.0 ** 2) / (sg * 1.0) ** 2
            unw = losses.mean().detach()
            loss = (losses / net.noise_weight.calculate_mean_loss(sg)).mean()
            loss.backward()
            out.update({f"{tag}_{mode}_sigma": sigma, f"{tag}_{mode}_eps": eps, f"{tag}_{mode}_Dx": Dx,
                        f"{tag}_{mode}_loss": loss, f"{tag}_{mode}_unweighted": unw})
            gn = {}
            for n, prm in net.named_parameters():
                if prm.grad is not None:
                    gn[n] = float(prm.grad.norm())
                    if prm.numel() <= 2048 and tag == "small":
                        out[f"{tag}_{mode}_g_{n}"] = prm.grad.clone()
            out[f"{tag}_{mode}_gradnorm_names"] = np.array(sorted(gn))
            out[f"{tag}_{mode}_gradnorm_vals"] = np.array([gn[k] for k in sorted(gn)], dtype=np.float64)
            unused = sorted(n for n, prm in net.named_parameters() if prm.grad is None)
            out[f"{tag}_{mode}_unused"] = np.array(unused)
        out.update({f"{tag}_seed": np.int64(seed), f"{tag}_images": images, f"{tag}_labels": labels})
        # also cross-check with the real EDM2Loss when sigma is passed (noise drawn internally -> only finite check)
        l, u = EDM2Loss(P_mean=0.4, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5)(net, images, labels)
        assert torch.isfinite(l)
    save("g8_unet", **out)


# ------------------------------------------------------------------ G9 sampler / rollout
def g9():
    out = {}
    for (n, smin, smax, rho) in [(32, 0.002, 80, 7), (16, 0.01, 80, 2)]:
        i = torch.arange(n, dtype=torch.float32)
        t = (smax ** (1 / rho) + i / (n - 1) * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho
        out[f"tsteps_{n}_{rho}"] = torch.cat([t, torch.zeros(1)])
    net, p = build_precond(SMALL_CFG, 90, 0.5)
    net.eval()
    g = torch.Generator().manual_seed(190)
    B, t0 = 1, 4
    ctx = torch.randn(B, t0, 4, 32, 32, generator=g)
    lab = torch.randint(0, 4, (B, t0), generator=g)
    with torch.no_grad():
        Dctx, cache = net(ctx, torch.ones(B, t0) * 0.05, lab, update_cache=True)
    out.update(seed=np.int64(90), ctx=ctx, ctx_labels=lab, prefill_D=Dctx)
    frames = []
    noises = []
    real_randn = torch.randn
    for step in range(2):
        noise = real_randn(B, 1, 4, 32, 32, generator=g)
        noises.append(noise)
        calls = {"n": 0}

        def fake_randn(*a, **k):
            calls["n"] += 1
            return noise.clone()
        torch.randn = fake_randn
        try:
            x, _, _, cache = edm_sampler_with_mse(net, cache, conditioning=torch.full((B, 1), 1 + step), num_steps=4,
                                                  sigma_min=0.01, sigma_max=80, rho=2, guidance=1, S_churn=0)
        finally:
            torch.randn = real_randn
        assert calls["n"] == 1
        frames.append(x)
    out.update(noise=torch.stack(noises), frames=torch.stack(frames))
    blk = cache[("enc", "8x8_block0")]
    out.update(cache_n_ctx=np.int64(cache["n_context_frames"]), cache_conv0_act=blk["conv_res0"]["activations"],
               cache_conv0_n=np.int64(blk["conv_res0"]["n_context_frames"]), cache_attn_k=blk["attn"][0],
               cache_attn_v=blk["attn"][1])
    save("g9_sampler", **out)


# ------------------------------------------------------------------ G9b the sampler's side branches
def g9b():
    """edm_sampler_with_mse beyond its default path (reference edm2/sampler.py): `guidance != 1` (:25-32: a second, eval-mode
    `just_2d=True` evaluation without cache + lerp), `S_churn > 0` (:52-59: noise injection before every Euler evaluation) and
    `target=` (:46-48, 78-83: start from target + noise, per-step MSE lists, cache NOT updated).  Every case starts from the
    G9 prefill cache; all random draws are replaced by recorded tensors (torch.randn: the initial noise; torch.randn_like: the
    churn noise of each step)."""
    import copy
    net, p = build_precond(SMALL_CFG, 90, 0.5)
    net.eval()
    g = torch.Generator().manual_seed(190)
    B, t0 = 1, 4
    ctx = torch.randn(B, t0, 4, 32, 32, generator=g)
    lab = torch.randint(0, 4, (B, t0), generator=g)
    with torch.no_grad():
        _, cache0 = net(ctx, torch.ones(B, t0) * 0.05, lab, update_cache=True)
    g2 = torch.Generator().manual_seed(191)
    noise = torch.randn(B, 1, 4, 32, 32, generator=g2)
    churn = torch.randn(4, B, 1, 4, 32, 32, generator=g2)
    target = torch.randn(B, 1, 4, 32, 32, generator=g2) * 0.5
    out = dict(seed=np.int64(90), ctx=ctx, ctx_labels=lab, noise=noise, churn_noise=churn, target=target)
    real_randn, real_like = torch.randn, torch.randn_like
    cases = dict(guid=dict(guidance=1.5, S_churn=0), churn=dict(guidance=1, S_churn=8, S_noise=1),
                 target=dict(guidance=1, S_churn=0, target=target), all=dict(guidance=0.7, S_churn=8, target=target))
    for tag, kw in cases.items():
        cache = copy.deepcopy(cache0)
        n_like = {"n": 0}

        def fake_randn(*a, **k):
            return noise.clone()

        def fake_like(x, **k):
            n_like["n"] += 1
            return churn[n_like["n"] - 1].clone()
        torch.randn, torch.randn_like = fake_randn, fake_like
        try:
            x, mse, mse_pred, cache = edm_sampler_with_mse(net, cache, conditioning=torch.full((B, 1), 2), num_steps=4,
                                                           sigma_min=0.01, sigma_max=80, rho=2, **kw)
        finally:
            torch.randn, torch.randn_like = real_randn, real_like
        assert n_like["n"] == (4 if kw["S_churn"] else 0)
        out[tag + "_x"] = x
        out[tag + "_mse"] = np.array(mse, dtype=np.float64)
        out[tag + "_mse_pred"] = np.array(mse_pred, dtype=np.float64)
        out[tag + "_cache_n_ctx"] = np.int64(cache["n_context_frames"])
        out[tag + "_cache_attn_frames"] = np.int64(cache[("enc", "8x8_block0")]["attn"][0].shape[2])
    save("g9b_sampler_branches", **out)


# ------------------------------------------------------------------ G10 import of a 2-D EDM2 net (load_from_2d)
def g10():
    """UNet.load_from_2d (networks_edm2.py:238-258): the reference net starts from parameter set A, imports the 2-D
    stand-in built from set B (tests/golden/twod.py); the fixture is the per-key sum / abs-sum of its state_dict
    afterwards (pure copies, so the sums pin exactly which tensors moved where)."""
    import twod
    out = {}
    for tag, cfg, sa, sb in [("small", SMALL_CFG, 90, 91), ("c1", C1_CFG, 92, 93)]:
        pa, pb = paramgen.unet_params(cfg, sa), paramgen.unet_params(cfg, sb)
        unet = UNet(**cfg)
        unet.load_state_dict({k: v.clone() for k, v in pa.items()}, strict=True)
        unet.load_from_2d(twod.Net2D(pb, list(unet.enc.keys()), list(unet.dec.keys())))
        keys, sums, asums = twod.state_sums(unet.state_dict())
        out[tag + "_keys"], out[tag + "_sums"], out[tag + "_abs"] = np.array(keys), np.array(sums), np.array(asums)
        out[tag + "_seeds"] = np.array([sa, sb])
    save("g10_load2d", **out)


# ------------------------------------------------------------------ G11 power-function EMA coefficients
def g11():
    """edm2/phema.py: std_to_exp and power_function_beta on a grid (known answers for parallel.power_function_*)."""
    from edm2 import phema
    stds = np.array([0.01, 0.05, 0.10, 0.15, 0.20, 0.25])
    tn = np.array([16.0, 1000.0, 123456.0, 5.0e7])
    td = np.array([8.0, 8.0, 64.0, 2048.0])
    exps = np.array([float(phema.std_to_exp(s)) for s in stds])
    betas = np.array([[float(phema.power_function_beta(std=s, t_next=a, t_delta=b)) for a, b in zip(tn, td)] for s in stds])
    save("g11_phema", stds=stds, t_next=tn, t_delta=td, exps=exps, betas=betas)


# ------------------------------------------------------------------ G12 checkpoint in the reference's own file format
CKPT_CFG = dict(img_resolution=16, img_channels=4, label_dim=4, model_channels=8, channel_mult=[1, 2], num_blocks=1,
                video_attn_resolutions=[8], frame_attn_resolutions=[16])     # 8 / 16 channels: no attention heads, ~80 K parameters


def g12():
    """`g12_ckpt_ref.pt`: a {"state_dict", "kwargs"} file WRITTEN BY THE REFERENCE's BetterModule.save_to_state_dict
    (edm2/utils.py:15-34) for a tiny UNet, + `g12_ckpt.npz`: an input and the reference's eval / 2-D-training outputs of the
    model the reference rebuilds from that file with UNet.from_pretrained (:36-64).  save_to_state_dict imports boto3
    unconditionally (:16, only used for s3:// paths): an empty module of that name is registered for the call."""
    import types
    sys.modules.setdefault("boto3", types.ModuleType("boto3"))
    torch.manual_seed(1200)
    unet = UNet(**CKPT_CFG)
    with torch.no_grad():                                   # the zero-initialised gains would hide most of the net
        unet.out_gain.fill_(0.8)
        for m in unet.modules():
            if hasattr(m, "emb_gain"):
                m.emb_gain.fill_(0.3)
            if isinstance(m, Gating):
                m.offset.copy_(torch.randn(2) * 0.3)
                m.max_gating.fill_(1.0)
    path = os.path.join(HERE, "g12_ckpt_ref.pt")
    unet.save_to_state_dict(path)
    print(f"g12_ckpt_ref.pt: {os.path.getsize(path) / 1024:.0f} KiB, kwargs = {unet.kwargs}")
    again = UNet.from_pretrained(path).eval()
    g = torch.Generator().manual_seed(1201)
    B, t = 2, 4
    x = torch.randn(B, t, 4, 16, 16, generator=g)
    c_noise = torch.randn(B, t, generator=g) * 0.5
    lab = torch.randint(0, 4, (B, t), generator=g)
    with torch.no_grad():
        y_eval, cache = again(x, c_noise, lab, update_cache=True)
        x1 = torch.randn(B, 1, 4, 16, 16, generator=g)
        y_next, _ = again(x1, c_noise[:, :1], lab[:, :1], cache=cache)
    again.train()
    with torch.no_grad():
        y_2d, _ = again(x, c_noise, lab, just_2d=True)      # (training mode: forced weight normalisation, conv.py:16-18)
    save("g12_ckpt", x=x, c_noise=c_noise, labels=lab, y_eval=y_eval, x1=x1, y_next=y_next, y_2d=y_2d,
         n_params=np.int64(unet.n_params), keys=np.array(sorted(unet.state_dict().keys())))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g5", "g6", "g6b", "g7", "g8", "g9", "g9b", "g10", "g11", "g12", "g13"]
    for w in which:
        globals()[w]()
