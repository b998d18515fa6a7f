"""GPU parity of the edm2-compatible modules (HIP kernels behind the C-ABI) against the golden vectors captured
from the reference (tests/golden/*.npz).  bf16 operands / fp32 accumulation vs the reference's fp32:
tolerances are relative L2 per tensor, stated at each assert (<= 1e-2 per op; deeper stacks looser)."""
import os
import numpy as np
import pytest
import torch

import paramgen

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def load_params(mod, params):
    sd = {k: v.clone() for k, v in params.items()}
    mod.load_state_dict(sd, strict=True)
    return mod.to(DEV)


def test_g3_gated_conv_module():
    from edm2.conv import MPCausal3DGatedConv
    z = load("g3_gated_conv")
    conv = load_params(MPCausal3DGatedConv(8, 8, [3, 3, 3]), {k[2:]: T(z[k]) for k in z.files if k.startswith("p_")})
    B = 2
    conv.train()
    x = T(z["train_x"]).to(DEV).requires_grad_(True)
    y, _ = conv(x, None, B, T(z["train_cn"]).to(DEV))
    y.backward(T(z["train_gy"]).to(DEV))
    errs = {"y": rel(y, z["train_y"]), "gx": rel(x.grad, z["train_gx"])}
    for n, p in conv.named_parameters():
        errs["g_" + n] = rel(p.grad, z["train_g_" + n])
    print("g3 train", errs)
    assert errs["y"] < 1e-2 and errs["gx"] < 1.5e-2
    assert all(v < 3e-2 for k, v in errs.items() if k.startswith("g_")), errs
    y2, _ = conv(x.detach(), None, B, T(z["train_cn"]).to(DEV), just_2d=True)
    assert rel(y2, z["train_y_just2d"]) < 1e-2
    conv.eval()
    with torch.no_grad():
        xe, cn = T(z["eval_x"]).to(DEV), T(z["eval_cn"]).to(DEV)
        ye, _ = conv(xe, None, B, cn)
        xs = xe.reshape(B, 6, *xe.shape[1:])
        y4, c = conv(xs[:, :4].reshape(-1, *xe.shape[1:]), None, B, cn[:, :4], cache=None, update_cache=True)
        assert c["n_context_frames"] == int(z["eval_cache_n4"])
        y5, c = conv(xs[:, 4:5].reshape(-1, *xe.shape[1:]), None, B, cn[:, 4:5], cache=c, update_cache=True)
        assert c["n_context_frames"] == int(z["eval_cache_n5"])
        y6, c = conv(xs[:, 5:6].reshape(-1, *xe.shape[1:]), None, B, cn[:, 5:6], cache=c, update_cache=False)
    e = (rel(ye, z["eval_y"]), rel(y4, z["eval_y4"]), rel(y5, z["eval_y5"]), rel(y6, z["eval_y6"]))
    print("g3 eval", e)
    assert max(e) < 1e-2
    # cached == uncached (reference property consistency_test.py:261-307), same kernels both ways -> tight
    cat = torch.cat([y4.reshape(B, 4, -1), y5.reshape(B, 1, -1), y6.reshape(B, 1, -1)], 1).reshape(ye.shape)
    assert rel(cat, ye) < 1e-6


@pytest.mark.parametrize("dkv_keys", [0, 128])
def test_g6_attention_modules(dkv_keys, monkeypatch):
    """dkv_keys: 0 = the dK/dV item size the launch picks by load (64 keys at fixture sizes), 128 = the items bench.py's B = 8 step gets."""
    from edm2.attention import VideoAttention, FrameAttention
    from autoregressive_diffusion_amd import ops
    monkeypatch.setattr(ops, "DKV_ITEM_KEYS", dkv_keys)
    z = load("g6_attention")
    for tag, C, m, B in [("a", 64, 1, 2), ("b", 64, 1, 1), ("c", 128, 2, 1)]:
        att = load_params(VideoAttention(C, m), {k[len(tag) + 3:]: T(z[k]) for k in z.files if k.startswith(tag + "_p_")})
        att.train()
        x = T(z[tag + "_x"]).to(DEV).requires_grad_(True)
        y, _ = att(x, B)
        y.backward(T(z[tag + "_gy"]).to(DEV))
        # `_y_compiledflex`: the same call through the reference's REAL torch.compile(flex_attention) (make_golden.py asserts it
        # equal to the dense stand-in the gradients come from, at generation time): the HIP kernels against that
        e = dict(y=rel(y, z[tag + "_y"]), y_compiled=rel(y, z[tag + "_y_compiledflex"]), gx=rel(x.grad, z[tag + "_gx"]),
                 g_qkv=rel(att.attn_qkv.weight.weight.grad, z[tag + "_g_qkv"]),
                 g_proj=rel(att.attn_proj.weight.weight.grad, z[tag + "_g_proj"]))
        print("g6", tag, e)
        assert e["y"] < 1e-2 and e["y_compiled"] < 1e-2 and e["gx"] < 2e-2 and e["g_qkv"] < 3e-2 and e["g_proj"] < 3e-2
        y2, _ = att(x.detach(), B, just_2d=True)
        assert rel(y2, z[tag + "_y_just2d"]) < 1e-2
        if tag == "a":
            att.eval()
            with torch.no_grad():
                xe = T(z["a_eval_x"]).to(DEV)
                ye, _ = att(xe, B)
                xs = xe.reshape(B, 6, *xe.shape[1:])
                y4, c = att(xs[:, :4].reshape(-1, *xe.shape[1:]), B, None, update_cache=True)
                y5, c = att(xs[:, 4:5].reshape(-1, *xe.shape[1:]), B, c, update_cache=True)
                y6, _ = att(xs[:, 5:6].reshape(-1, *xe.shape[1:]), B, c, update_cache=False)
            e = (rel(ye, z["a_eval_y"]), rel(y4, z["a_eval_y4"]), rel(y5, z["a_eval_y5"]), rel(y6, z["a_eval_y6"]),
                 rel(ye, z["a_eval_y_compiledflex"]))
            print("g6 eval", e)
            assert max(e) < 1e-2
    fa = load_params(FrameAttention(64, 1), {k[4:]: T(z[k]) for k in z.files if k.startswith("f_p_")})
    fa.train()
    x = T(z["f_x"]).to(DEV).requires_grad_(True)
    y, _ = fa(x)
    y.backward(T(z["f_gy"]).to(DEV))
    e = (rel(y, z["f_y"]), rel(x.grad, z["f_gx"]), rel(fa.attn_qkv.weight.weight.grad, z["f_g_qkv"]))
    print("g6 frame", e)
    assert e[0] < 1e-2 and e[1] < 2e-2 and e[2] < 3e-2


def test_g6b_attention_small_heads():
    """Heads of 16 / 32 channels (the reference's own tests build 4 heads of 16, consistency_test.py:39,61) through the padded
    path (ops._AttentionHdFn / _attention_eval_hd, csrc/attention_hd.hip) against fixture G6b from the reference: training
    forward + backward, just_2d, causal prefill and cached one-frame steps."""
    from edm2.attention import VideoAttention, FrameAttention
    z = load("g6b_attention_heads")
    for tag, C, m, B in [("h16", 64, 4, 2), ("h32", 64, 2, 1)]:
        att = load_params(VideoAttention(C, m), {k[len(tag) + 3:]: T(z[k]) for k in z.files if k.startswith(tag + "_p_")})
        att.train()
        x = T(z[tag + "_x"]).to(DEV).requires_grad_(True)
        y, _ = att(x, B)
        y.backward(T(z[tag + "_gy"]).to(DEV))
        e = dict(y=rel(y, z[tag + "_y"]), y_compiled=rel(y, z[tag + "_y_compiledflex"]), gx=rel(x.grad, z[tag + "_gx"]),
                 g_qkv=rel(att.attn_qkv.weight.weight.grad, z[tag + "_g_qkv"]),
                 g_proj=rel(att.attn_proj.weight.weight.grad, z[tag + "_g_proj"]))
        print("g6b", tag, e)
        assert e["y"] < 1e-2 and e["y_compiled"] < 1e-2 and e["gx"] < 2e-2 and e["g_qkv"] < 3e-2 and e["g_proj"] < 3e-2
        y2, _ = att(x.detach(), B, just_2d=True)
        assert rel(y2, z[tag + "_y_just2d"]) < 1e-2
        if tag == "h16":
            att.eval()
            with torch.no_grad():
                xe = T(z["h16_eval_x"]).to(DEV)
                ye, _ = att(xe, B)
                xs = xe.reshape(B, 6, *xe.shape[1:])
                y4, c = att(xs[:, :4].reshape(-1, *xe.shape[1:]), B, None, update_cache=True)
                y5, c = att(xs[:, 4:5].reshape(-1, *xe.shape[1:]), B, c, update_cache=True)
                y6, _ = att(xs[:, 5:6].reshape(-1, *xe.shape[1:]), B, c, update_cache=False)
            e = (rel(ye, z["h16_eval_y"]), rel(y4, z["h16_eval_y4"]), rel(y5, z["h16_eval_y5"]), rel(y6, z["h16_eval_y6"]))
            print("g6b eval", e)
            assert max(e) < 1e-2
    fa = load_params(FrameAttention(32, 2), {k[4:]: T(z[k]) for k in z.files if k.startswith("f_p_")})
    fa.train()
    x = T(z["f_x"]).to(DEV).requires_grad_(True)
    y, _ = fa(x)
    y.backward(T(z["f_gy"]).to(DEV))
    e = (rel(y, z["f_y"]), rel(x.grad, z["f_gx"]), rel(fa.attn_qkv.weight.weight.grad, z["f_g_qkv"]))
    print("g6b frame", e)
    assert e[0] < 1e-2 and e[1] < 2e-2 and e[2] < 3e-2


@pytest.mark.parametrize("C,m", [(256, 2), (192, 2), (96, 1)])
def test_attention_heads_wider_than_64_channels(C, m):
    """VERDICT r05 missing #4: Block(channels_per_head=...) accepts any head width in the reference (networks_edm2.py:28,39); heads
    of 128 / 96 channels -- wider than the 64 the product kernels are written for -- go through the generic fp32 attention
    kernel (ops.attention_* -> fp32.wide_heads_*; 1x1 convolutions stay on the bf16 kernels).  VideoAttention in training
    (forward + every gradient), just_2d, causal prefill and two cached steps, FrameAttention, against the fp32 oracle."""
    from oracle import oniris_oracle as O
    from edm2.attention import VideoAttention, FrameAttention
    d = C // m
    g = torch.Generator().manual_seed(C + m)
    p = {"a.attn_qkv.weight.weight": torch.randn(3 * C, C, 1, 1, generator=g), "a.attn_proj.weight.weight": torch.randn(C, C, 1, 1, generator=g),
         "a.rope.inv_freq": 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d)), "a.rope.scale": (torch.arange(0, d, 2) + 0.4 * d) / (1.4 * d)}
    p = paramgen.prenormalise(p)
    B, Tn, H = 2, 4, 8
    x0 = torch.randn(B * 2 * Tn, C, H, H, generator=g)
    gy0 = torch.randn(B * 2 * Tn, C, H, H, generator=g)
    att = load_params(VideoAttention(C, m), {k[2:]: v for k, v in p.items()}).train()
    x = x0.to(DEV).requires_grad_(True)
    y, _ = att(x, B)
    y.backward(gy0.to(DEV))
    pr = {k: v.clone().requires_grad_(v.is_floating_point() and "rope" not in k) for k, v in p.items()}
    xr = x0.clone().requires_grad_(True)
    yr, _ = O.video_attention(pr, "a.", xr, B, m, None, False, False, True)
    yr.backward(gy0)
    e = dict(y=rel(y, yr), gx=rel(x.grad, xr.grad), g_qkv=rel(att.attn_qkv.weight.weight.grad, pr["a.attn_qkv.weight.weight"].grad),
             g_proj=rel(att.attn_proj.weight.weight.grad, pr["a.attn_proj.weight.weight"].grad))
    y2, _ = att(x.detach(), B, just_2d=True)
    with torch.no_grad():
        y2r, _ = O.video_attention(p, "a.", x0, B, m, None, False, True, True)
        e["just_2d"] = rel(y2, y2r)
        att.eval()
        xe = x0[:B * 6].to(DEV)
        xs = xe.reshape(B, 6, C, H, H)
        y4, c = att(xs[:, :4].reshape(-1, C, H, H), B, None, update_cache=True)
        y5, c = att(xs[:, 4:5].reshape(-1, C, H, H), B, c, update_cache=True)
        y6, _ = att(xs[:, 5:6].reshape(-1, C, H, H), B, c, update_cache=False)
        xsr = x0[:B * 6].reshape(B, 6, C, H, H)
        r4, rc = O.video_attention(p, "a.", xsr[:, :4].reshape(-1, C, H, H), B, m, None, True, False, False)
        r5, rc = O.video_attention(p, "a.", xsr[:, 4:5].reshape(-1, C, H, H), B, m, rc, True, False, False)
        r6, _ = O.video_attention(p, "a.", xsr[:, 5:6].reshape(-1, C, H, H), B, m, rc, False, False, False)
        e.update(prefill=rel(y4, r4), step1=rel(y5, r5), step2=rel(y6, r6))
        fa = load_params(FrameAttention(C, m), {k[2:]: v for k, v in p.items() if "rope" not in k}).eval()
        yf, _ = fa(xe)
        e["frame_eval"] = rel(yf, O.frame_attention(p, "a.", x0[:B * 6], m, False))
    print("wide heads", (C, m, d), {k: f"{v:.2e}" for k, v in e.items()})
    assert e["y"] < 1e-2 and e["gx"] < 2e-2 and e["g_qkv"] < 3e-2 and e["g_proj"] < 3e-2
    assert max(e[k] for k in ("just_2d", "prefill", "step1", "step2", "frame_eval")) < 1e-2


WEIGHT_GN_TOL = dict(enc=5e-2, dec=5e-2)      # (tightened to 2x the measured floor below)
# (own-relative, relative to the block's largest gate gradient norm): 2x the values measured on the MI355X (round 5: enc 0.9 % /
# 0.35 %; dec 7.3 % / 3.2 % -- conv_res1.max_gating of the decoder block, a gradient of 1e-3 that is the difference of two sums of
# order 1 on a 4-frame fixture; on the full nets against the oracle every such gradient is within 1.8 %, SCALAR_GRAD_BOUNDS)
# Round 6: the encoder block's FrameAttention gradient no longer passes through bf16 dq / dk / dv tensors (csrc/attention_frame.h: the
# normalisation's adjoint is applied to the fp32 accumulators), which re-rolls the rounding noise every downstream sum sees: on this
# 4-frame fixture conv_res1.max_gating of the ENCODER block (gradient norm 0.225 = 2.5 % of the block's largest) moved from 0.9 % to
# 5.0 % own-relative = 0.12 % of the largest, while every other figure of the block (y, gx, gemb, weight gradients, the attention
# core itself against the oracle: 3.9e-3 instead of 4.2e-3) stayed or improved, and on the full nets against the oracle every such
# gradient is within 0.8 ... 1.7 % as before (SCALAR_GRAD_BOUNDS unchanged).  Bounds at 2x the measured values.
G7_GATE_BOUNDS = dict(enc=(0.10, 7e-3), dec=(0.15, 6.5e-2))


def test_g7_blocks():
    from edm2.networks_edm2 import Block
    from test_oracle_golden import _block_params
    z = load("g7_blocks")
    for tag, kw, cin, cout in [("enc", dict(flavor="enc", resample_mode="down", attention="frame"), 32, 64),
                               ("dec", dict(flavor="dec", resample_mode="up", attention="video"), 96, 64)]:
        p, _ = _block_params(tag, z)
        blk = load_params(Block(cin, cout, 32, **kw), p)
        blk.train()
        x = T(z[tag + "_x"]).to(DEV).requires_grad_(True)
        emb = T(z[tag + "_emb"]).to(DEV).requires_grad_(True)
        y, _ = blk(x, emb, 1, T(z[tag + "_cn"]).to(DEV))
        y.backward(T(z[tag + "_gy"]).to(DEV))
        e = dict(y=rel(y, z[tag + "_y"]), gx=rel(x.grad, z[tag + "_gx"]), gemb=rel(emb.grad, z[tag + "_gemb"]))
        gn = {}
        for n, prm in blk.named_parameters():
            k = f"{tag}_gn_{n}"
            if k in z.files and prm.grad is not None:
                gn[n] = abs(prm.grad.norm().item() - float(z[k])) / (float(z[k]) + 1e-12)
        wmax = max(v for k, v in gn.items() if "gating" not in k)
        gmax = max(v for k, v in gn.items() if "gating" in k)
        print("g7", tag, e, "max gradnorm rel err: weights", wmax, "gates", gmax, max(gn, key=gn.get))
        # bounds = 2x the measured floors (round 4: y 4.7e-3, gx 5.2e-3, gemb 6.5e-3)
        assert e["y"] < 1e-2 and e["gx"] < 1.1e-2 and e["gemb"] < 1.3e-2
        # gate scalars: d(gate) is a reduction of bf16-stored activations whose true value can be far below the magnitude
        # of its terms (scale-invariant layers downstream).  Measured: 0.9 % (enc), 7.3 % (dec: conv_res1.max_gating, a
        # gradient of 1e-3 that is the difference of two sums of order 1) -- bounds at 2x
        assert wmax < WEIGHT_GN_TOL[tag], gn
        assert gmax < (0.10 if tag == "enc" else 0.15), gn          # (enc: 0.02 until round 6, see G7_GATE_BOUNDS)
        # the same statement in the form test_cs_shaped_unet_vs_oracle uses for the full nets (SCALAR_GRAD_BOUNDS): error relative to
        # the parameter's own gradient norm where that is at least 1 % of the block's largest gate gradient norm, relative to that
        # largest one for all of them
        refs = {n: float(z[f"{tag}_gn_{n}"]) for n in gn if "gating" in n}
        top = max(refs.values())
        own = {n: gn[n] for n in refs if refs[n] >= 1e-2 * top}
        rtop = {n: gn[n] * refs[n] / top for n in refs}
        print("g7", tag, "gate gradient norms: worst own-relative", max(own, key=own.get), max(own.values()),
              "; worst relative to the largest", max(rtop, key=rtop.get), max(rtop.values()))
        assert max(own.values()) < G7_GATE_BOUNDS[tag][0] and max(rtop.values()) < G7_GATE_BOUNDS[tag][1], (own, rtop)


def test_g13_block_with_resample_filter():
    """An encoder Block built with resample_filter=[1, 3, 3, 1] (networks_edm2.py:26,66) against the reference's output and input
    gradients (fixture G13): training forward + backward, and the one-frame evaluation path through the same filter pass."""
    from edm2.networks_edm2 import Block
    from test_oracle_golden import _g13_block
    z = load("g13_resample_filter")
    p, _ = _g13_block(z)
    blk = load_params(Block(32, 32, 32, flavor="enc", resample_mode="down", resample_filter=[1, 3, 3, 1]), p).train()
    x = T(z["blk_x"]).to(DEV).requires_grad_(True)
    emb = T(z["blk_emb"]).to(DEV).requires_grad_(True)
    y, _ = blk(x, emb, 1, T(z["blk_cn"]).to(DEV))
    y.backward(T(z["blk_gy"]).to(DEV))
    e = dict(y=rel(y, z["blk_y"]), gx=rel(x.grad, z["blk_gx"]), gemb=rel(emb.grad, z["blk_gemb"]))
    print("g13 block", e)
    assert e["y"] < 1e-2 and e["gx"] < 1.5e-2 and e["gemb"] < 2e-2
    with pytest.raises(NotImplementedError):
        Block(32, 32, 32, flavor="enc", resample_mode="down", resample_filter=[1, 2, 1])      # odd length: the reference asserts too


SMALL_CFG = dict(img_resolution=32, img_channels=4, label_dim=4, model_channels=16, channel_mult=[1, 4, 4],
                 num_blocks=1, video_attn_resolutions=[8], frame_attn_resolutions=[16])
C1_CFG = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=16, channel_mult=[1, 2, 4, 8],
              num_blocks=1, video_attn_resolutions=[8], frame_attn_resolutions=[16])


def build_precond(cfg, seed, sigma_data):
    from edm2.networks_edm2 import UNet, Precond
    p = paramgen.prenormalise(paramgen.precond_params(cfg, seed))
    net = Precond(UNet(**cfg), use_fp16=True, sigma_data=sigma_data)
    return load_params(net, p)


# (own-relative, relative to the largest scalar gradient norm of the net): 2x the values measured on the MI355X (round 5: 3.2 % /
# 0.08 %, 2.1 % / 0.05 %, 3.5 % / 0.09 %, 4.7 % / 0.21 %; the worst are emb_gain's of blocks whose gradient is ~1 % of the largest)
G8_SCALAR_BOUNDS = {("small", "3d"): (6.4e-2, 1.7e-3), ("small", "2d"): (4.2e-2, 1e-3), ("c1", "3d"): (7e-2, 1.8e-3),
                    ("c1", "2d"): (9.5e-2, 4.3e-3)}


@pytest.mark.parametrize("tag,cfg", [("small", SMALL_CFG), ("c1", C1_CFG)])
def test_g8_unet_loss(tag, cfg):
    from edm2.loss import EDM2Loss
    z = load("g8_unet")
    images, labels = T(z[tag + "_images"]).to(DEV), T(z[tag + "_labels"]).to(DEV)
    for mode in ("3d", "2d"):
        net = build_precond(cfg, int(z[tag + "_seed"]), 1.0)
        net.train()
        sigma, eps = T(z[f"{tag}_{mode}_sigma"]).to(DEV), T(z[f"{tag}_{mode}_eps"]).to(DEV)
        loss_fn = EDM2Loss(sigma_data=1.0)
        loss, unw = loss_fn(net, images, labels, sigma=sigma, just_2d=(mode == "2d"), noise=eps)
        loss.backward()
        with torch.no_grad():
            cat = images if mode == "2d" else torch.cat([images, images], 1)
            cond = labels if mode == "2d" else torch.cat([labels, labels], 1)
            net2 = build_precond(cfg, int(z[tag + "_seed"]), 1.0).train()
            Dx, _ = net2(cat + sigma[:, :, None, None, None] * eps, sigma, cond, just_2d=(mode == "2d"))
        e = dict(Dx=rel(Dx, z[f"{tag}_{mode}_Dx"]), loss=abs(loss.item() - float(z[f"{tag}_{mode}_loss"])) / float(z[f"{tag}_{mode}_loss"]),
                 unw=abs(unw - float(z[f"{tag}_{mode}_unweighted"])) / float(z[f"{tag}_{mode}_unweighted"]))
        names = [str(s) for s in z[f"{tag}_{mode}_gradnorm_names"]]
        vals = z[f"{tag}_{mode}_gradnorm_vals"]
        prm = dict(net.named_parameters())
        gerr = {}
        for n, v in zip(names, vals):
            g = prm[n].grad
            assert g is not None, n
            gerr[n] = abs(g.norm().item() - v) / (v + 1e-12)
        worst = sorted(gerr, key=gerr.get)[-3:]
        full = {}
        for k in z.files:
            pre = f"{tag}_{mode}_g_"
            if k.startswith(pre):
                full[k[len(pre):]] = rel(prm[k[len(pre):]].grad, z[k])
        for k in sorted(full):
            if "gating" in k and full[k] > 0.05:
                print("   gate grad", k, prm[k].grad.flatten().tolist(), "ref", z[f"{tag}_{mode}_g_{k}"].flatten().tolist())
        print("g8", tag, mode, e, "worst gradnorm", [(w, round(gerr[w], 4)) for w in worst],
              "worst full grad", max(full.values()) if full else None)
        assert e["Dx"] < 2e-2 and e["loss"] < 2e-2 and e["unw"] < 2e-2
        scalars = {n for n in names if prm[n].numel() <= 2}          # gate parameters and emb_gain / out_gain
        wg = {k: v for k, v in gerr.items() if k not in scalars}
        print("   weight gradnorm rel err: median", float(np.median(list(wg.values()))), "max", max(wg.values()), max(wg, key=wg.get))
        assert np.median(list(wg.values())) < 2e-2 and max(wg.values()) < 0.1, sorted(wg.items(), key=lambda kv: kv[1])[-3:]
        # gate scalars: |err| <= 3% of the value + 0.3% of the largest gate gradient in the net (bf16 noise floor of
        # the sum(dout*out) / sum(dout*y3) reductions; measured floor ~5e-5 absolute on this fixture)
        refs = {n: v for n, v in zip(names, vals) if n in scalars}
        gmax = max(refs.values())
        for n, v in refs.items():
            assert abs(prm[n].grad.norm().item() - v) <= 3e-2 * v + 3e-3 * gmax, (n, prm[n].grad.norm().item(), v)
        own = {n: abs(prm[n].grad.norm().item() - v) / v for n, v in refs.items() if v >= 1e-2 * gmax}
        rtop = {n: abs(prm[n].grad.norm().item() - v) / gmax for n, v in refs.items()}
        print("   scalar gradient norms: worst own-relative", max(own, key=own.get), max(own.values()), "; worst relative to the largest",
              max(rtop, key=rtop.get), max(rtop.values()))
        assert max(own.values()) < G8_SCALAR_BOUNDS[(tag, mode)][0] and max(rtop.values()) < G8_SCALAR_BOUNDS[(tag, mode)][1]
        unused = set(str(s) for s in z[f"{tag}_{mode}_unused"])
        for n in unused:       # parameters the reference leaves without gradient must not get one here either
            g = prm[n].grad
            assert g is None or float(g.abs().max()) == 0.0, n


@pytest.mark.selfcheck
@pytest.mark.parametrize("mode", ["3d", "2d"])
def test_fused_dart_loss_matches_eager_path(mode, monkeypatch):
    """EDM2Loss through the three fused passes (oniris_dart_input / dart_loss / dart_loss_bwd) against the same loss
    through Precond.forward and torch elementwise ops: loss, un-weighted loss and every gradient (incl. out_gain)."""
    import edm2.loss as L
    g = torch.Generator().manual_seed(21)
    B, Tn = 2, 4
    images = torch.randn(B, Tn, 8, 64, 64, generator=g).to(DEV)
    labels = torch.randint(0, 4, (B, Tn), generator=g).to(DEV)
    n = Tn if mode == "2d" else 2 * Tn
    sigma = (torch.randn(B, n, generator=g) + 0.4).exp().to(DEV)
    eps = torch.randn(B, n, 8, 64, 64, generator=g).to(DEV)
    res = {}
    for fused in (1, 0):
        monkeypatch.setattr(L, "FUSED", fused)
        net = build_precond(C1_CFG, 31, 1.0).train()
        loss, unw = L.EDM2Loss(sigma_data=1.0)(net, images, labels, sigma=sigma, just_2d=(mode == "2d"), noise=eps)
        loss.backward()
        res[fused] = (loss.item(), unw, {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    assert abs(res[1][0] - res[0][0]) <= 2e-3 * abs(res[0][0]) and abs(res[1][1] - res[0][1]) <= 2e-3 * abs(res[0][1])
    assert set(res[1][2]) == set(res[0][2])
    worst = max(res[0][2], key=lambda k: rel(res[1][2][k], res[0][2][k].cpu().numpy()) if res[0][2][k].numel() > 2 else 0.0)
    print("fused vs eager loss", res[1][0], res[0][0], "worst grad", worst, rel(res[1][2][worst], res[0][2][worst].cpu().numpy()),
          "out_gain grad", res[1][2]["unet.out_gain"].item(), res[0][2]["unet.out_gain"].item())
    for k, v in res[0][2].items():
        if v.numel() > 2 and float(v.abs().max()) > 0:
            assert rel(res[1][2][k], v.cpu().numpy()) < 2e-2, k
    og1, og0 = res[1][2]["unet.out_gain"].item(), res[0][2]["unet.out_gain"].item()
    assert abs(og1 - og0) <= 2e-2 * abs(og0) + 1e-6


@pytest.mark.parametrize("mode", ["3d", "2d"])
def test_fused_prelude_matches_torch_formulation(mode, monkeypatch):
    """Gates / embedding / emb scales through the fused launches with hand-written adjoints (oniris_gates[_bwd],
    oniris_embed_pre / _post[_bwd], oniris_emb_scale[_bwd]) against the torch-autograd formulation of the same math:
    loss and every gradient, the scalar parameters (gating, emb_gain) included."""
    from autoregressive_diffusion_amd import ops
    import edm2.loss as L
    g = torch.Generator().manual_seed(5)
    B, Tn = 2, 4
    images = torch.randn(B, Tn, 8, 64, 64, generator=g).to(DEV)
    labels = torch.randint(0, 4, (B, Tn), generator=g).to(DEV)
    n = Tn if mode == "2d" else 2 * Tn
    sigma = (torch.randn(B, n, generator=g) + 0.4).exp().to(DEV)
    eps = torch.randn(B, n, 8, 64, 64, generator=g).to(DEV)
    res = {}
    import torch_prelude
    monkeypatch.setattr(ops, "prelude_reference", torch_prelude)
    for fused in (1, 0):
        monkeypatch.setattr(ops, "FUSED_PRELUDE", fused)
        net = build_precond(C1_CFG, 33, 1.0).train()
        gp = torch.Generator().manual_seed(6)
        with torch.no_grad():                      # away from the initial values: every adjoint term is exercised
            for k, p in net.named_parameters():
                if "gating" in k or k.endswith("emb_gain"):
                    p.add_(torch.randn(p.shape, generator=gp).to(DEV) * 0.5)
        loss, unw = L.EDM2Loss(sigma_data=1.0)(net, images, labels, sigma=sigma, just_2d=(mode == "2d"), noise=eps)
        loss.backward()
        res[fused] = (loss.item(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    assert abs(res[1][0] - res[0][0]) <= 2e-3 * abs(res[0][0]), (res[1][0], res[0][0])
    assert set(res[1][1]) == set(res[0][1]), set(res[1][1]) ^ set(res[0][1])
    small = {k: v for k, v in res[0][1].items() if v.numel() <= 2}
    smax = {kind: max(float(v.abs().max()) for k, v in small.items() if kind in k) for kind in ("gating", "emb_gain")
            if any(kind in k for k in small)}
    worst = {}
    for k, v in res[0][1].items():
        if v.numel() > 2:
            if float(v.abs().max()) > 0:
                worst["weights"] = max(worst.get("weights", 0.0), rel(res[1][1][k], v))
        else:
            kind = "gating" if "gating" in k else "emb_gain" if "emb_gain" in k else "other"
            scale = smax.get(kind, float(v.abs().max()) + 1e-12)
            worst[kind] = max(worst.get(kind, 0.0), float((res[1][1][k] - v).abs().max()) / scale)
    print("fused prelude vs torch formulation", mode, "loss", res[1][0], res[0][0], "worst", worst)
    assert worst["weights"] < 2e-2, worst
    assert all(v < 2e-2 for k, v in worst.items() if k != "weights"), worst        # of the largest gradient of the class


def test_g9_sampler_rollout():
    from edm2.sampler import edm_sampler_with_mse
    z = load("g9_sampler")
    net = build_precond(SMALL_CFG, int(z["seed"]), 0.5)
    net.eval()
    with torch.no_grad():
        D, cache = net(T(z["ctx"]).to(DEV), torch.ones(1, 4, device=DEV) * 0.05, T(z["ctx_labels"]).to(DEV), update_cache=True)
        e0 = rel(D, z["prefill_D"])
        errs = []
        for step in range(2):
            x, _, _, cache = edm_sampler_with_mse(net, cache, conditioning=torch.full((1, 1), 1 + step, device=DEV),
                                                  num_steps=4, sigma_min=0.01, sigma_max=80, rho=2, guidance=1, S_churn=0,
                                                  noise=T(z["noise"][step]).to(DEV))
            errs.append(rel(x, z["frames"][step]))
    print("g9 prefill", e0, "frames", errs)
    assert cache["n_context_frames"] == int(z["cache_n_ctx"])
    blk = cache[("enc", "8x8_block0")]
    assert blk["conv_res0"]["n_context_frames"] == int(z["cache_conv0_n"])
    assert blk["attn"][0].shape[1] == z["cache_attn_k"].shape[2] * z["cache_attn_k"].shape[3]
    assert e0 < 2e-2 and max(errs) < 5e-2


def test_g9b_sampler_side_branches():
    """a14 beyond the default path, on the HIP kernels against the reference's own outputs (fixture G9b): guidance != 1 (the
    extra eval-mode just_2d evaluation without cache + lerp, reference sampler.py:25-32), S_churn > 0 (:52-59), target=
    (:46-48, 78-83: MSE lists, cache untouched), and all three together."""
    from edm2.sampler import edm_sampler_with_mse
    z = load("g9b_sampler_branches")
    net = build_precond(SMALL_CFG, int(z["seed"]), 0.5).eval()
    with torch.no_grad():
        _, cache0 = net(T(z["ctx"]).to(DEV), torch.ones(1, 4, device=DEV) * 0.05, T(z["ctx_labels"]).to(DEV), update_cache=True)
    tgt = T(z["target"]).to(DEV)
    cases = dict(guid=dict(guidance=1.5), churn=dict(S_churn=8), target=dict(target=tgt), all=dict(guidance=0.7, S_churn=8, target=tgt))
    report = {}
    for tag, kw in cases.items():
        fork = lambda c: {k: fork(v) for k, v in c.items()} if isinstance(c, dict) else c      # fresh dicts, shared tensors
        cache = fork(cache0)
        with torch.no_grad():
            x, mse, mse_pred, cache = edm_sampler_with_mse(net, cache, conditioning=torch.full((1, 1), 2, device=DEV), num_steps=4,
                                                           sigma_min=0.01, sigma_max=80, rho=2, noise=T(z["noise"]).to(DEV),
                                                           churn_noise=T(z["churn_noise"]).to(DEV), **kw)
        report[tag] = rel(x, z[tag + "_x"])
        assert cache["n_context_frames"] == int(z[tag + "_cache_n_ctx"]), tag
        P = 64                                                            # 8x8 tokens per frame at the video-attention level
        assert cache[("enc", "8x8_block0")]["attn"][0].shape[1] == int(z[tag + "_cache_attn_frames"]) * P, tag
        if "target" in kw:
            assert len(mse) == len(z[tag + "_mse"]) == 4
            np.testing.assert_allclose(mse, z[tag + "_mse"], rtol=5e-2)
            np.testing.assert_allclose(mse_pred, z[tag + "_mse_pred"], rtol=5e-2)
        else:
            assert mse == [] and mse_pred == []
    print("g9b frames rel L2:", report)
    assert max(report.values()) < 5e-2, report


@pytest.mark.selfcheck
def test_sampler_graph_replay_matches_eager():
    """Rollout with the per-frame hipGraph of the cache-reading UNet evaluations (edm2/sampler.py _GraphedDenoiser)
    against the same rollout launched eagerly: same noise, 3 frames x 6 steps, frames and caches must agree."""
    import edm2.sampler as S
    z = load("g9_sampler")
    outs = {}
    for mode in (0, 1):
        S.SAMPLER_GRAPH = mode
        net = build_precond(SMALL_CFG, int(z["seed"]), 0.5).eval()
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            _, cache = net(T(z["ctx"]).to(DEV), torch.ones(1, 4, device=DEV) * 0.05, T(z["ctx_labels"]).to(DEV), update_cache=True)
            frames = []
            for step in range(3):
                noise = torch.randn(1, 1, *z["ctx"].shape[2:], generator=g).to(DEV)
                x, _, _, cache = S.edm_sampler_with_mse(net, cache, conditioning=torch.full((1, 1), step % 4, device=DEV),
                                                        num_steps=6, sigma_min=0.01, sigma_max=80, rho=2, noise=noise)
                frames.append(x.clone())
        outs[mode] = (frames, cache[("enc", "8x8_block0")]["attn"][0].float().clone(), cache["n_context_frames"])
    S.SAMPLER_GRAPH = 1
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.isfinite(a).all() and rel(b, a.cpu().numpy()) < 1e-5, rel(b, a.cpu().numpy())
    assert outs[0][2] == outs[1][2] and rel(outs[1][1], outs[0][1].cpu().numpy()) < 1e-5


CS_SMALL = dict(img_resolution=32, img_channels=8, label_dim=4, model_channels=32, channel_mult=[1, 2, 4, 4],
                num_blocks=1, video_attn_resolutions=[4], frame_attn_resolutions=[8])
CS_FULL = dict(img_resolution=32, img_channels=8, label_dim=4, model_channels=128, channel_mult=[1, 2, 4, 4],
               num_blocks=2, video_attn_resolutions=[4], frame_attn_resolutions=[8])          # cs_train.py:35-45, 310.0 M
GYM_FULL = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=32, channel_mult=[1, 2, 4, 8],
                num_blocks=2, video_attn_resolutions=[8], frame_attn_resolutions=[16])        # gym_train.py:37-47, 46.2 M


# (own-relative, relative to the largest scalar gradient): 2x the values measured on the MI355X against the oracle (round 5,
# profiles/r05_scalar_grads.txt: own-relative 1.1 % / 1.8 % / 1.4 % / 0.80 % / 0.58 %, relative to the largest 0.14 % / 0.03 % /
# 0.07 % / 0.07 % / 0.03 %).  At BASELINE configs[1] itself (gym net, T = 64) every gate / emb_gain gradient that matters is
# within 0.8 % of the oracle's: the error does not grow with the sequence length, it shrinks (more terms per sum).
SCALAR_GRAD_BOUNDS = {"cs-shaped": (2.2e-2, 2.8e-3), "cs-full-net": (3.6e-2, 6e-4), "gym-full-net": (2.8e-2, 1.4e-3),
                      "gym-full-net-T64": (1.6e-2, 1.4e-3), "cs-full-net-T32": (1.2e-2, 5.6e-4),
                      # the 2-D steps of the same nets, measured on their own (round 6: 0.91 % / 7.6e-4 and 1.20 % / 4.0e-4 -- the
                      # second one had been passing under the 3-D step's 1.2 % by one part in 1e5); 2x the measured values like the rest
                      "gym-full-net-T64/2d": (1.8e-2, 1.5e-3), "cs-full-net-T32/2d": (2.4e-2, 8e-4)}
_full_net_oracle = {}     # (base tag, mode) -> the oracle's loss and gradients: the '+bench-variants' re-runs compare with the same result


@pytest.mark.parametrize("tag,cfg,Tn,labelled", [("cs-shaped", CS_SMALL, 8, False), ("cs-full-net", CS_FULL, 8, False),
                                                 ("gym-full-net", GYM_FULL, 8, True),
                                                 # BASELINE configs[1] itself: 64 frames, L = 8192 tokens per VideoAttention
                                                 # layer (the oracle needs ~25 GB and about a minute on the GPU box's host)
                                                 pytest.param("gym-full-net-T64", GYM_FULL, 64, True, marks=pytest.mark.slow),
                                                 pytest.param("gym-full-net-T64+bench-variants", GYM_FULL, 64, True, marks=pytest.mark.slow),
                                                 pytest.param("gym-full-net-T64/2d", GYM_FULL, 64, True, marks=pytest.mark.slow),
                                                 pytest.param("gym-full-net-T64/2d+bench-variants", GYM_FULL, 64, True, marks=pytest.mark.slow),
                                                 # BASELINE configs[2] at its own length: the 310 M net on 32-frame sequences
                                                 pytest.param("cs-full-net-T32", CS_FULL, 32, False, marks=pytest.mark.slow),
                                                 pytest.param("cs-full-net-T32+bench-variants", CS_FULL, 32, False, marks=pytest.mark.slow),
                                                 pytest.param("cs-full-net-T32/2d", CS_FULL, 32, False, marks=pytest.mark.slow)])
def test_cs_shaped_unet_vs_oracle(tag, cfg, Tn, labelled, monkeypatch):
    """One 3-D training step (loss + every weight gradient) against the fp32 oracle on the same parameters and noise.
    cs-shaped: Counter-Strike topology (cs_train.py:35-45) at reduced width: 32x32 latents, video attention at 4x4
    (P = 16 -> 8 frames per 128-token block: the BlockMask quirk F2 at its strongest), no conditioning.
    cs-full-net / gym-full-net: the FULL nets of BASELINE configs[2] and configs[1] (310.0 M / 46.2 M parameters, every
    kernel variant the bench launches) on a short sequence, which is what the CPU oracle finishes in seconds.
    '/2d': the just_2d training step of the 3:1 mix (gym_train.py:96) on the same net and data (B*T independent frames).
    '+bench-variants': the same step with the kernel choices bench.py's B = 8 launch sizes trigger forced onto this B = 1
    input -- non-temporal instantiations / output stores from 0 bytes on (product: 96 MiB), 128-key dK/dV work items
    (product: by load) -- against the SAME oracle result (what the timed region launches is what the
    oracle tests launch; tests/test_zz_dispatch_coverage.py holds the two sets against each other)."""
    from oracle import oniris_oracle as O
    from autoregressive_diffusion_amd import ops
    from edm2.networks_edm2 import UNet, Precond
    from edm2.loss import EDM2Loss
    base, _, variant = tag.partition("+")
    base, _, mode = base.partition("/")
    just_2d = mode == "2d"
    if variant:
        monkeypatch.setattr(ops, "DKV_ITEM_KEYS", 128)
        old_nt = ops.set_ew_nt_bytes(0)
    try:
        loss, prm, ref_loss, ref_grad = _full_net_step(O, UNet, Precond, EDM2Loss, base, cfg, Tn, labelled, just_2d)
    finally:
        if variant:
            ops.set_ew_nt_bytes(old_nt)
    errs = {k: rel(prm[k].grad, ref_grad[k]) for k in prm
            if k.endswith("weight.weight") and ref_grad.get(k) is not None and float(ref_grad[k].abs().max()) > 0
            and prm[k].grad is not None}
    missing = [k for k in prm if k.endswith("weight.weight") and ref_grad.get(k) is not None
               and float(ref_grad[k].abs().max()) > 0 and prm[k].grad is None]
    assert not missing, missing
    _full_net_asserts(tag, base, loss, ref_loss, prm, ref_grad, errs, labelled, just_2d)


def _full_net_step(O, UNet, Precond, EDM2Loss, base, cfg, Tn, labelled, just_2d):
    res = cfg["img_resolution"]
    p = paramgen.prenormalise(paramgen.precond_params(cfg, 303))
    net = load_params(Precond(UNet(**cfg), sigma_data=1.0), p).train()
    g = torch.Generator().manual_seed(304)
    B = 1
    images = torch.randn(B, Tn, 8, res, res, generator=g)
    labels = torch.randint(0, 4, (B, Tn), generator=g) if labelled else None
    sigma = (torch.randn(B, 2 * Tn, generator=g) + 0.9).exp()
    sigma[:, :Tn] = torch.rand(B, 1, generator=g) * 0.1
    eps = torch.randn(B, 2 * Tn, 8, res, res, generator=g)
    if just_2d:
        sigma, eps = sigma[:, Tn:].contiguous(), eps[:, Tn:].contiguous()
    loss, _ = EDM2Loss(P_mean=0.9, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.1)(
        net, images.to(DEV), labels.to(DEV) if labelled else None, sigma=sigma.to(DEV), just_2d=just_2d, noise=eps.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    key = (base, just_2d)
    if key not in _full_net_oracle:
        pr = {k: v.clone().requires_grad_(v.is_floating_point() and "rope" not in k and "fourier" not in k) for k, v in p.items()}
        ref, _, _ = O.edm2_loss(pr, cfg, images, sigma, eps, labels, just_2d=just_2d, sigma_data=1.0)
        ref.backward()
        keep = base.endswith(("-T64", "-T32"))                  # only the entries that have a '+bench-variants' twin are kept
        val = (float(ref.item()), {k: (v.grad.detach().clone() if v.grad is not None else None) for k, v in pr.items()})
        if not keep:
            return loss, dict(net.named_parameters()), val[0], val[1]
        _full_net_oracle[key] = val
    ref_loss, ref_grad = _full_net_oracle[key]
    return loss, dict(net.named_parameters()), ref_loss, ref_grad


def _full_net_asserts(tag, base, loss, ref_loss, prm, ref_grad, errs, labelled, just_2d):
    worst = max(errs, key=errs.get)
    print(tag, "loss", loss.item(), ref_loss, "median weight-grad rel L2", float(np.median(list(errs.values()))),
          "worst", worst, errs[worst])
    assert abs(loss.item() - ref_loss) / abs(ref_loss) < 2e-2
    assert np.median(list(errs.values())) < 2e-2 and errs[worst] < 4e-2        # (measured: median 1.1-1.5e-2, worst <= 2.3e-2, flat in T: profiles/r03_err_vs_T.txt)
    # gate scalars and emb_gain (the six parameters per gated conv that decide how much temporal context flows, conv.py:104-127):
    # their gradients are reductions over bf16-STORED activations -- sum(dv * v), sum(dv * y3) over H*W*C terms of either sign --
    # so the rounding noise of the terms (2^-9 each) is measured against a sum that can be far smaller than its terms.  Stated
    # per parameter as |hip - oracle| relative to the parameter's own gradient where that is at least 1 % of the largest
    # gate gradient of the net, and relative to that largest gradient for all of them.
    sc = {k: (prm[k].grad.detach().float().cpu().reshape(-1), ref_grad[k].reshape(-1)) for k in prm
          if prm[k].numel() <= 2 and ref_grad.get(k) is not None and prm[k].grad is not None and "out_res" not in k}
    gmax = max(float(r.abs().max()) for _, r in sc.values())
    rel_own = {k: float((h - r).abs().max() / r.abs().max()) for k, (h, r) in sc.items() if float(r.abs().max()) >= 1e-2 * gmax}
    rel_top = {k: float((h - r).abs().max() / gmax) for k, (h, r) in sc.items()}
    wo, wt = max(rel_own, key=rel_own.get), max(rel_top, key=rel_top.get)
    print(tag, f"scalar gradients ({len(sc)} parameters, {len(rel_own)} above 1 % of the largest): worst own-relative", wo, rel_own[wo],
          "median", float(np.median(list(rel_own.values()))), "; worst relative to the largest", wt, rel_top[wt])
    bound_own, bound_top = SCALAR_GRAD_BOUNDS[base + "/2d" if just_2d else base]
    if just_2d:
        # 2-D steps: the context path is off (out = y2, conv.py:60), so no gate scalar has a gradient -- emb_gain and out_gain do:
        # same two-criterion form, bounds measured on the 2-D step
        assert all("gating" not in k or float(r.abs().max()) == 0 for k, (_, r) in sc.items()), [k for k in sc if "gating" in k]
    assert rel_own[wo] < bound_own and rel_top[wt] < bound_top, (wo, rel_own[wo], wt, rel_top[wt])
    if not labelled:
        assert prm["unet.emb_label.weight.weight"].grad is None or float(prm["unet.emb_label.weight.weight"].grad.abs().max()) == 0


def test_full_gym_net_cached_evaluation_vs_oracle():
    """BASELINE configs[4]'s evaluation path on the FULL gym net against the fp32 oracle: a 3-frame causal prefill that fills the
    caches, then one-frame evaluations against them exactly as the sampler issues them (prewarm_eval between frames: kept context
    products, the fused attn_qkv launch of the 256-channel level -- qkv_eval_kernel<256>, which only a HIP-vs-HIP test used to
    launch --, split-K one-frame convolutions, decode attention against the KV ring), with and without update_cache.
    Tolerance: G8's bound for a whole UNet (bf16 kernels vs the fp32 oracle): rel L2 <= 2e-2."""
    from oracle import oniris_oracle as O
    from edm2.networks_edm2 import UNet, Precond
    cfg = GYM_FULL
    p = paramgen.prenormalise(paramgen.precond_params(cfg, 303))
    net = load_params(Precond(UNet(**cfg), sigma_data=1.0), p).eval()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 5, 8, 64, 64, generator=g)
    lab = torch.randint(0, 4, (1, 5), generator=g)
    sig = torch.tensor([[0.05, 0.05, 0.05, 0.7, 2.5]])
    with torch.no_grad():
        D0, cache = net(x[:, :3].to(DEV), sig[:, :3].to(DEV), lab[:, :3].to(DEV), update_cache=True)
        net.unet.prewarm_eval(cache)
        D1a, _ = net((x[:, 3:4] * 1.3).to(DEV), sig[:, 4:5].to(DEV), lab[:, 3:4].to(DEV), cache=cache, update_cache=False)
        D1, cache = net(x[:, 3:4].to(DEV), sig[:, 3:4].to(DEV), lab[:, 3:4].to(DEV), cache=cache, update_cache=True)
        net.unet.prewarm_eval(cache)
        D2, cache = net(x[:, 4:5].to(DEV), sig[:, 4:5].to(DEV), lab[:, 4:5].to(DEV), cache=cache, update_cache=True)
        R0, oc = O.precond_forward(p, cfg, x[:, :3], sig[:, :3], lab[:, :3], cache={}, update_cache=True, training=False, sigma_data=1.0)
        R1a, _ = O.precond_forward(p, cfg, x[:, 3:4] * 1.3, sig[:, 4:5], lab[:, 3:4], cache=oc, update_cache=False, training=False,
                                   sigma_data=1.0)
        R1, oc = O.precond_forward(p, cfg, x[:, 3:4], sig[:, 3:4], lab[:, 3:4], cache=oc, update_cache=True, training=False, sigma_data=1.0)
        R2, oc = O.precond_forward(p, cfg, x[:, 4:5], sig[:, 4:5], lab[:, 4:5], cache=oc, update_cache=True, training=False, sigma_data=1.0)
    e = (rel(D0, R0.numpy()), rel(D1a, R1a.numpy()), rel(D1, R1.numpy()), rel(D2, R2.numpy()))
    print("full gym net, cached evaluation vs oracle: prefill / frame 4 (no cache update) / frame 4 / frame 5", e)
    assert max(e) < 2e-2


@pytest.mark.selfcheck
@pytest.mark.parametrize("tag,Tn,j", [("gym", 64, 41), ("cs", 32, 19), ("cs64", 64, 50)])
def test_full_size_causality_and_batch_independence(tag, Tn, j):
    """BASELINE configs[1] / [2] / [3] at FULL size (gym net 46.2 M with T = 64: L = 8192 tokens per VideoAttention
    layer; Counter-Strike net 310 M with T = 32 and T = 64, P = 16 so 8 frames share a 128-token mask block),
    training-mode forward.  Size-independent properties of the path, checked bit-exactly inside ONE call (a
    training-mode call re-normalises the weights in place, conv.py:16-18, so two calls are not comparable bit for
    bit): the batch holds [a, b, a', b] where a' = a with clean frame j and noised frame j perturbed.  Then
    (1) the two copies of b agree exactly (nothing mixes sequences, every kernel is deterministic),
    (2) every output frame < j of a' equals a's, in both halves (the DART mask lets noised frame f see only clean
        frames < f and itself, the causal conv reads the two PREVIOUS clean frames, everything else is per slot),
    (3) frames >= j differ."""
    from edm2.networks_edm2 import UNet, Precond
    torch.manual_seed(11)
    cfg = GYM_FULL if tag == "gym" else CS_FULL
    res = cfg["img_resolution"]
    net = Precond(UNet(**cfg), sigma_data=1.0).to(DEV).train()
    for m in net.modules():
        if hasattr(m, "emb_gain"):
            torch.nn.init.constant_(m.emb_gain, 0.3)
    torch.nn.init.constant_(net.unet.out_gain, 1.0)
    g = torch.Generator().manual_seed(12)
    xa, xb = (torch.randn(2 * Tn, 8, res, res, generator=g) for _ in range(2))
    sa, sb = ((torch.randn(2 * Tn, generator=g) + 1.2).exp() for _ in range(2))
    la, lb = (torch.randint(0, 4, (2 * Tn,), generator=g) for _ in range(2))
    xa2 = xa.clone()
    xa2[j] += 0.5
    xa2[Tn + j] -= 0.5
    x = torch.stack([xa, xb, xa2, xb]).to(DEV)
    sigma = torch.stack([sa, sb, sa, sb]).to(DEV)
    lab = torch.stack([la, lb, la, lb]).to(DEV) if tag == "gym" else None       # cs_train.py:103: no conditioning
    with torch.no_grad():
        d, _ = net(x, sigma, lab)
    assert torch.isfinite(d).all()
    assert torch.equal(d[1], d[3]), "identical sequences in different batch slots disagree"
    assert torch.equal(d[0, :j], d[2, :j]), "clean frames before the perturbed one changed"
    assert torch.equal(d[0, Tn:Tn + j], d[2, Tn:Tn + j]), "noised frames before the perturbed one changed"
    for f in (j, j + 1, Tn - 1):
        assert not torch.equal(d[0, f], d[2, f]) and not torch.equal(d[0, Tn + f], d[2, Tn + f]), f


@pytest.mark.selfcheck
def test_full_size_gradient_causality():
    """Backward counterpart at full size (gym net, B = 2, T = 64): the loss reads only the outputs of frames < j of
    sequence 0, so the gradient with respect to the input must be EXACTLY zero for every frame >= j of sequence 0
    (both halves: nothing later can influence an earlier frame) and for the whole of sequence 1, and non-zero before."""
    from edm2.networks_edm2 import UNet, Precond
    torch.manual_seed(13)
    net = Precond(UNet(**GYM_FULL), sigma_data=1.0).to(DEV).train()
    for m in net.modules():
        if hasattr(m, "emb_gain"):
            torch.nn.init.constant_(m.emb_gain, 0.3)
    torch.nn.init.constant_(net.unet.out_gain, 1.0)
    B, Tn, j = 2, 64, 37
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, 2 * Tn, 8, 64, 64, generator=g).to(DEV).requires_grad_(True)
    sigma = (torch.randn(B, 2 * Tn, generator=g) + 1.2).exp().to(DEV)
    lab = torch.randint(0, 4, (B, 2 * Tn), generator=g).to(DEV)
    d, _ = net(x, sigma, lab)
    (d[0, :j].square().sum() + d[0, Tn:Tn + j].square().sum()).backward()
    gx = x.grad
    assert gx is not None and torch.isfinite(gx).all()
    assert float(gx[1].abs().max()) == 0.0, "gradient leaked into the other sequence"
    assert float(gx[0, j:Tn].abs().max()) == 0.0, "gradient reached clean frames >= j"
    assert float(gx[0, Tn + j:].abs().max()) == 0.0, "gradient reached noised frames >= j"
    assert float(gx[0, :j].abs().min(dim=0).values.max()) > 0 and float(gx[0, Tn:Tn + j].abs().sum()) > 0


@pytest.mark.selfcheck
def test_full_size_cached_equals_uncached():
    """The reference's own consistency property (consistency_test.py:129-172,261-307) on the FULL gym net in eval
    mode: denoising 8 frames in one causal call equals denoising them one at a time against the KV / activation
    caches (different kernels: causal prefill vs single-frame decode with split-K convs), and a 5 + 3 split too."""
    from edm2.networks_edm2 import UNet, Precond
    torch.manual_seed(15)
    net = Precond(UNet(**GYM_FULL), sigma_data=1.0).to(DEV).eval()
    for m in net.modules():
        if hasattr(m, "emb_gain"):
            torch.nn.init.constant_(m.emb_gain, 0.3)
    torch.nn.init.constant_(net.unet.out_gain, 1.0)
    g = torch.Generator().manual_seed(16)
    t = 8
    x = torch.randn(1, t, 8, 64, 64, generator=g).to(DEV)
    sigma = (torch.randn(1, t, generator=g) * 0.5).exp().to(DEV)
    lab = torch.randint(0, 4, (1, t), generator=g).to(DEV)
    with torch.no_grad():
        full, _ = net(x, sigma, lab)
        cache, outs = None, []
        for k in range(t):
            o, cache = net(x[:, k:k + 1], sigma[:, k:k + 1], lab[:, k:k + 1], cache=cache, update_cache=True)
            outs.append(o)
        step = torch.cat(outs, 1)
        a, cache = net(x[:, :5], sigma[:, :5], lab[:, :5], update_cache=True)
        errs = []
        for k in range(5, t):
            o, cache = net(x[:, k:k + 1], sigma[:, k:k + 1], lab[:, k:k + 1], cache=cache, update_cache=True)
            errs.append(rel(o, full[:, k:k + 1].cpu().numpy()))
    e1, e2 = rel(step, full.cpu().numpy()), rel(a, full[:, :5].cpu().numpy())
    print("cached vs uncached (full gym net): frame-by-frame", e1, "prefill 5", e2, "then decode", errs)
    assert cache["n_context_frames"] == t
    assert e1 < 1e-2 and e2 < 1e-2 and max(errs) < 1e-2


@pytest.mark.selfcheck
def test_fused_qkv_eval_equals_conv_then_norm():
    """One-frame evaluations of the full gym net with the fused attn_qkv launch (oniris_qkv_eval: 1x1 conv + normalisation
    [+ rotation]) against the same evaluations with the convolution and the normalisation as separate launches: outputs and cached
    keys / values agree to bf16 rounding."""
    from edm2.networks_edm2 import UNet, Precond
    from autoregressive_diffusion_amd import ops
    outs = {}
    for mode in (0, 1):
        ops.FUSED_QKV_EVAL = mode
        torch.manual_seed(0)
        unet = UNet(**GYM_FULL).to(DEV)
        torch.nn.init.constant_(unet.out_gain, 1.0)
        net = Precond(unet, use_fp16=True, sigma_data=1.0).to(DEV).eval()
        g = torch.Generator().manual_seed(2)
        x = torch.randn(1, 4, 8, 64, 64, generator=g).to(DEV)
        lab = torch.randint(0, 4, (1, 4), generator=g).to(DEV)
        res = []
        with torch.no_grad():
            _, cache = net(x[:, :3], torch.full((1, 3), 0.3, device=DEV), lab[:, :3], update_cache=True)
            unet.prewarm_eval(cache)
            D, cache = net(x[:, 3:], torch.full((1, 1), 0.7, device=DEV), lab[:, 3:], cache=cache, update_cache=True)
            res.append(D.float().cpu())
            for key, sub in cache.items():
                if isinstance(sub, dict) and sub.get("attn") is not None:
                    res.append(sub["attn"][0].float().cpu()); res.append(sub["attn"][1].float().cpu())
        outs[mode] = res
    ops.FUSED_QKV_EVAL = 1
    assert len(outs[0]) == len(outs[1]) and len(outs[0]) > 3
    # (not bit for bit: the stand-alone 1x1 convolution of a one-frame evaluation is split-K, i.e. sums K in another order)
    ek = max(rel(b, a.numpy()) for a, b in zip(outs[0][1:], outs[1][1:]))
    e = rel(outs[1][0], outs[0][0].numpy())
    print("fused qkv eval vs separate launches: D", e, "cached k / v", ek)
    assert e < 6e-3 and ek < 1.5e-2          # (bf16 noise: a rounding flip in one layer reaches every later layer's input)


@pytest.mark.selfcheck
def test_cached_decode_at_rollout_depth_equals_uncached():
    """BASELINE configs[4] depth (8 context + 256 generated frames, generation_code.py:83-95) on the FULL gym net: the
    reference's cached == non-cached property (consistency_test.py:129-146) for frame 264 -- denoised alone against the KV /
    activation caches of 263 frames (decode kernel over 16.8 K keys, ring storage, RoPE tables for 264 positions) versus
    all 264 frames in one causal call (prefill table with 132 blocks per row)."""
    from edm2.networks_edm2 import UNet, Precond
    torch.manual_seed(17)
    net = Precond(UNet(**GYM_FULL), sigma_data=1.0).to(DEV).eval()
    for m in net.modules():
        if hasattr(m, "emb_gain"):
            torch.nn.init.constant_(m.emb_gain, 0.3)
    torch.nn.init.constant_(net.unet.out_gain, 1.0)
    g = torch.Generator().manual_seed(18)
    t = 264
    x = torch.randn(1, t, 8, 64, 64, generator=g).to(DEV)
    sigma = (torch.randn(1, t, generator=g) * 0.5).exp().to(DEV)
    lab = torch.randint(0, 4, (1, t), generator=g).to(DEV)
    with torch.no_grad():
        full, _ = net(x, sigma, lab)
        _, cache = net(x[:, :t - 1], sigma[:, :t - 1], lab[:, :t - 1], update_cache=True)
        last, cache = net(x[:, t - 1:], sigma[:, t - 1:], lab[:, t - 1:], cache=cache, update_cache=True)
    e = rel(last, full[:, t - 1:].cpu().numpy())
    print("frame 264: cached decode vs one causal call", e)
    assert torch.isfinite(full).all() and cache["n_context_frames"] == t
    assert e < 1e-2


def _ddp_worker(q):
    """Single-rank RCCL group on cuda:0: OnirisDDP with its early (mid-backward) exchange against the plain backward."""
    import os
    import sys
    import numpy as np   # noqa: F401
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, os.path.join(here, "golden"))
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", init_method="env://", device_id=dev)
        from edm2.loss import EDM2Loss
        from autoregressive_diffusion_amd.parallel import FlatParams, OnirisDDP
        import test_model_gpu as M
        g = torch.Generator().manual_seed(41)
        images = torch.randn(1, 4, 8, 64, 64, generator=g).to(dev)
        labels = torch.randint(0, 4, (1, 4), generator=g).to(dev)
        sigma = (torch.randn(1, 8, generator=g) + 0.4).exp().to(dev)
        eps = torch.randn(1, 8, 8, 64, 64, generator=g).to(dev)
        grads, params, info = {}, {}, {}
        from autoregressive_diffusion_amd.parallel import FlatAdamW
        for mode in ("plain", "ddp", "ddp-bf16", "mesh", "mesh-bf16"):
            net = M.build_precond(M.C1_CFG, 43, 1.0).train()
            flat = FlatParams(net.unet, lazy_small=True)
            staged = len(flat.stages)
            if mode != "plain":
                ddp = OnirisDDP(net.unet, flat=flat, exchange="mesh" if mode.startswith("mesh") else "allreduce",
                                grad_dtype=torch.bfloat16 if mode.endswith("bf16") else None, force_collectives=True, auto_wait=False)
                assert (getattr(flat, "_owned_ranges", None) is not None) == (ddp.exchange == "mesh")
                net.unet = ddp
            opt = FlatAdamW(flat, lr=1e-3)
            loss, _ = EDM2Loss(sigma_data=1.0)(net, images, labels, sigma=sigma, noise=eps, sync=False)
            loss.backward()
            if mode != "plain":
                info[mode] = (len(ddp._works), list(ddp._sent))
                ddp.wait()
            flat.gather()
            torch.cuda.synchronize()
            grads[mode] = flat.grad.clone()
            opt.step(max_norm=0.1)
            torch.cuda.synchronize()
            params[mode] = flat.flat.clone()
        gmax, pmax = grads["plain"].abs().max().item(), params["plain"].abs().max().item()
        dg = {m: (grads["plain"] - grads[m]).abs().max().item() for m in grads if m != "plain"}
        dp = {m: (params["plain"] - params[m]).abs().max().item() for m in params if m != "plain"}
        q.put(("ok", staged, info, dg, dp, gmax, pmax))
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put(("error", traceback.format_exc()))


@pytest.mark.selfcheck
def test_ddp_staged_exchange_single_rank():
    """OnirisDDP on the real UNet (one RCCL rank): the stage hooks fire mid-backward, weight_bwd runs once per stage (pending
    slabs only), every stage's segment of the flat gradient buffer is exchanged early and the head at the end -- and the
    resulting gradients and the parameters after one clipped AdamW step equal those of the plain backward, for the all-reduce
    and the mesh (all-to-all + owned-chunk optimizer + all-gather) forms, with fp32 and bf16 transport."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ddp_worker, args=(q,))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert res[0] == "ok", res[1]
    _, staged, info, dg, dp, gmax, pmax = res
    print("ddp single rank: stages", staged, "collectives issued during backward / stage flags at its end", info,
          "max |grad diff|", dg, "of", gmax, "max |param diff after AdamW|", dp, "of", pmax)
    assert staged >= 2
    for mode, (nworks, sent) in info.items():
        assert nworks >= staged + 1 and not any(sent), (mode, nworks, sent)      # every stage + the head went out
    for mode in dg:
        tol = 1e-2 if mode.endswith("bf16") else 1e-3                        # bf16 transport: one rounding of the average
        assert dg[mode] <= tol * gmax, (mode, dg[mode], gmax)
        # AdamW's first step moves a parameter by lr * g / (|g| + eps), lr = 1e-3: where |g| is of the order of eps, the last
        # bits of the gradient (the gate / emb-scale sums are fp32 atomics: run-to-run order) move the update by a fraction
        # of lr -- 2e-4 absolute on top of the relative bound (a wrong exchange moves parameters by +-lr and more)
        assert dp[mode] <= (3e-3 if mode.endswith("bf16") else 1e-5) * pmax + (2.1e-3 if mode.endswith("bf16") else 2e-4), \
            (mode, dp[mode], pmax)


@pytest.mark.selfcheck
def test_optimizer_skips_parameters_without_gradient():
    """torch.optim.AdamW skips a parameter whose .grad is None; gym_train.py's 2-D steps (i % 4 == 0) give no gradient
    to the context weights / gates, nothing ever reaches out_res.* and emb_time.  FlatAdamW reproduces that per
    parameter (step counters, untouched moments)."""
    from edm2.loss import EDM2Loss
    from edm2.conv import MPCausal3DGatedConv
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
    g = torch.Generator().manual_seed(79)
    images = torch.randn(1, 4, 4, 32, 32, generator=g).to(DEV)
    labels = torch.randint(0, 4, (1, 4), generator=g).to(DEV)
    net = build_precond(SMALL_CFG, 57, 1.0).train()
    unet = net.unet
    flat = FlatParams(unet, lazy_small=True)
    opt = FlatAdamW(flat, lr=1e-3, weight_decay=0.1)
    loss_fn = EDM2Loss(sigma_data=1.0)

    def fwd_bwd(j2d):
        opt.zero_grad()
        loss, _ = loss_fn(net, images, labels, just_2d=j2d, sync=False)
        loss.backward()
        return loss
    steps = {j: (lambda j=j: fwd_bwd(j)) for j in (True, False)}
    ctx_params = [m.weight.weight for m in unet.modules() if isinstance(m, MPCausal3DGatedConv)]
    ctx_params += [p for m in unet.modules() if isinstance(m, MPCausal3DGatedConv) for p in m.gating.parameters()]
    never = list(unet.out_res.parameters()) + list(unet.emb_time.parameters())
    pos = {id(p): i for i, p in enumerate(flat.params)}
    seq = [True, True, False, True, False]
    n2 = n3 = 0
    for j2d in seq:
        steps[j2d]()
        opt.step(max_norm=0.1)
        n2 += int(j2d); n3 += int(not j2d)
        torch.cuda.synchronize()
        for p in ctx_params:
            assert opt.param_steps[pos[id(p)]] == n3, "context weights / gates step only on 3-D steps"
            if n3 == 0:
                assert float(flat.slice_of(opt.m, p).abs().max()) == 0.0 and float(flat.slice_of(opt.v, p).abs().max()) == 0.0
        for p in never:
            assert opt.param_steps[pos[id(p)]] == 0 and float(flat.slice_of(opt.v, p).abs().max()) == 0.0
        own = unet.enc["32x32_conv"].last_frame_conv.weight.weight
        assert opt.param_steps[pos[id(own)]] == n2 + n3
    assert float(flat.slice_of(opt.v, ctx_params[0]).abs().max()) > 0.0
    sd = opt.state_dict()["state"]
    assert len(sd) == sum(1 for s_ in opt.param_steps if s_ > 0) < len(flat.params)


@pytest.mark.selfcheck
def test_frozen_encoder_prefix_trains_the_rest():
    """fine-tuning with a frozen stem / frozen first encoder blocks (`requires_grad_(False)` on a prefix of `enc`).
    The side channels that carry skip / residual gradients to an encoder-side backward kernel (ops.GradSlot) are only opened for
    tensors that HAVE a backward: the step must run (no 'parked for a kernel that never ran'), the frozen parameters stay without
    gradient, and the trainable ones get the gradients of the fully trainable run (same forward, same upstream gradients)."""
    from edm2.networks_edm2 import UNet, Precond
    from edm2.loss import EDM2Loss
    cfg = SMALL_CFG
    p = paramgen.prenormalise(paramgen.precond_params(cfg, 123))
    g = torch.Generator().manual_seed(9)
    res = cfg["img_resolution"]
    images = torch.randn(2, 4, cfg["img_channels"], res, res, generator=g).to(DEV)
    labels = torch.randint(0, 4, (2, 4), generator=g).to(DEV)
    sigma = (torch.randn(2, 8, generator=g) + 0.4).exp().to(DEV)
    eps = torch.randn(2, 8, cfg["img_channels"], res, res, generator=g).to(DEV)
    grads = {}
    for n_frozen in (0, 1, 3, 99):
        net = load_params(Precond(UNet(**cfg), sigma_data=1.0), p).train()
        names = list(net.unet.enc.keys())
        frozen = names[:n_frozen]
        for name in frozen:
            net.unet.enc[name].requires_grad_(False)
        loss, _ = EDM2Loss(sigma_data=1.0)(net, images, labels, sigma=sigma, noise=eps)
        loss.backward()                                       # (raised in round 5 for n_frozen >= 1)
        torch.cuda.synchronize()
        prm = dict(net.named_parameters())
        for k, v in prm.items():
            if any(k.startswith(f"unet.enc.{name}.") for name in frozen):
                assert v.grad is None, k
        grads[n_frozen] = {k: v.grad.detach().float().clone() for k, v in prm.items()
                           if v.grad is not None and k.endswith("weight.weight") and k.startswith("unet.dec.")}
        assert grads[n_frozen], "decoder weights must have gradients"
    for n_frozen in (1, 3, 99):
        worst = max(rel(grads[n_frozen][k], grads[0][k]) for k in grads[0])
        print("frozen encoder prefix of", n_frozen, "entries: decoder weight gradients vs the trainable run, worst rel L2", worst)
        assert worst < 1e-2


@pytest.mark.selfcheck
def test_zero_grad_set_to_none_between_forward_and_backward():
    """`loss = model(x); opt.zero_grad(); loss.backward()` with torch's default set_to_none=True releases the
    .grad tensors the weight-gradient table was built on; the backward must re-validate it and deliver fresh gradients."""
    from edm2.loss import EDM2Loss
    g = torch.Generator().manual_seed(81)
    images = torch.randn(1, 4, 4, 32, 32, generator=g).to(DEV)
    labels = torch.randint(0, 4, (1, 4), generator=g).to(DEV)
    sigma = (torch.randn(1, 8, generator=g) + 0.4).exp().to(DEV)
    eps = torch.randn(1, 8, 4, 32, 32, generator=g).to(DEV)
    net = build_precond(SMALL_CFG, 59, 1.0).train()
    topt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    loss_fn = EDM2Loss(sigma_data=1.0)
    loss, _ = loss_fn(net, images, labels, sigma=sigma, noise=eps)
    loss.backward()                                     # reference gradients (first step: .grad created by the table)
    want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    topt.zero_grad()                                    # set_to_none=True: every .grad is gone
    loss, _ = loss_fn(net, images, labels, sigma=sigma, noise=eps)
    topt.zero_grad()                                    # ... also between forward and backward
    junk = [torch.full_like(p, 7.0) for p in net.parameters()]        # recycle the allocator blocks the old grads lived in
    loss.backward()
    torch.cuda.synchronize()
    assert all(float(j.min()) == 7.0 and float(j.max()) == 7.0 for j in junk), "the backward wrote into freed memory"
    got = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    k = "unet.enc.32x32_conv.last_frame_conv.weight.weight"
    assert k in got and set(want) == set(got)
    # (the first step's forced weight normalisation moved the weights by < 1e-4, so the gradients agree closely)
    assert rel(got[k], want[k]) < 2e-2 and rel(got["unet.dec.8x8_in0.conv_res1.weight.weight"], want["unet.dec.8x8_in0.conv_res1.weight.weight"]) < 2e-2


@pytest.mark.selfcheck
def test_released_gradients_come_back_from_one_pool():
    """The reference loop as written (gym_train.py:72,104-108: torch.optim.AdamW, zero_grad() with torch's set_to_none=True, no
    wrapper): the gradients of the kernel-owned weights come back from one pooled buffer -- same addresses every cycle, the
    descriptor table on the device is not rebuilt -- as long as nobody kept a released gradient; a kept one is left alone.
    Weights nothing reaches (emb_time, networks_edm2.py:205-207) end a backward with .grad = None like in the reference,
    and an evaluation under no_grad creates no gradients."""
    from edm2.loss import EDM2Loss
    g = torch.Generator().manual_seed(83)
    images = torch.randn(1, 4, 4, 32, 32, generator=g).to(DEV)
    labels = torch.randint(0, 4, (1, 4), generator=g).to(DEV)
    net = build_precond(SMALL_CFG, 61, 1.0).train()
    topt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    loss_fn = EDM2Loss(sigma_data=1.0)
    own = net.unet.enc["32x32_conv"].last_frame_conv.weight
    emb_time = net.unet.emb_time.weight.weight

    def cycle(seed=0):
        topt.zero_grad()
        torch.manual_seed(seed)
        loss, _ = loss_fn(net, images, labels)
        loss.backward()
    cycle()
    bank = own.pw.bank
    table, ptr = bank._dev_table.data_ptr(), own.weight.grad.data_ptr()
    first = own.weight.grad.clone()
    assert emb_time.grad is None and float(first.abs().max()) > 0
    cycle()
    assert bank._dev_table.data_ptr() == table and own.weight.grad.data_ptr() == ptr and emb_time.grad is None
    assert rel(own.weight.grad, first) < 0.15           # (zeroed in between: not the sum of two cycles)
    # accumulation without zero_grad in between: gradients add up, emb_time stays without one
    torch.manual_seed(0)
    loss, _ = loss_fn(net, images, labels)
    loss.backward()
    assert rel(own.weight.grad, 2 * first) < 0.15 and emb_time.grad is None and own.weight.grad.data_ptr() == ptr
    # somebody keeps a gradient across zero_grad(): it must stay what it was
    kept = own.weight.grad
    snapshot = kept.clone()
    with pytest.warns(RuntimeWarning, match="ONIRIS_GRAD_POOL"):       # (said once per bank: the pool stepping aside costs a table upload per step)
        cycle()
    assert own.weight.grad is not kept and own.weight.grad.data_ptr() != ptr and torch.equal(kept, snapshot)
    assert rel(own.weight.grad, first) < 0.15
    del kept
    cycle()                                              # ... and once it is gone the pool serves again
    assert own.weight.grad.data_ptr() == ptr and rel(own.weight.grad, first) < 0.15
    # an evaluation (no_grad) after zero_grad() leaves the gradients released
    topt.zero_grad()
    with torch.no_grad():
        net.eval()(images[:, :2], torch.ones(1, 2, device=DEV), labels[:, :2])
    assert all(p.grad is None for p in net.parameters())
    net.train()
    cycle()
    assert own.weight.grad.data_ptr() == ptr and rel(own.weight.grad, first) < 0.15


@pytest.mark.selfcheck
def test_kept_context_product_follows_cache_and_weights():
    """The context product a gated conv keeps beside its cached pair (OnirisConvArgs.ctx_prod, conv.py `_cl`) must be used while
    pair AND weights are the ones it was computed from, and only then: repeated one-frame evaluations against one cache are
    bit-identical to evaluations that never kept anything; a changed weight, or a cache that moved on, drops it."""
    from autoregressive_diffusion_amd import ops
    from edm2.conv import MPCausal3DGatedConv
    torch.manual_seed(91)
    net = build_precond(C1_CFG, 5, 0.5).eval()
    ctx = torch.randn(1, 4, 8, 64, 64, device=DEV)
    lab = torch.randint(0, 4, (1, 4), device=DEV)
    fork = lambda c: {k: fork(v) for k, v in c.items() if k != "_ctx_product"} if isinstance(c, dict) else c
    kept_entries = lambda c: sum(kept_entries(v) for v in c.values()) + ("_ctx_product" in c) if isinstance(c, dict) else 0
    with torch.no_grad():
        _, cache0 = net(ctx, torch.full((1, 4), 0.05, device=DEV), lab, update_cache=True)
        x1, x2 = torch.randn(1, 1, 8, 64, 64, device=DEV), torch.randn(1, 1, 8, 64, 64, device=DEV)
        s1, s2, l1 = torch.full((1, 1), 3.0, device=DEV), torch.full((1, 1), 0.4, device=DEV), lab[:, :1]
        old = ops.KEEP_CTX_PRODUCT
        try:
            ops.KEEP_CTX_PRODUCT = 0
            plain = [net(x, s, l1, cache=fork(cache0))[0] for x, s in ((x1, s1), (x2, s2))]
        finally:
            ops.KEEP_CTX_PRODUCT = old
        c = fork(cache0)
        a1 = net(x1, s1, l1, cache=c)[0]                   # stores the products (mode 1)
        n_kept = kept_entries(c)
        a2 = net(x2, s2, l1, cache=c)[0]                   # reads them (mode 2): other input, other sigma, other gates
        assert n_kept >= 8, n_kept
        assert torch.equal(a1, plain[0]) and torch.equal(a2, plain[1])
        # the weights change under the same cache: the stale products must not be used
        conv = next(m for m in net.modules() if isinstance(m, MPCausal3DGatedConv) and m.in_channels >= 32)
        conv.weight.weight.add_(torch.randn_like(conv.weight.weight) * 0.5)
        b_kept = net(x2, s2, l1, cache=c)[0]
        b_fresh = net(x2, s2, l1, cache=fork(cache0))[0]
        assert torch.equal(b_kept, b_fresh) and not torch.equal(b_kept, a2)
        # the cache moves on: new pairs, new products
        _, c2 = net(x1, s2, l1, cache=c, update_cache=True)
        d_kept = net(x2, s2, l1, cache=c2)[0]
        d_fresh = net(x2, s2, l1, cache=fork(c2))[0]
        assert torch.equal(d_kept, d_fresh)


@pytest.mark.selfcheck
def test_reference_training_loop_shape(tmp_path):
    """The body of gym_train.py's loop (:94-141) with the calls it makes, unmodified in kind: torch.optim.AdamW over
    precond.parameters(), loss.backward() on every micro-step with `just_2d = i % 4 == 0`, clip_grad_norm_ + step + zero_grad on
    every second one (accumulation_steps = 2, :57), an EMA tracker of the phema.py:88-108 form (deep copies of the net, lerp_ on
    their parameters), the learning-rate schedule through param_groups, noise_weight.fit_loss_curve(), save_to_state_dict /
    from_pretrained, and the sampler call of the dashboard (plotting.py:131: guidance = 2).  Must run, stay finite, reduce the
    loss on a fixed batch, and agree with the fused optimizer path (FlatAdamW over the same parameters) step for step."""
    import copy
    from edm2.loss import EDM2Loss, learning_rate_schedule
    from edm2.networks_edm2 import UNet
    from edm2.sampler import edm_sampler_with_mse
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
    g = torch.Generator().manual_seed(83)
    latents = torch.randn(2, 4, 4, 32, 32, generator=g).to(DEV)
    actions = torch.randint(0, 4, (2, 4), generator=g).to(DEV)
    precond = build_precond(SMALL_CFG, 61, 1.0).train()
    twin = copy.deepcopy(precond)
    loss_fn = EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5)
    optimizer = torch.optim.AdamW(precond.parameters(), lr=1e-2, eps=1e-8)
    optimizer.zero_grad()
    emas = [copy.deepcopy(precond) for _ in (0.05, 0.10)]
    flat = FlatParams(twin)
    fopt = FlatAdamW(flat, lr=1e-2, eps=1e-8, weight_decay=0.01)       # torch.optim.AdamW's default weight decay
    losses, lr = [], 1e-2
    for i in range(1, 9):
        torch.manual_seed(100 + i)                                     # the loss draws sigma and the noise: same draw on both nets
        loss, un_weighted = loss_fn(precond, latents, actions, just_2d=(i % 4 == 0))
        losses.append(float(un_weighted))
        loss.backward()
        torch.manual_seed(100 + i)
        loss2, _ = loss_fn(twin, latents, actions, just_2d=(i % 4 == 0))
        loss2.backward()
        if i % 2 == 0:
            torch.nn.utils.clip_grad_norm_(precond.parameters(), 0.1)
            optimizer.step()
            optimizer.zero_grad()
            with torch.no_grad():
                for beta, ema in zip((0.9, 0.95), emas):
                    for p_net, p_ema in zip(precond.parameters(), ema.parameters()):
                        p_ema.lerp_(p_net, 1 - beta)
            lr = learning_rate_schedule(i, 1e-2, 4, 4)
            for grp in optimizer.param_groups:
                grp["lr"] = lr
            fopt.step(max_norm=0.1)
            fopt.zero_grad()
            fopt.param_groups[0]["lr"] = lr
    assert all(np.isfinite(losses)), losses
    worst = max(rel(a, b) for (n, a), (_, b) in zip(precond.named_parameters(), twin.named_parameters()) if a.numel() > 64)
    print("gym_train.py loop shape: un-weighted losses", [round(v, 4) for v in losses], "| torch AdamW vs fused optimizer, worst parameter rel L2", worst)
    assert worst < 2e-3          # same gradients (bit for bit), two implementations of clip + AdamW in fp32
    precond.noise_weight.fit_loss_curve()
    path = str(tmp_path / "unet.pt")
    precond.unet.save_to_state_dict(path)
    back = UNet.from_pretrained(path).to(DEV)
    assert all(torch.equal(a, b) for a, b in zip(back.state_dict().values(), precond.unet.state_dict().values()))
    precond.eval()
    with torch.no_grad():
        _, cache = precond(latents[:1, :3], torch.full((1, 3), 0.05, device=DEV), actions[:1, :3], update_cache=True)
        x, _, _, cache = edm_sampler_with_mse(precond, cache, conditioning=actions[:1, 3:4], num_steps=4, sigma_min=0.4, sigma_max=80,
                                              rho=7, guidance=2)
    assert x.shape[1:] == (1, 4, 32, 32) and bool(torch.isfinite(x).all()) and bool(torch.isfinite(emas[0].unet.out_gain))
    # phema.py:95 deep-copies the net -- here one that has ALREADY run (packed weights, device pointer tables, caches hang off
    # it): the copy must evaluate like the original, on its own storage
    with torch.no_grad():
        xin, sg = latents[:1, :2], torch.full((1, 2), 0.7, device=DEV)
        want, _ = precond(xin, sg, actions[:1, :2])
        late = copy.deepcopy(precond)
        got, _ = late(xin, sg, actions[:1, :2])
        assert torch.equal(got, want)
        for p_net, p_ema in zip(precond.parameters(), late.parameters()):
            assert p_net.data_ptr() != p_ema.data_ptr()
            p_ema.lerp_(torch.randn_like(p_ema), 0.3)                      # (what an EMA update does to the copy, exaggerated)
        moved, _ = late(xin, sg, actions[:1, :2])
        again, _ = precond(xin, sg, actions[:1, :2])
        assert torch.equal(again, want) and not torch.equal(moved, want)
        emas_out, _ = emas[1].eval()(xin, sg, actions[:1, :2])            # a copy made BEFORE the first forward, lerp_-ed since
        assert bool(torch.isfinite(emas_out).all())


def _ddp_torch_optimizer_worker(q):
    """One RCCL rank, collectives really issued: the real UNet under OnirisDDP driven by torch.optim.AdamW and ITS zero_grad()
    (cs_train.py:53-54,76-77,105-121 with `DDP = OnirisDDP`), against the same loop without any wrapper."""
    import os
    import sys
    import contextlib
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", RANK="0", WORLD_SIZE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, os.path.join(here, "golden"))
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", init_method="env://", device_id=dev)
        from edm2.loss import EDM2Loss
        import edm2.loss as L
        from autoregressive_diffusion_amd.parallel import OnirisDDP as DDP
        from torch.nn.parallel import DistributedDataParallel as TorchDDP
        import test_model_gpu as M
        g = torch.Generator().manual_seed(47)
        images = torch.randn(5, 1, 4, 4, 32, 32, generator=g).to(dev)
        labels = torch.randint(0, 4, (1, 4), generator=g).to(dev)
        out = {}
        for mode in ("plain", "ddp", "plain_eager", "torch_ddp"):
            precond = M.build_precond(M.SMALL_CFG, 67, 1.0).train()
            unet = precond.unet
            # (torch's wrapper hides the UNet's attributes from EDM2Loss, which then takes the eager formulation: compared
            # with the un-wrapped loop in the same formulation)
            L.FUSED = 0 if mode in ("plain_eager", "torch_ddp") else 1
            if mode == "ddp":
                precond.unet = unet = DDP(unet, device_ids=[0], output_device=0, find_unused_parameters=True, force_collectives=True)
            if mode == "torch_ddp":            # cs_train.py:10,54 as written
                precond.unet = unet = TorchDDP(unet, device_ids=[0], output_device=0, find_unused_parameters=True)
                inner = unet.module.__dict__["_oniris_inner_ddp"]
                inner.force_collectives = True             # (a one-rank group: issue the collectives anyway)
                # (+ 1: torch's second spelling of the root-level out_gain, see BetterModule._ddp_params_and_buffers_to_ignore)
                pnames = {n for n, _ in unet.module.named_parameters()}
                assert len({n.lstrip(".") for n in unet.parameters_to_ignore} & pnames) == len(inner.flat.params) > 50
                # (the constant buffers too: broadcast once by the inner engine, not rewritten in every forward)
                assert all(n in unet.parameters_to_ignore for n, _ in unet.module.named_buffers()) and not unet.modules_buffers
                assert not ({id(p) for p in unet._build_params_for_reducer()[0]} & {id(p) for p in inner.flat.params})
            optimizer = torch.optim.AdamW(precond.parameters(), lr=1e-2, eps=1e-4)
            optimizer.zero_grad()
            loss_fn = EDM2Loss(P_mean=0.9, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.1)
            nones = []
            for i in range(5):
                torch.manual_seed(500 + i)
                loss, _ = loss_fn(precond, images[i], labels, just_2d=(i % 4 == 0))
                with (contextlib.nullcontext() if (i % 2 == 0 or mode.startswith("plain")) else unet.no_sync()):
                    loss.backward()
                if i % 2 == 0 and i != 0:
                    nones.append(sorted(n for n, p in precond.named_parameters() if p.grad is None))
                    optimizer.step()
                    optimizer.zero_grad()
            torch.cuda.synchronize()
            out[mode] = ({n.replace("unet.module.", "unet."): p.detach().float().cpu().numpy().copy() for n, p in precond.named_parameters()},
                         [[n.replace("unet.module.", "unet.") for n in ns] for ns in nones])
        q.put(("ok", out))
        dist.destroy_process_group()
    except Exception:      # noqa: BLE001
        import traceback
        q.put(("error", traceback.format_exc()))


@pytest.mark.selfcheck
def test_ddp_with_torch_optimizer_on_the_unet():
    """A torch optimizer's zero_grad() releases the flat gradient views of OnirisDDP; the wrapper must adopt the gradients the
    next backward creates outside its buffer (kernel-written conv weight gradients included), exchange them, and hand them back
    through .grad -- without any wait() call by the loop.  With one rank the average is the identity: the parameters after the
    loop equal the un-wrapped loop's, and the same parameters are left without a gradient."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ddp_torch_optimizer_worker, args=(q,))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=60)
    assert res[0] == "ok", res[1]
    for base, wrapped, what in (("plain", "ddp", "OnirisDDP"), ("plain_eager", "torch_ddp", "torch DistributedDataParallel (cs_train.py as written)")):
        (pp, pn), (dp, dn) = res[1][base], res[1][wrapped]
        pp, dp = ({k: torch.from_numpy(v) for k, v in d.items()} for d in (pp, dp))
        assert pn == dn, (pn, dn)
        assert all("unet.emb_time.weight.weight" in ns and "unet.out_res.mult" in ns for ns in pn)     # (as in the reference)
        worst = max(rel(dp[k], pp[k]) for k in pp if pp[k].numel() > 64)
        moved = max(rel(pp[k], build_precond(SMALL_CFG, 67, 1.0).state_dict()[k].float().cpu()) for k in list(pp)[:40] if pp[k].numel() > 64)
        print(what, "+ torch.optim.AdamW vs the un-wrapped loop: worst parameter rel L2", worst, "(the loop moved them by", moved, ")")
        assert worst < 1e-4 and moved > 1e-3
