"""GPU parity of every HIP op (called through the C-ABI wrappers) against the fp32 CPU oracle.
Tolerance: bf16 operands / fp32 accumulation vs fp32 oracle -> relative L2 <= 1e-2 forward, 2e-2 gradients
(stated per assert).  Mask tables are bit-exact."""
import math
import numpy as np
import pytest
import torch

from oracle import oniris_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def nhwc(x):   # (N,C,H,W) fp32 -> (N,H,W,C) bf16 on device
    return x.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)


def nchw(x):   # (N,H,W,C) bf16 device -> (N,C,H,W) fp32 cpu
    return x.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def bfr(x):    # round to bf16 and back (so both paths see identical inputs)
    return x.to(torch.bfloat16).float()


def make_bank(params, **kw):
    from autoregressive_diffusion_amd import ops
    bank = ops.WeightBank()
    pws = [bank.add(p, **kw) for p in params]
    return bank, pws


def test_mask_tables_bit_exact():
    from autoregressive_diffusion_amd import ops
    for (T, P) in [(64, 64), (32, 16), (64, 16), (8, 64), (4, 256), (3, 128)]:
        num, idx, blk = ops.train_mask_table(T, P)
        rn, ri, rb = O.train_table(T, P)
        assert np.array_equal(num, rn) and np.array_equal(idx, ri) and blk == rb
    assert ops.train_mask_table(3, 64) is None


@pytest.mark.parametrize("shape", [(24, 16, ()), (40, 24, (1, 1)), (16, 8, (3, 3)), (64, 32, (2, 3, 3)),
                                   (32, 160, (3, 3)), (16, 288, (2, 3, 3)), (8, 640, (3, 3)), (8, 5, (3, 3))])     # several LDS rounds / odd cin
def test_weight_prep_and_bwd(shape):
    from autoregressive_diffusion_amd import ops
    cout, cin, k = shape
    torch.manual_seed(0)
    w0 = torch.randn(cout, cin, *k) * 1.7
    p = torch.nn.Parameter(w0.clone().to(DEV))
    bank, (pw,) = make_bank([p], gain=0.8)
    bank.prepare(training=True)
    w_ref = w0.clone().requires_grad_(True)
    w_eff, w_new = O.weight_effective(w_ref, 0.8, training=True)
    assert rel(p.data, w_new) < 1e-5, "forced normalisation written back"
    taps = pw.taps
    wf = pw.wf.float().cpu().reshape(taps, pw.CoutP, pw.CinP)[:, :cout, :cin]          # [tap][co][ci]
    ref = w_eff.detach().reshape(cout, cin, taps).permute(2, 0, 1)
    assert rel(wf, ref) < 5e-3, "packed forward weight"
    wb = pw.wb.float().cpu().reshape(taps, pw.CoutPb, pw.CinPb)[:, :cin, :cout]        # [tapflip][ci][co]
    per = taps // pw.kt
    refb = w_eff.detach().reshape(cout, cin, pw.kt, per).flip(-1).reshape(cout, cin, taps).permute(2, 1, 0)
    assert rel(wb, refb) < 5e-3, "packed dgrad weight (flipped, transposed)"
    # backward: random packed gradient
    G = bfr(torch.randn(cout, cin, taps))               # (the split-K slabs are bf16: power-of-two multiples are exact)
    dwp = torch.zeros(pw.CoutP, taps, pw.CinP)          # two split-K slabs [co][tap][ci]: 0.5*G + 1.5*G = 2*G
    dwp[:cout, :, :cin] = G.permute(0, 2, 1)
    assert pw.nsplit_cap >= 2 and pw.dwp.dtype == torch.bfloat16
    pw.dwp[:dwp.numel()].copy_(0.5 * dwp.reshape(-1))
    pw.dwp[dwp.numel():2 * dwp.numel()].copy_(bfr(1.5 * dwp.reshape(-1)))
    G = (0.5 * G + bfr(1.5 * G))
    pw.nsplit.fill_(2)
    p.grad.zero_()
    bank.backward()
    (w_eff * G.reshape(w_eff.shape)).sum().backward()
    assert rel(p.grad, w_ref.grad) < 1e-4, "gradient through the normalisation"
    bank.prepare(training=True)
    assert int(pw.nsplit.item()) == 0, "weight_prep invalidates the slabs of the previous step"


def test_weight_perm3():
    from autoregressive_diffusion_amd import ops
    C = 64
    torch.manual_seed(1)
    w0 = torch.randn(3 * C, C, 1, 1)
    p = torch.nn.Parameter(w0.clone().to(DEV))
    bank, (pw,) = make_bank([p], perm3=True)
    bank.prepare(training=False)
    w_eff, _ = O.weight_effective(w0, 1.0, training=False)
    wf = pw.wf.float().cpu().reshape(pw.CoutP, pw.CinP)[:3 * C, :C]
    ref = w_eff.reshape(C, 3, C).permute(1, 0, 2).reshape(3 * C, C)      # rows (m c s) -> (s m c)
    assert rel(wf, ref) < 5e-3


@pytest.mark.usefixtures("nt_policy")
@pytest.mark.parametrize("N,H,cin,cout,k", [(6, 16, 32, 64, 3), (5, 8, 96, 32, 3), (16, 4, 64, 64, 3), (3, 32, 16, 40, 3),
                                            (7, 8, 64, 192, 1), (3, 16, 48, 32, 1), (130, 1, 64, 96, 1), (6, 32, 32, 96, 1),
                                            # 1x1 with >= 8192 positions and Cin % 64 == 0: the LDS-DMA GEMM (conv1x1_glds.h);
                                            # ragged last position tile, Cout not a multiple of the 128-channel tile
                                            (15, 24, 64, 192, 1), (9, 32, 128, 136, 1), (8, 32, 192, 64, 1), (2, 64, 256, 768, 1),
                                            # even frame counts on Cin % 32 == 0: the LDS-DMA kernel without context phases
                                            (4, 8, 32, 128, 3), (2, 32, 32, 32, 3), (10, 16, 64, 96, 3),
                                            # 32 -> <= 32 channels with >= 512 tiles of 8x16 pixels: the plain streaming kernel
                                            # (conv_plain_stream.h; forward and data gradient): segments with a ragged last one,
                                            # an odd frame count, ragged Cout
                                            (20, 64, 32, 32, 3), (70, 32, 32, 8, 3), (33, 64, 32, 24, 3),
                                            # few-tile 1x1 launches (one generated frame of the cached sampler)
                                            (1, 8, 256, 256, 1), (1, 16, 384, 128, 1), (2, 16, 128, 128, 1), (1, 64, 96, 32, 1),
                                            (1, 8, 768, 256, 1), (1, 32, 192, 64, 1),
                                            # ... with a ragged last 32- / 64-position tile (36 and 100 positions per image)
                                            (1, 6, 128, 64, 1), (2, 10, 64, 32, 1), (5, 10, 192, 96, 1)])
def test_conv_plain(N, H, cin, cout, k):
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(2)
    kk = (k, k) if k == 3 else ((1, 1) if H > 1 else ())
    w0 = O.normalize(O.normalize(torch.randn(cout, cin, *kk)))
    p = torch.nn.Parameter(w0.clone().to(DEV))
    bank, (pw,) = make_bank([p])
    bank.prepare(training=True)
    x0 = bfr(torch.randn(N, cin, H, H))
    x = nhwc(x0).requires_grad_(True)
    y = ops.conv(x, pw)
    gy0 = bfr(torch.randn(N, cout, H, H))
    y.backward(nhwc(gy0))
    bank.backward()
    wr = w0.clone().requires_grad_(True)
    xr = x0.clone().requires_grad_(True)
    w_eff, _ = O.weight_effective(wr, 1.0, training=True)
    yr = torch.nn.functional.conv2d(xr, w_eff.reshape(cout, cin, k, k), padding=k // 2)
    (yr * gy0).sum().backward()
    e = (rel(nchw(y), yr), rel(nchw(x.grad), xr.grad), rel(p.grad, wr.grad))
    print("conv_plain", (N, H, cin, cout, k), "rel err y/dx/dw", e)
    assert e[0] < 1e-2 and e[1] < 1e-2 and e[2] < 2e-2


@pytest.mark.usefixtures("nt_policy")
@pytest.mark.parametrize("B,T,H,cin,cout", [(2, 4, 8, 32, 32), (1, 2, 16, 64, 64), (1, 8, 4, 32, 64), (2, 3, 8, 96, 32),
                                            (1, 2, 32, 16, 32),
                                            # shapes the persistent LDS-DMA kernel takes (16x16 tiles, Cin % 32 == 0):
                                            # several tiles per frame / per workgroup, 1..4 channel chunks, ragged Cout
                                            (2, 5, 32, 32, 96), (1, 3, 16, 128, 64), (3, 7, 16, 32, 32),
                                            # 8x8 images: two frames per workgroup tile (odd T: a ragged last tile)
                                            (2, 3, 8, 64, 128), (1, 5, 8, 32, 64),
                                            # BASELINE configs[1]'s own 64x64 level (B = 2 x 64 frames x 2 slots, C = 32): the
                                            # weight gradients are summed from the MAXIMUM number of bf16 split-K slabs here
                                            # (the rounding of every partial sum must not show against the fp32 oracle)
                                            (2, 64, 64, 32, 32),
                                            # the streaming kernels of the 32-channel level (conv_stream.h / conv_wgrad_stream.h):
                                            # ragged Cout, several segments per sequence with a ragged last one, one frame
                                            (2, 5, 32, 32, 8), (1, 19, 16, 32, 32), (4, 1, 16, 32, 24), (1, 64, 32, 32, 32),
                                            # ... and its 8x16-pixel form (B * 4 * 16 tiles of 4x16 pixels > the 512 slabs of a weight)
                                            (9, 2, 64, 32, 24)])
def test_gated_conv_train(B, T, H, cin, cout):
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(3)
    w2 = O.normalize(O.normalize(torch.randn(cout, cin, 3, 3)))
    w3 = O.normalize(O.normalize(torch.randn(cout, cin, 2, 3, 3)))
    p2, p3 = torch.nn.Parameter(w2.clone().to(DEV)), torch.nn.Parameter(w3.clone().to(DEV))
    bank, (pw2, pw3) = make_bank([p2, p3])
    bank.prepare(training=True)
    N = B * 2 * T
    x0 = bfr(torch.randn(N, cin, H, H))
    g0 = torch.rand(N) * 0.6 + 0.05
    gy0 = bfr(torch.randn(N, cout, H, H))
    x = nhwc(x0).requires_grad_(True)
    g = g0.clone().to(DEV).requires_grad_(True)
    y = ops.gated_conv_train(x, g, pw2, pw3, B, T)
    y.backward(nhwc(gy0))
    bank.backward()
    # oracle (fused algebraic form, fp32)
    xr, gr = x0.clone().requires_grad_(True), g0.clone().requires_grad_(True)
    w2r, w3r = w2.clone().requires_grad_(True), w3.clone().requires_grad_(True)
    e2, _ = O.weight_effective(w2r, 1.0, True)
    e3, _ = O.weight_effective(w3r, 1.0, True)
    F = torch.nn.functional
    y2 = F.conv2d(xr, e2, padding=1)
    clean = xr.reshape(B, 2, T, cin, H, H)[:, 0]
    ctx = torch.cat([torch.ones(B, 2, cin, H, H), clean], 1)
    y3 = F.conv2d(ctx[:, 0:T].reshape(B * T, cin, H, H), e3[:, :, 0], padding=1) + \
        F.conv2d(ctx[:, 1:T + 1].reshape(B * T, cin, H, H), e3[:, :, 1], padding=1)
    y3 = y3.reshape(B, 1, T, cout, H, H).expand(B, 2, T, cout, H, H).reshape(N, cout, H, H)
    yr = O.mp_sum(y2, y3, gr)
    (yr * gy0).sum().backward()
    e = dict(y=rel(nchw(y), yr), dx=rel(nchw(x.grad), xr.grad), dw2=rel(p2.grad, w2r.grad), dw3=rel(p3.grad, w3r.grad),
             dg=rel(g.grad, gr.grad))
    print("gated_conv", (B, T, H, cin, cout), e)
    # 2x the measured floors (round 4, all eleven shapes: y / dx / dw 2.0-2.5e-3 = the bf16 rounding of inputs and outputs,
    # including the 64x64-level shape with the maximum number of bf16 split-K slabs; dg 1.9-5.7e-3)
    assert e["y"] < 5e-3 and e["dx"] < 5e-3 and e["dw2"] < 5e-3 and e["dw3"] < 5e-3 and e["dg"] < 1.2e-2


def test_gated_conv_eval_matches_oracle():
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(4)
    B, t, H, cin, cout = 2, 3, 8, 32, 32
    w2 = torch.randn(cout, cin, 3, 3)
    w3 = torch.randn(cout, cin, 2, 3, 3)
    p2, p3 = torch.nn.Parameter(w2.clone().to(DEV)), torch.nn.Parameter(w3.clone().to(DEV))
    bank, (pw2, pw3) = make_bank([p2, p3])
    bank.prepare(training=False)
    x0 = bfr(torch.randn(B * t, cin, H, H))
    cache0 = bfr(torch.randn(B, cin, 2, H, H))
    g0 = torch.rand(B * t) * 0.6 + 0.05
    x = nhwc(x0)
    ctxf = torch.cat([cache0.permute(0, 2, 3, 4, 1).to(DEV, torch.bfloat16), x.reshape(B, t, H, H, cin)], 1).contiguous()
    y = ops.gated_conv_eval(x, g0.to(DEV), pw2, pw3, B, t, ctxf)
    e2, _ = O.weight_effective(w2, 1.0, False)
    e3, _ = O.weight_effective(w3, 1.0, False)
    F = torch.nn.functional
    ctx = torch.cat([cache0.permute(0, 2, 1, 3, 4), x0.reshape(B, t, cin, H, H)], 1)
    y3 = F.conv2d(ctx[:, 0:t].reshape(B * t, cin, H, H), e3[:, :, 0], padding=1) + \
        F.conv2d(ctx[:, 1:t + 1].reshape(B * t, cin, H, H), e3[:, :, 1], padding=1)
    yr = O.mp_sum(F.conv2d(x0, e2, padding=1), y3, g0)
    e = rel(nchw(y), yr)
    print("gated_conv_eval", e)
    assert e < 1e-2


@pytest.mark.parametrize("B,H,cin,cout,epi", [(1, 8, 256, 256, "silu"), (1, 16, 128, 128, "mpsum"), (2, 32, 64, 64, "none"),
                                              (1, 8, 96, 160, "mpsum"), (1, 64, 32, 32, "silu"), (3, 4, 128, 64, "none"),
                                              # more than 128 32-channel tiles: conv_eval1_kernel<32> (fewer run <16>)
                                              (4, 64, 32, 32, "mpsum"), (3, 32, 64, 96, "silu")])
def test_gated_conv_eval_one_frame_splitk(B, H, cin, cout, epi, monkeypatch):
    """One generated frame per sequence (the sampler's shape): the context pair comes straight from the cache tensor
    and the K loop of the few tiles is split over workgroups (OnirisConvArgs.splitk_ws).  Checked against the fp32
    reference for every epilogue, and split-K on/off must agree to the bf16 rounding of the output."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(cin + H)
    w2, w3 = torch.randn(cout, cin, 3, 3), torch.randn(cout, cin, 2, 3, 3)
    p2, p3 = torch.nn.Parameter(w2.clone().to(DEV)), torch.nn.Parameter(w3.clone().to(DEV))
    bank, (pw2, pw3) = make_bank([p2, p3])
    bank.prepare(training=False)
    x0 = bfr(torch.randn(B, cin, H, H))
    cache0 = bfr(torch.randn(B, cin, 2, H, H))
    g0 = torch.rand(B) * 0.6 + 0.05
    x = nhwc(x0)
    pad = cache0.permute(0, 2, 3, 4, 1).to(DEV, torch.bfloat16).contiguous()          # (B, 2, H, H, C)
    kw, F = {}, torch.nn.functional
    if epi == "silu":
        cs0 = torch.rand(B, cout) + 0.5
        kw = dict(cscale=cs0.to(DEV))
    elif epi == "mpsum":
        r0 = bfr(torch.randn(B, cout, H, H))
        kw = dict(res=nhwc(r0), ta=0.7, tb=0.5, clip=2.0)
    outs = {}
    for sk in (1, 0):
        monkeypatch.setattr(ops, "SPLITK", sk)
        outs[sk] = ops.gated_conv_eval(x, g0.to(DEV), pw2, pw3, B, 1, pad, ctx_T=2, **kw).float().cpu()
    e2, _ = O.weight_effective(w2, 1.0, False)
    e3, _ = O.weight_effective(w3, 1.0, False)
    y3 = F.conv2d(cache0[:, :, 0], e3[:, :, 0], padding=1) + F.conv2d(cache0[:, :, 1], e3[:, :, 1], padding=1)
    yr = O.mp_sum(F.conv2d(x0, e2, padding=1), y3, g0)
    if epi == "silu":
        yr = F.silu(yr * cs0[:, :, None, None]) / 0.596
    elif epi == "mpsum":
        yr = (0.7 * r0 + 0.5 * yr).clamp(-2.0, 2.0)
    got = outs[1].permute(0, 3, 1, 2)[:, :cout]
    e = rel(got, yr)
    d = (outs[1] - outs[0]).abs().max().item()
    print("one-frame eval", (B, H, cin, cout, epi), "rel", e, "split vs unsplit max abs", d)
    assert e < 1e-2
    assert d <= 2.0 ** -6 * max(1.0, float(outs[0].abs().max()))          # one bf16 ulp of the largest value


@pytest.mark.selfcheck
@pytest.mark.parametrize("B,H,cin,cout,epi", [(1, 8, 256, 256, "silu"), (2, 16, 128, 128, "mpsum"), (1, 8, 512, 256, "none"),
                                              (8, 8, 256, 256, "mpsum"), (1, 16, 384, 136, "silu")])
def test_one_frame_conv_same_bits_with_16_and_32_channel_workgroups(B, H, cin, cout, epi, monkeypatch):
    """conv_eval1_kernel<16> (few workgroups, long reduction: round 6) against conv_eval1_kernel<32> (big_tile bit 256): the (tap, k-step)
    products are dealt to the waves the same way and the partial tiles are added in the same order, so the same bits -- with the
    context product stored (mode 1) and read back (mode 2) across the two."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(cin + H + B)
    p2 = torch.nn.Parameter(torch.randn(cout, cin, 3, 3).to(DEV)); p3 = torch.nn.Parameter(torch.randn(cout, cin, 2, 3, 3).to(DEV))
    bank, (pw2, pw3) = make_bank([p2, p3])
    bank.prepare(training=False)
    x = nhwc(bfr(torch.randn(B, cin, H, H)))
    pad = bfr(torch.randn(B, cin, 2, H, H)).permute(0, 2, 3, 4, 1).to(DEV, torch.bfloat16).contiguous()
    g = (torch.rand(B) * 0.6 + 0.05).to(DEV)
    kw = {}
    if epi == "silu":
        kw = dict(cscale=(torch.rand(B, cout) + 0.5).to(DEV))
    elif epi == "mpsum":
        kw = dict(res=nhwc(bfr(torch.randn(B, cout, H, H))), ta=0.7, tb=0.5, clip=2.0)
    Co = ops.roundup(cout, 8)
    outs = []
    for bits in (0, 256):
        monkeypatch.setattr(ops, "BIG_TILE", (ops.BIG_TILE & ~256) | bits)
        kept = torch.full((B, H, H, Co), float("nan"), device=DEV)
        plain = ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, **kw)
        stored = ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, ctx_prod=kept, ctx_prod_mode=1, **kw)
        outs.append((plain, stored, kept))
    monkeypatch.setattr(ops, "BIG_TILE", ops.BIG_TILE & ~256)
    read = ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, ctx_prod=outs[1][2], ctx_prod_mode=2, **kw)    # 16-channel launch reads the 32-channel launch's product
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert torch.equal(read, outs[0][0]) and bool(torch.isfinite(read.float()).all())


@pytest.mark.parametrize("B,H,cin,cout,epi", [(1, 8, 256, 256, "silu"), (2, 16, 128, 128, "mpsum"), (3, 32, 64, 64, "none"),
                                              (1, 8, 96, 160, "mpsum"), (2, 64, 32, 32, "silu"), (1, 16, 512, 24, "none"),
                                              (5, 64, 32, 32, "none"), (3, 32, 64, 96, "mpsum")])      # (conv_eval1_kernel<32>)
def test_one_frame_conv_kept_context_product(B, H, cin, cout, epi):
    """OnirisConvArgs.ctx_prod (ABI 11): the context product of the cached pair stored by an all-phases launch (mode 1) or by
    the context-phases-only launch (mode 3), and read back by own-phases-only launches (mode 2) -- all BIT-identical to the
    plain launch, for every epilogue, and with other gate coefficients than the ones the product was stored under (the
    sampler's 31 evaluations per frame differ in exactly that).  y3 itself against the fp32 reference."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(cin + H + 1)
    w2, w3 = torch.randn(cout, cin, 3, 3), torch.randn(cout, cin, 2, 3, 3)
    p2, p3 = torch.nn.Parameter(w2.clone().to(DEV)), torch.nn.Parameter(w3.clone().to(DEV))
    bank, (pw2, pw3) = make_bank([p2, p3])
    bank.prepare(training=False)
    cache0 = bfr(torch.randn(B, cin, 2, H, H))
    pad = cache0.permute(0, 2, 3, 4, 1).to(DEV, torch.bfloat16).contiguous()          # (B, 2, H, H, C)
    assert ops.ctx_product_ok(H, H, cin, cout)
    Co = ops.roundup(cout, 8)
    F = torch.nn.functional
    kept = None
    for it in range(3):                                                               # three "evaluations": new x, new gates
        x0, g0 = bfr(torch.randn(B, cin, H, H)), torch.rand(B) * 0.6 + 0.05
        x, g = nhwc(x0), g0.to(DEV)
        kw = {}
        if epi == "silu":
            cs0 = torch.rand(B, cout) + 0.5
            kw = dict(cscale=cs0.to(DEV))
        elif epi == "mpsum":
            r0 = bfr(torch.randn(B, cout, H, H))
            kw = dict(res=nhwc(r0), ta=0.7, tb=0.5, clip=2.0)
        plain = ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, **kw)
        if kept is None:
            kept = torch.full((B, H, H, Co), float("nan"), device=DEV)
            stored = ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, ctx_prod=kept, ctx_prod_mode=1, **kw)
            assert torch.equal(plain, stored)
            only = ops.gated_conv_ctx_product(pad, pw2, pw3, B)
            assert torch.equal(only, kept) and bool(torch.isfinite(kept).all())
            e3, _ = O.weight_effective(w3, 1.0, False)
            y3 = F.conv2d(cache0[:, :, 0], e3[:, :, 0], padding=1) + F.conv2d(cache0[:, :, 1], e3[:, :, 1], padding=1)
            e = rel(kept.cpu().permute(0, 3, 1, 2)[:, :cout], y3)
            print("kept context product", (B, H, cin, cout), "rel", e)
            assert e < 6e-3                                                           # bf16 operands, fp32 sum, fp32 store
        before = kept.clone()
        read = ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, ctx_prod=kept, ctx_prod_mode=2, **kw)
        assert torch.equal(plain, read), (it, (plain.float() - read.float()).abs().max().item())
        assert torch.equal(before, kept)                                              # mode 2 only reads
        # ... and the own-phases-only launch against the fp32 reference directly (the form every evaluation of the rollout runs)
        e2, _ = O.weight_effective(w2, 1.0, False)
        yr = O.mp_sum(F.conv2d(x0, e2, padding=1), y3, g0)
        if epi == "silu":
            yr = F.silu(yr * cs0[:, :, None, None]) / 0.596
        elif epi == "mpsum":
            yr = (0.7 * r0 + 0.5 * yr).clamp(-2.0, 2.0)
        assert rel(read.float().cpu().permute(0, 3, 1, 2)[:, :cout], yr) < 1e-2
    # a launch the one-frame kernel cannot serve must refuse the mode instead of ignoring it
    x2 = nhwc(bfr(torch.randn(2 * B, cin, H, H)))
    ctx2 = torch.cat([pad, x2.reshape(B, 2, H, H, cin)], 1).contiguous()
    with pytest.raises(RuntimeError):
        ops._conv_launch(x2, ctx2, pw2.wf, pw3.wf, torch.empty(2 * B, H, H, Co, dtype=torch.bfloat16, device=DEV),
                         torch.ones(2 * B, device=DEV), torch.ones(2 * B, device=DEV), B, 1, 2, H, H, cin, pw2.CinP, Co, pw2.CoutP, 9,
                         ctx_bstride=4, ctx_T=4, coff=(0, 1), ctx_prod=kept, ctx_prod_mode=2)


@pytest.mark.usefixtures("nt_policy")
@pytest.mark.parametrize("N,H,cin,cout,clip", [(9, 32, 128, 128, 2.0), (15, 24, 64, 200, 0.0), (33, 16, 256, 256, 3.0)])
def test_conv1x1_mpsum_large(N, H, cin, cout, clip):
    """attn_proj-shaped op on the LDS-DMA GEMM: out = clip(ta*res + tb*(W x)) with >= 8192 positions, + gradients."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(cin + N)
    w0 = O.normalize(O.normalize(torch.randn(cout, cin, 1, 1)))
    p = torch.nn.Parameter(w0.clone().to(DEV))
    bank, (pw,) = make_bank([p])
    bank.prepare(training=True)
    x0, r0 = bfr(torch.randn(N, cin, H, H)), bfr(torch.randn(N, cout, H, H) * 1.5)
    gy0 = bfr(torch.randn(N, cout, H, H))
    x, res = nhwc(x0).requires_grad_(True), nhwc(r0).requires_grad_(True)
    y = ops.conv(x, pw, res=res, ta=0.8, tb=0.6, clip=clip)
    y.backward(nhwc(gy0))
    bank.backward()
    wr, xr, rr = w0.clone().requires_grad_(True), x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
    w_eff, _ = O.weight_effective(wr, 1.0, training=True)
    pre = 0.8 * rr + 0.6 * torch.nn.functional.conv2d(xr, w_eff)
    if clip > 0:
        keep = (nchw(y).abs() < clip).float()              # (mask from the bf16 result, see test_conv_epilogues)
        yr = pre * keep + pre.detach().clamp(-clip, clip) * (1 - keep)
    else:
        yr = pre
    (yr * gy0).sum().backward()
    e = dict(y=rel(nchw(y), yr), dx=rel(nchw(x.grad), xr.grad), dres=rel(nchw(res.grad), rr.grad), dw=rel(p.grad, wr.grad))
    print("conv1x1 mpsum", (N, H, cin, cout, clip), e)
    assert e["y"] < 1e-2 and e["dx"] < 1.5e-2 and e["dres"] < 1e-2 and e["dw"] < 2e-2


def _attn_ref(x0, wq, wp, B, m, training, just_2d=False, rope=True):
    p = {"a.attn_qkv.weight.weight": wq, "a.attn_proj.weight.weight": wp,
         "a.rope.inv_freq": 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)),
         "a.rope.scale": (torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)}
    return p


@pytest.mark.parametrize("persistent,dkv_keys", [(1, 0), (1, 128), (0, 0)])
@pytest.mark.parametrize("B,T,H,m,chunks", [(2, 4, 8, 1, 1), (1, 8, 4, 2, 1), (1, 2, 16, 1, 1), (1, 16, 8, 2, 1),
                                            # dK/dV with every key block's query list split over 3 workgroups
                                            (1, 16, 8, 2, 3), (2, 4, 8, 1, 3),
                                            # 12 (batch, head) pairs (4 XCD groups), Counter-Strike shape (P = 16, 8 heads)
                                            (3, 16, 8, 4, 1), (1, 32, 4, 8, 1)])
def test_video_attention_core_train(B, T, H, m, chunks, persistent, dkv_keys, monkeypatch):
    """qkv -> attention output (the part between the qkv conv and the proj conv), forward + backward; the forward
    through the persistent wave-specialised kernel (attn_fwd_ws_kernel) and through the grid kernel; dK/dV through the
    persistent kernel with the item size the launch picks by load (64 keys at these sizes) AND with the 128-key items
    bench.py's B = 8 step gets (attn_bwd_dkv_ws_kernel<2,128>; ops._dkv_item_keys)."""
    from autoregressive_diffusion_amd import ops
    monkeypatch.setattr(ops, "ATTN_PERSISTENT", persistent)
    monkeypatch.setattr(ops, "DKV_ITEM_KEYS", dkv_keys)
    monkeypatch.setattr(ops, "ATTN_DKV_CHUNKS", chunks)
    monkeypatch.setattr(ops, "ATTN_DKV_MIN_L", 128)
    torch.manual_seed(5)
    C, P, N = 64 * m, H * H, B * 2 * T
    qkv0 = bfr(torch.randn(N, 3 * C, H, H))            # reference channel order (m c s)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    sc = (torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)
    go0 = bfr(torch.randn(N, C, H, H))
    # oracle
    qr_in = qkv0.clone().requires_grad_(True)
    q, k, v = O._split_qkv(qr_in, m)
    q, k, v = (z.reshape(B, 2 * T, m, P, 64).permute(0, 2, 1, 3, 4) for z in (q, k, v))
    q, k = O.rope_apply(q, k, inv, sc, True)
    q, k, v = (z.reshape(B, m, -1, 64) for z in (q, k, v))
    allowed = torch.from_numpy(O.train_allowed_tokens(T, P))
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=allowed)
    o = o.reshape(B, m, 2 * T, P, 64).permute(0, 2, 1, 4, 3).reshape(N, C, H, H)
    (o * go0).sum().backward()
    # HIP: channel order (s m c)
    perm = qkv0.reshape(N, m * 64, 3, H, H).permute(0, 2, 1, 3, 4).reshape(N, 3 * C, H, H)
    x = nhwc(perm).reshape(N, P, 3 * C).requires_grad_(True)
    out = ops.attention_train(x, "video", B, T, m, (inv.to(DEV), sc.to(DEV)))
    out.backward(nhwc(go0).reshape(N, P, C))
    dqkv = x.grad.reshape(N, H, H, 3, m * 64).permute(0, 4, 3, 1, 2).reshape(N, 3 * C, H, H).float().cpu()
    e = (rel(nchw(out.reshape(N, H, H, C)), o), rel(dqkv, qr_in.grad))
    print("video_attention", (B, T, H, m), "rel err out/dqkv", e)
    assert e[0] < 1e-2 and e[1] < 2.5e-2


_c2_oracle = {}


def _c2_attention_oracle(B, T, H, m, seed):
    """Dense `table AND mask_mod` SDPA oracle (O.train_allowed_tokens) forward + backward at a size where the score matrix
    of ONE (sequence, head) pair is L x L fp32 = 268 MB (L = 8192): evaluated pair by pair so the host never holds more than
    a few of them.  Cached: the persistent and the grid forward are checked against the same oracle result."""
    key = (B, T, H, m, seed)
    if key in _c2_oracle:
        return _c2_oracle[key]
    torch.manual_seed(seed)
    C, P, N = 64 * m, H * H, B * 2 * T
    qkv0 = bfr(torch.randn(N, 3 * C, H, H))            # reference channel order (m c s)
    go0 = bfr(torch.randn(N, C, H, H))
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    sc = (torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)
    qr_in = qkv0.clone().requires_grad_(True)
    q, k, v = O._split_qkv(qr_in, m)
    q, k, v = (z.reshape(B, 2 * T, m, P, 64).permute(0, 2, 1, 3, 4) for z in (q, k, v))
    q, k = O.rope_apply(q, k, inv, sc, True)
    q, k, v = (z.reshape(B, m, -1, 64) for z in (q, k, v))
    allowed = torch.from_numpy(O.train_allowed_tokens(T, P))
    go = go0.reshape(B, 2 * T, m, 64, P).permute(0, 2, 1, 4, 3).reshape(B, m, -1, 64)
    o = torch.empty(B, m, 2 * T * P, 64)
    for b in range(B):
        for h in range(m):
            oh = torch.nn.functional.scaled_dot_product_attention(q[b:b + 1, h:h + 1], k[b:b + 1, h:h + 1], v[b:b + 1, h:h + 1],
                                                                  attn_mask=allowed)
            (oh * go[b:b + 1, h:h + 1]).sum().backward(retain_graph=True)
            o[b, h] = oh.detach()[0, 0]
    o = o.reshape(B, m, 2 * T, P, 64).permute(0, 2, 1, 4, 3).reshape(N, C, H, H)
    _c2_oracle[key] = (qkv0, go0, inv, sc, o, qr_in.grad.clone())
    return _c2_oracle[key]


@pytest.mark.parametrize("persistent,dkv_keys", [(1, 0), (1, 128), (0, 0)])
def test_video_attention_bench_shape_vs_dense_oracle(persistent, dkv_keys, monkeypatch):
    """the shape bench.py runs -- BASELINE configs[1]: B = 2, T = 64, P = 64, 4 heads, L = 8192 tokens,
    8 (sequence, head) pairs x 64 query blocks on the longest-first schedule, dK/dV in 4 query chunks (ops defaults, exactly
    what _AttentionFn uses in the step) -- forward AND backward against the dense masked-softmax oracle, through the
    persistent wave-specialised forward (attn_fwd_ws_kernel) and through the grid kernel.
    Tolerance (bf16 operands, fp32 accumulation vs the fp32 oracle): rel L2 <= 1e-2 out, <= 2.5e-2 dqkv."""
    from autoregressive_diffusion_amd import ops
    monkeypatch.setattr(ops, "ATTN_PERSISTENT", persistent)
    monkeypatch.setattr(ops, "DKV_ITEM_KEYS", dkv_keys)        # 0: by load (64-key items at B = 2); 128: what the B = 8 bench step launches
    assert ops.ATTN_DKV_CHUNKS == 4 and ops.ATTN_DKV_MIN_L == 2048          # the bench's own setting: 8192 // 2048 = 4 chunks
    B, T, H, m = 2, 64, 8, 4
    C, P, N = 64 * m, H * H, B * 2 * T
    qkv0, go0, inv, sc, o, dqkv_ref = _c2_attention_oracle(B, T, H, m, 21)
    perm = qkv0.reshape(N, m * 64, 3, H, H).permute(0, 2, 1, 3, 4).reshape(N, 3 * C, H, H)
    x = nhwc(perm).reshape(N, P, 3 * C).requires_grad_(True)
    out = ops.attention_train(x, "video", B, T, m, (inv.to(DEV), sc.to(DEV)))
    out.backward(nhwc(go0).reshape(N, P, C))
    dqkv = x.grad.reshape(N, H, H, 3, m * 64).permute(0, 4, 3, 1, 2).reshape(N, 3 * C, H, H).float().cpu()
    ho = nchw(out.reshape(N, H, H, C))
    e = (rel(ho, o), rel(dqkv, dqkv_ref))
    # per-frame errors too: a wrong softmax weight inside the allowed set of a few rows would hide in the global norm
    fo = ((ho - o).reshape(N, -1).norm(dim=1) / o.reshape(N, -1).norm(dim=1)).max().item()
    fg = ((dqkv - dqkv_ref).reshape(N, -1).norm(dim=1) / (dqkv_ref.reshape(N, -1).norm(dim=1) + 1e-12)).max().item()
    print("video_attention C2 shape", (B, T, H, m), "persistent" if persistent else "grid", f"dkv item keys {dkv_keys or 'by load'}", "rel out/dqkv", e,
          "worst frame out/dqkv", (fo, fg))
    assert e[0] < 1e-2 and e[1] < 2.5e-2
    assert fo < 2e-2 and fg < 5e-2


def test_decode_attention_at_rollout_depth_vs_oracle():
    """BASELINE configs[4] (256-frame rollout: generation_code.py:83-95, attention_modules.py:51-57,69-70): one new frame
    against a KV ring that already holds 8 context + 255 generated = 263 frames (16.8 K keys, gym shape: P = 64, 4 heads),
    i.e. the attention of generated frame 256, compared with the oracle's dense SDPA over the same cache -- including the
    re-rotation of ALL keys with T = 264 (fp16-rounded tables, RoPe.py:21-32,55-57) and the ring growing past a capacity."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(31)
    B, H, m, n_old = 1, 8, 4, 263
    C, P = 64 * m, H * H
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    sc = (torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)
    # the cache holds normalised, un-rotated k and v: unit-RMS vectors per (token, head), bf16-rounded
    kc = bfr(O.normalize(torch.randn(B, m, n_old, P, 64), dim=-1))
    vc = bfr(O.normalize(torch.randn(B, m, n_old, P, 64), dim=-1))
    ring = ops.KVRing(B, P, C, 264, DEV)                   # exactly full after this frame; the NEXT frame must re-home it
    to_ring = lambda z: z.permute(0, 2, 3, 1, 4).reshape(B, n_old * P, C).to(DEV, torch.bfloat16)
    ring.K[:, :n_old * P], ring.V[:, :n_old * P], ring.n = to_ring(kc), to_ring(vc), n_old
    cache = ring.views()
    outs = []
    rk, rv = kc, vc
    rb = (inv.to(DEV), sc.to(DEV))
    for step in range(2):                                  # frame 256 (fills the ring), frame 257 (grows it)
        a = bfr(torch.randn(B, 3 * C, H, H))
        perm = a.reshape(B, m * 64, 3, H, H).permute(0, 2, 1, 3, 4).reshape(B, 3 * C, H, H)
        if step == 0:
            # the sampler's path: the cached keys rotated once for this key count (UNet.prewarm_eval), then ONE launch for
            # the new frame's q / k / v (oniris_qkv_norm_rope_eval) and the decode kernel straight over the ring; the
            # three-launch path without preparation must give the same answer
            o3, _ = ops.attention_eval(nhwc(perm).reshape(B, P, 3 * C), B, m, rb, cache, False, P)
            cache[0]._oniris_ring.rotate_committed(rb)
            assert cache[0]._oniris_ring.kr_state == (n_old, n_old + 1)
            o1, _ = ops.attention_eval(nhwc(perm).reshape(B, P, 3 * C), B, m, rb, cache, False, P)
            assert rel(o1, o3) < 2e-3, rel(o1, o3)
        out, cache = ops.attention_eval(nhwc(perm).reshape(B, P, 3 * C), B, m, rb, cache, True, P)
        q, k, v = O._split_qkv(a, m)
        q, k, v = (z.reshape(B, 1, m, P, 64).permute(0, 2, 1, 3, 4) for z in (q, k, v))
        rk, rv = torch.cat([rk, k], 2), torch.cat([rv, v], 2)
        qq, kk = O.rope_apply(q, rk, inv, sc, False)
        o = torch.nn.functional.scaled_dot_product_attention(qq.reshape(B, m, -1, 64), kk.reshape(B, m, -1, 64),
                                                             rv.reshape(B, m, -1, 64))
        o = o.reshape(B, m, 1, P, 64).permute(0, 2, 1, 4, 3).reshape(B, C, H, H)
        outs.append(rel(nchw(out.reshape(B, H, H, C)), o))
        assert cache[0].shape[1] == (n_old + 1 + step) * P
    print("decode attention at 264 / 265 cached frames: rel", outs)
    assert max(outs) < 1e-2
    assert cache[0]._oniris_ring is not ring and cache[0]._oniris_ring.cap >= 265        # re-homed, old frames copied
    assert rel(cache[0][:, :n_old * P].reshape(B, n_old, P, m, 64).permute(0, 3, 1, 2, 4), kc) == 0.0


@pytest.mark.selfcheck
def test_fused_qkv_norm_rope_matches_three_launch_path(monkeypatch):
    """oniris_qkv_norm_rope[_bwd] (normalisation + both rotations in one pass, one bf16 rounding) against oniris_qkv_norm +
    2 x oniris_rope (+ their adjoints): same attention output and qkv gradient up to the skipped intermediate rounding."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(8)
    B, T, H, m = 2, 8, 8, 2
    C, P, N = 64 * m, H * H, B * 2 * T
    inv = (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).to(DEV)
    sc = ((torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)).to(DEV)
    x0 = torch.randn(N, P, 3 * C, device=DEV).to(torch.bfloat16)
    go = torch.randn(N, P, C, device=DEV).to(torch.bfloat16)
    res = {}
    for fused in (1, 0):
        monkeypatch.setattr(ops, "FUSED_ROPE", fused)
        x = x0.clone().requires_grad_(True)
        out = ops.attention_train(x, "video", B, T, m, (inv, sc))
        out.backward(go)
        res[fused] = (out.detach().float(), x.grad.float())
    e = (rel(res[1][0], res[0][0]), rel(res[1][1], res[0][1]))
    print("fused qkv_norm+rope vs three launches: rel out / dqkv", e)
    assert e[0] < 6e-3 and e[1] < 1e-2


@pytest.mark.parametrize("frame_kernel", [3, 1, 2, 0])
@pytest.mark.parametrize("N,H,m,frame_ws", [(5, 16, 2, 1), (16, 16, 2, 1), (24, 16, 1, 1), (5, 16, 2, 0), (6, 8, 2, 1), (3, 4, 1, 1),
                                            (7, 8, 3, 0), (9, 16, 1, 0), (2, 32, 1, 0)])
def test_frame_attention_core_train(N, H, m, frame_ws, frame_kernel, monkeypatch):
    """FrameAttention's core (dense softmax inside every frame, attention_modules.py:105-119), forward + backward.  Frames of
    128 * 2^k tokens (16x16 latents) run on the persistent work lists of the VideoAttention kernels -- g pseudo-sequences of
    N / g frames under a block-diagonal table, mask_mode 1 (ops.frame_tables; g = 1, 8, 8 for N = 5, 16, 24) --, smaller frames
    and ONIRIS_FRAME_WS=0 on the grid kernels."""
    from autoregressive_diffusion_amd import ops
    monkeypatch.setattr(ops, "FRAME_WS", frame_ws)
    # frame_kernel = 1: frames of 64 / 128 / 256 tokens (8x8, 16x16 latents; odd frame counts leave a partial 256-token super-block)
    # on the whole-frame kernels of csrc/attention_frame.h (forward, dQ, dK / dV); 0: the generic grid kernels
    # 3 (the product): the two launches that read the raw qkv (normalisation and its adjoint inside); 1: qkv_norm passes around the
    # frame forward + one-launch backward; 2: ... around the frame forward, frame dQ and grid dK / dV kernels; 0: grid kernels
    monkeypatch.setattr(ops, "FRAME_KERNEL", min(frame_kernel, 1))
    monkeypatch.setattr(ops, "FRAME_BWD_FUSED", int(frame_kernel in (1, 3)))
    monkeypatch.setattr(ops, "FRAME_QKV_FUSED", int(frame_kernel == 3))
    torch.manual_seed(6)
    C, P = 64 * m, H * H
    qkv0 = bfr(torch.randn(N, 3 * C, H, H))
    go0 = bfr(torch.randn(N, C, H, H))
    qr_in = qkv0.clone().requires_grad_(True)
    q, k, v = O._split_qkv(qr_in, m)
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 1, 3, 2).reshape(N, C, H, H)
    (o * go0).sum().backward()
    perm = qkv0.reshape(N, m * 64, 3, H, H).permute(0, 2, 1, 3, 4).reshape(N, 3 * C, H, H)
    x = nhwc(perm).reshape(N, P, 3 * C).requires_grad_(True)
    out = ops.attention_train(x, "frame", N, 1, m)
    out.backward(nhwc(go0).reshape(N, P, C))
    dqkv = x.grad.reshape(N, H, H, 3, m * 64).permute(0, 4, 3, 1, 2).reshape(N, 3 * C, H, H).float().cpu()
    e = (rel(nchw(out.reshape(N, H, H, C)), o), rel(dqkv, qr_in.grad))
    path = ("work lists" if frame_ws and (H * H) % 128 == 0 else f"frame kernels (variant {frame_kernel})" if frame_kernel and H * H in (64, 128, 256)
            else "grid kernels")
    print("frame_attention", (N, H, m), path, "rel err out/dqkv", e)
    assert e[0] < 1e-2 and e[1] < 2.5e-2


@pytest.mark.parametrize("N,P,m", [(6, 128, 2), (5, 128, 1), (3, 256, 4), (13, 64, 4)])
def test_frame_attention_kernels_on_128_token_frames(N, P, m):
    """The frame kernels (csrc/attention_frame.h) take frames of 64, 128 AND 256 tokens -- a 256-token super-block is 4, 2 or 1
    whole frames.  Square latents only give 64 and 256 (8x8, 16x16); 128 is exercised here directly on (N, P, 3C) token rows,
    with odd frame counts (partial last super-block), against the dense per-frame softmax of the oracle's formulas."""
    from autoregressive_diffusion_amd import ops
    g = torch.Generator().manual_seed(N * P + m)
    C = 64 * m
    x0 = bfr(torch.randn(N, P, 3 * C, generator=g))            # packed attn_qkv order: channel = s * C + head * 64 + c
    go0 = bfr(torch.randn(N, P, C, generator=g))
    xr = x0.clone().requires_grad_(True)
    q, k, v = (O.normalize(xr[:, :, s * C:(s + 1) * C].reshape(N, P, m, 64).permute(0, 2, 1, 3), dim=-1) for s in range(3))
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(N, P, C)
    (o * go0).sum().backward()
    x = x0.to(DEV, torch.bfloat16).requires_grad_(True)
    out = ops.attention_train(x, "frame", N, 1, m)
    out.backward(go0.to(DEV, torch.bfloat16))
    e = (rel(out, o), rel(x.grad, xr.grad))
    print("frame kernels", (N, P, m), "rel err out/dqkv", e)
    assert e[0] < 1e-2 and e[1] < 2.5e-2


@pytest.mark.parametrize("streams", [1, 0], ids=["four-key-streams", "one-key-stream"])
def test_attention_eval_prefill_and_decode(streams, monkeypatch):
    from autoregressive_diffusion_amd import ops
    monkeypatch.setattr(ops, "DECODE_STREAMS", streams)      # decode below the split-KV threshold: attn_fwd_kernel<0, 4> / <0, 1>
    torch.manual_seed(7)
    B, H, m = 2, 8, 1
    C, P = 64 * m, H * H
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    sc = (torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)
    rb = (inv.to(DEV), sc.to(DEV))

    def hip(qkv0, t, cache, upd):
        N = B * t
        perm = qkv0.reshape(N, m * 64, 3, H, H).permute(0, 2, 1, 3, 4).reshape(N, 3 * C, H, H)
        out, cache = ops.attention_eval(nhwc(perm).reshape(N, P, 3 * C), B, m, rb, cache, upd, P)
        return nchw(out.reshape(N, H, H, C)), cache

    def ref(qkv0, t, cache, upd):
        q, k, v = O._split_qkv(qkv0, m)
        q, k, v = (z.reshape(B, t, m, P, 64).permute(0, 2, 1, 3, 4) for z in (q, k, v))
        if cache is not None:
            k, v = torch.cat([cache[0], k], 2), torch.cat([cache[1], v], 2)
        if upd:
            cache = (k, v)
        nk = k.shape[2]
        q, k = O.rope_apply(q, k, inv, sc, False)
        q, k, v = (z.reshape(B, m, -1, 64) for z in (q, k, v))
        allowed = None if t == 1 else torch.from_numpy(O.infer_allowed_tokens(t, P))
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=allowed)
        return o.reshape(B, m, t, P, 64).permute(0, 2, 1, 4, 3).reshape(B * t, C, H, H), cache

    for t0 in (4, 5, 1):      # table path, dense-fallback path (t*P % 128 != 0), score_mod path
        a0 = bfr(torch.randn(B * t0, 3 * C, H, H))
        o1, c1 = hip(a0, t0, None, True)
        r1, rc1 = ref(a0, t0, None, True)
        a1 = bfr(torch.randn(B, 3 * C, H, H))
        o2, c2 = hip(a1, 1, c1, True)
        r2, rc2 = ref(a1, 1, rc1, True)
        e = (rel(o1, r1), rel(o2, r2))
        print("attention_eval t0 =", t0, e)
        assert e[0] < 1e-2 and e[1] < 1e-2
        assert c2[0].shape[1] == (t0 + 1) * P
    # KV ring semantics (ops.KVRing): the cache pair is a view of preallocated storage -- an evaluation that does not
    # update the cache leaves it untouched, and two continuations of ONE cache (two futures) do not see each other
    a0 = bfr(torch.randn(B * 4, 3 * C, H, H))
    _, c4 = hip(a0, 4, None, True)
    k4 = c4[0].float().clone()
    assert getattr(c4[0], "_oniris_ring", None) is not None and c4[0]._oniris_ring.cap >= 5
    fa, fb = bfr(torch.randn(B, 3 * C, H, H)), bfr(torch.randn(B, 3 * C, H, H))
    o_a0, c_same = hip(fa, 1, c4, False)                       # (writes the ring's uncommitted slot)
    assert c_same is c4 and torch.equal(c4[0].float(), k4)
    o_a, ca = hip(fa, 1, c4, True)
    assert torch.equal(o_a, o_a0) and ca[0].data_ptr() == c4[0].data_ptr()            # appended in place
    ka = ca[0].float().clone()
    o_b, cb = hip(fb, 1, c4, True)                             # a second future from the OLD cache: a ring of its own
    assert cb[0].data_ptr() != ca[0].data_ptr()
    assert torch.equal(ca[0].float(), ka) and torch.equal(cb[0][:, :4 * P].float(), k4)
    assert not torch.equal(cb[0][:, 4 * P:].float(), ka[:, 4 * P:])
    r_b, _ = ref(fb, 1, ref(a0, 4, None, True)[1], True)
    assert rel(o_b, r_b) < 1e-2
    c = ca
    for _ in range(14):                                        # grow past the first capacity (16 frames)
        _, c = hip(bfr(torch.randn(B, 3 * C, H, H)), 1, c, True)
    assert c[0].shape[1] == 19 * P and torch.equal(c[0][:, :5 * P].float(), ka)


def test_adamw():
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(8)
    n = 100003
    p0, g0 = torch.randn(n), torch.randn(n)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in (1, 2, 3):
        pr.grad = g0 * step
        opt.step()
        ops.adamw_(p, (g0 * step).to(DEV), m, v, 1e-2, 0.9, 0.99, 1e-8, 0.01, step)
    assert rel(p, pr.data) < 1e-5


@pytest.mark.usefixtures("nt_policy")
@pytest.mark.parametrize("C1,C2,norm", [(64, 0, True), (32, 0, True), (256, 0, True), (64, 32, False), (96, 96, False), (48, 0, False)])
def test_act_fused(C1, C2, norm):
    """[mp_cat | pixel norm] + mp_silu in one pass (utils.py:83-134) and its adjoint, vs the oracle primitives."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(9)
    N, H = 3, 8
    x0 = bfr(torch.randn(N, C1, H, H) * 1.5)
    s0 = bfr(torch.randn(N, C2, H, H)) if C2 else None
    w1, w2 = (0.8, 1.3) if C2 else (1.0, 1.0)
    x = nhwc(x0).requires_grad_(True)
    sk = nhwc(s0).requires_grad_(True) if C2 else None
    res = ops.act(x, sk, w1, w2, norm=norm, want_xo=bool(C2))
    xo, a = res if isinstance(res, tuple) else (None, res)
    ga0 = bfr(torch.randn(N, C1 + C2, H, H))
    gx0 = bfr(torch.randn(N, C1 + C2, H, H))
    tot = (a.float() * nhwc(ga0).float()).sum()
    if xo is not None:
        tot = tot + (xo.float() * nhwc(gx0).float()).sum()
    tot.backward()
    xr = x0.clone().requires_grad_(True)
    sr = s0.clone().requires_grad_(True) if C2 else None
    v = torch.cat([xr * w1, sr * w2], 1) if C2 else xr
    if norm:
        v = O.normalize(v, dim=1)
    ar = O.mp_silu(v)
    totr = (ar * ga0).sum() + ((v * gx0).sum() if xo is not None else 0)
    totr.backward()
    e = [rel(nchw(a), ar), rel(nchw(x.grad), xr.grad)]
    if xo is not None:
        e.append(rel(nchw(xo), v))
    if C2:
        e.append(rel(nchw(sk.grad), sr.grad))
    print("act", (C1, C2, norm), e)
    assert max(e) < 1e-2


@pytest.mark.selfcheck
@pytest.mark.parametrize("mode,norm", [("down", True), ("up", False), ("down", False)])
def test_act_with_resample_equals_resample_then_act(mode, norm):
    """ops.act(..., resample=mode) (one forward launch) against ops.resample followed by ops.act: outputs and the input
    gradient bit for bit, including a second gradient of the input parked in a GradSlot (an encoder output that is also a
    skip connection)."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(12)
    x0 = nhwc(bfr(torch.randn(3, 64, 8, 8) * 1.3))
    extra = nhwc(bfr(torch.randn(3, 64, 8, 8)))
    outs = []
    for fused in (False, True):
        x = x0.clone().requires_grad_(True)
        slot = ops.GradSlot()
        if fused:
            xo, a = ops.act(x, norm=norm, want_xo=True, in_slot=slot, resample=mode)
        else:
            xo, a = ops.act(ops.resample(x, mode, slot), norm=norm, want_xo=True)
        g = torch.Generator().manual_seed(7)
        ga, gx = (bfr(torch.randn(a.shape, generator=g)).to(DEV) for _ in range(2))
        slot.put(extra.clone())
        ((a.float() * ga.float()).sum() + (xo.float() * gx.float()).sum()).backward()
        outs.append((xo.detach().clone(), a.detach().clone(), x.grad.clone()))
        assert slot.g is None
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.usefixtures("nt_policy")
def test_resample_fused():
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(10)
    x0 = bfr(torch.randn(3, 32, 8, 8))
    for mode in ("down", "up"):
        x = nhwc(x0).requires_grad_(True)
        y = ops.resample(x, mode)
        g0 = bfr(torch.randn(nchw(y).shape))
        y.backward(nhwc(g0))
        xr = x0.clone().requires_grad_(True)
        yr = O.resample(xr, mode)
        (yr * g0).sum().backward()
        assert rel(nchw(y), yr) < 5e-3 and rel(nchw(x.grad), xr.grad) < 5e-3


@pytest.mark.selfcheck
@pytest.mark.parametrize("N,H,C1,C2,cout", [(1, 64, 32, 32, 32), (1, 16, 128, 64, 128), (2, 8, 256, 256, 256), (1, 32, 64, 32, 64),
                                            (3, 8, 128, 256, 256)])
def test_conv_cat_act_matches_act_then_conv(N, H, C1, C2, cout):
    """OnirisConvArgs.x2 / act_out (round 5): the head of a decoder Block in an evaluation -- mp_cat, mp_silu and the 1x1 skip
    conv of the concatenation -- as ONE launch, bit-identical to oniris_act_fwd followed by the 1x1 conv (also through the
    split-K pair of launches the one-frame sizes take)."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(N + H + C1)
    p = torch.nn.Parameter(O.normalize(O.normalize(torch.randn(cout, C1 + C2, 1, 1))).to(DEV))
    bank, (pw,) = make_bank([p])
    bank.prepare(training=False)
    x = nhwc(bfr(torch.randn(N, C1, H, H) * 1.5))
    skip = nhwc(bfr(torch.randn(N, C2, H, H)))
    w1, w2 = 0.83, 1.21
    with torch.no_grad():
        assert ops.conv_cat_act_ok(x, skip, pw)
        y, a = ops.conv_cat_act(x, skip, w1, w2, pw)
        xo, a_ref = ops.act(x, skip, w1, w2, want_xo=True)
        y_ref = ops.conv(xo, pw)
    assert torch.equal(a, a_ref), "mp_silu(mp_cat) differs from the activation kernel's"
    assert torch.equal(y, y_ref), "1x1 conv of the concatenation differs from the two-launch form"
    assert float(y.float().abs().mean()) > 0.1


@pytest.mark.parametrize("N,m", [(1, 2), (3, 2), (8, 4), (33, 1), (16, 2)])       # (the last three: qkv_eval_kernel<256 / 64 / 128>, the first two qkv_eval_few_kernel)
def test_frame_attention_eval_query_halves(N, m, monkeypatch):
    """FrameAttention of a few 256-token frames without autograd (the 16x16 level of the cached sampler): two query halves per frame
    (frame_attn_fwd_kernel<1>, round 6) give the bits of the one-workgroup-per-frame launch, and the oracle's SDPA within 1e-2."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(N + m)
    C, H = 64 * m, 16
    p = torch.nn.Parameter(O.normalize(O.normalize(torch.randn(3 * C, C, 1, 1))).to(DEV))
    bank, (pw,) = make_bank([p], perm3=True)                 # (attn_qkv: packed rows (m c s) -> (s m c), attention_modules.py)
    bank.prepare(training=False)
    x0 = bfr(torch.randn(N, C, H, H))
    outs = []
    for halves in (1, 0):
        monkeypatch.setattr(ops, "FRAME_FWD_HALVES", halves)
        with torch.no_grad():
            outs.append(ops.frame_attention_eval(nhwc(x0), pw, m))
    assert torch.equal(outs[0], outs[1])
    w_eff, _ = O.weight_effective(p.detach().float().cpu(), 1.0, False)
    q, k, v = O._split_qkv(torch.nn.functional.conv2d(x0, w_eff), m)                    # (N, m, P, 64), normalised
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(2, 3).reshape(N, C, H, H)
    e = rel(nchw(outs[0].reshape(N, H, H, C)), ref)
    print("frame attention eval", (N, m), "rel", e)
    assert e < 1e-2


@pytest.mark.selfcheck
@pytest.mark.parametrize("N,H,C1,C2,cout", [(1, 8, 256, 256, 256), (1, 16, 256, 128, 128), (1, 32, 64, 32, 64), (8, 8, 256, 256, 256),
                                            (1, 64, 64, 32, 32)])
def test_conv1x1_few_tiles_same_bits_on_both_tile_widths(N, H, C1, C2, cout, monkeypatch):
    """The few-tile 1x1 launches on conv_fwd_kernel (round 6): 32-channel x 32- / 64-position tiles (big_tile bit 512) against the
    64-channel x 128-position tiles of rounds 1-5 (bit 64) -- the K order of an output element is the same, so plain, attn_proj-epilogue
    and two-source launches, whose activation rounds are dealt to the output-channel blocks, give the same bits.  The default,
    conv1x1_few_kernel (four waves split the K of one tile), sums in another order: the activation output is the same bits, the conv
    outputs agree to the bf16 rounding of a sum of four fp32 partials."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(N + H + C1)
    p = torch.nn.Parameter(O.normalize(O.normalize(torch.randn(cout, C1 + C2, 1, 1))).to(DEV))
    bank, (pw,) = make_bank([p])
    bank.prepare(training=False)
    x = nhwc(bfr(torch.randn(N, C1, H, H) * 1.5))
    skip = nhwc(bfr(torch.randn(N, C2, H, H)))
    xc = torch.cat([x, skip], -1).contiguous()
    res = nhwc(bfr(torch.randn(N, cout, H, H)))
    outs = []
    for bits in (512, 64, 0):
        monkeypatch.setattr(ops, "BIG_TILE", (ops.BIG_TILE & ~(64 | 512)) | bits)
        with torch.no_grad():
            y, a = ops.conv_cat_act(x, skip, 0.83, 1.21, pw)
            outs.append((y, a, ops.conv(xc, pw), ops.conv(xc, pw, res=res, ta=0.8, tb=0.6, clip=2.0)))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    assert torch.equal(outs[2][1], outs[0][1])                                   # mp_silu(mp_cat): no sum in it
    for u, v in zip(outs[2], outs[0]):
        d = (u.float() - v.float()).abs().max().item()
        assert d <= 2.0 ** -7 * max(1.0, float(v.float().abs().max())), d          # one bf16 ulp of the largest value
    assert float(outs[0][0].float().abs().mean()) > 0.1


@pytest.mark.parametrize("f", [[1, 1], [1, 3, 3, 1], [1, 2, 3, 3, 2, 1], [2, 5]])
def test_resample_general_filter(f):
    """oniris_resample_filter (Block(resample_filter=...), reference utils.py:94-107) against the oracle's conv2d /
    conv_transpose2d form, forward and adjoint, with a parked second gradient joining in the backward kernel; [1, 1] must
    reproduce the fused 2x2-mean / nearest-x2 kernel bit for bit."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(10 + len(f))
    taps = ops.resample_taps(f)
    assert (taps is None) == (f == [1, 1])
    x0 = bfr(torch.randn(3, 32, 12, 16))
    for mode in ("down", "up"):
        x = nhwc(x0).requires_grad_(True)
        slot = ops.GradSlot()
        y = ops.resample(x, mode, slot, taps)
        g0 = bfr(torch.randn(nchw(y).shape))
        extra = bfr(torch.randn(x0.shape))
        slot.put(nhwc(extra))
        y.backward(nhwc(g0))
        xr = x0.clone().requires_grad_(True)
        yr = O.resample(xr, mode, f)
        (yr * g0).sum().backward()
        e = (rel(nchw(y), yr), rel(nchw(x.grad), xr.grad + extra))
        print("resample", f, mode, e)
        assert max(e) < 5e-3
        if f == [1, 1]:                                       # the general kernel on the [1, 1] taps == the dedicated one
            from autoregressive_diffusion_amd._lib import lib, check
            import ctypes
            out = torch.empty_like(y)
            arr = (ctypes.c_float * 2)(0.5, 0.5)
            N, H, W, C = x.shape
            check(lib.oniris_resample_filter(ops._p(x.detach()), ops._p(out), None, N, H, W, C, 0 if mode == "down" else 1, arr, 2, 1.0,
                                             ops._stream()), "resample_filter")
            assert torch.equal(out, y.detach())
    ops.GradSlot.live = []


@pytest.mark.selfcheck
def test_clip_flags_survive_a_second_forward_before_backward():
    """the forward's "did the clip change anything" flag is read by the backward.  A second grad-enabled forward
    before that backward (two micro-batches summed into one loss; a train-mode evaluation in between) rewinds and refills the
    step's zero arena -- where the flags lived in round 4 -- so the first forward's flag was cleared (mask dropped: gradient
    flows through clipped elements) or aliased with the second forward's.  Now a flag's storage lives with its autograd graph."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(3)
    B, T, cin, cout, H = 1, 4, 32, 64, 16
    N = B * 2 * T
    p2 = torch.nn.Parameter(O.normalize(O.normalize(torch.randn(cout, cin, 3, 3))).to(DEV))
    p3 = torch.nn.Parameter(O.normalize(O.normalize(torch.randn(cout, cin, 2, 3, 3))).to(DEV))
    bank, (pw2, pw3) = make_bank([p2, p3])
    xa0, xb0 = bfr(torch.randn(N, cin, H, H)), bfr(torch.randn(N, cin, H, H))
    ra0, rb0 = bfr(torch.randn(N, cout, H, H) * 200), bfr(torch.randn(N, cout, H, H))       # A clips, B does not
    g0, gy0 = torch.rand(N) * 0.5 + 0.05, bfr(torch.randn(N, cout, H, H))

    def run(interleave):
        bank.prepare(training=True)
        xa, ra = nhwc(xa0).requires_grad_(True), nhwc(ra0).requires_grad_(True)
        ya = ops.gated_conv_train(xa, g0.to(DEV), pw2, pw3, B, T, res=ra, ta=0.9, tb=0.4, clip=256.0, grad_private=True)
        if interleave:
            bank.prepare(training=True)                     # what the second forward of a net does first
            xb, rb = nhwc(xb0).requires_grad_(True), nhwc(rb0).requires_grad_(True)
            yb = ops.gated_conv_train(xb, g0.to(DEV), pw2, pw3, B, T, res=rb, ta=0.9, tb=0.4, clip=256.0, grad_private=True)
        ya.backward(nhwc(gy0).clone())
        if interleave:
            yb.backward(nhwc(gy0).clone())
        torch.cuda.synchronize()
        return xa.grad.clone(), ra.grad.clone(), float((ya.abs() >= 256).float().mean())
    dx1, dr1, frac = run(False)
    dx2, dr2, _ = run(True)
    assert frac > 0.01, "clip not exercised"
    assert float((dr1 == 0).float().mean()) > 0.01, "the mask of the clipped elements is missing in the single-forward run"
    assert torch.equal(dr1, dr2) and torch.equal(dx1, dx2), "a second forward changed the first one's backward"


# H = 16: the LDS-DMA tile kernel's epilogues (cout = 64) and the streaming kernel's (cout = 32); clipped = False: the +-256 clip
# is armed but never reached -- the usual case, in which the backward pre-pass reads no mask (OnirisConvArgs.clip_flag)
@pytest.mark.usefixtures("nt_policy")
@pytest.mark.parametrize("gated,H,cout,clipped", [(False, 8, 64, True), (True, 8, 64, True), (True, 16, 64, True),
                                                  (True, 16, 64, False), (True, 16, 32, True), (True, 16, 32, False),
                                                  # the plain streaming kernel's epilogues (8 frames of 128x128 pixels = 1024 tiles)
                                                  (False, 128, 32, True), (False, 128, 32, False), (False, 128, 24, True)])
def test_conv_epilogues(gated, H, cout, clipped):
    """conv + (x c, mp_silu) and conv + (mp_sum with residual, clip) epilogues, forward and backward."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(11)
    B, T, cin = 1, 4, 32
    N = B * 2 * T
    w2 = O.normalize(O.normalize(torch.randn(cout, cin, 3, 3)))
    w3 = O.normalize(O.normalize(torch.randn(cout, cin, 2, 3, 3)))
    F = torch.nn.functional
    # private: the caller vouches that nobody else reads the output's gradient (ops.ConvCfg.grad_private, what UNet.forward does):
    # the clip_flag aliasing protocol -- no masked copy; the gradient is masked IN PLACE when the forward clipped.  Without it
    # (the default: y.backward(g) with the caller's own g) the gradient tensor must come back untouched.
    for epi, private in (("emb_silu", False), ("mpsum", False), ("mpsum", True)):
        p2, p3 = torch.nn.Parameter(w2.clone().to(DEV)), torch.nn.Parameter(w3.clone().to(DEV))
        bank, (pw2, pw3) = make_bank([p2, p3])
        bank.prepare(training=True)
        x0 = bfr(torch.randn(N, cin, H, H))
        g0 = torch.rand(N) * 0.5 + 0.05
        c0 = 1 + 0.3 * torch.randn(N, cout)
        r0 = bfr(torch.randn(N, cout, H, H) * (200 if clipped else 2))      # large residual: the +-256 clip is active
        gy0 = bfr(torch.randn(N, cout, H, H))
        x = nhwc(x0).requires_grad_(True)
        gate = g0.clone().to(DEV).requires_grad_(True)
        cs = c0.clone().to(DEV).requires_grad_(True)
        res = nhwc(r0).requires_grad_(True)
        kw = dict(cscale=cs) if epi == "emb_silu" else dict(res=res, ta=0.9, tb=0.4, clip=256.0)
        y = (ops.gated_conv_train(x, gate, pw2, pw3, B, T, grad_private=private, **kw) if gated else
             ops.conv(x, pw2, grad_private=private, **kw) if epi == "mpsum" else ops.conv(x, pw2, **kw))
        gy = nhwc(gy0)
        gy_before = gy.clone()
        y.backward(gy)
        if not private:
            assert torch.equal(gy, gy_before), "the backward wrote into the caller's gradient tensor"
        elif clipped and ops.CLIP_FLAG:
            assert not torch.equal(gy, gy_before), "aliasing protocol not exercised (expected the in-place mask)"
        # oracle
        xr, gr = x0.clone().requires_grad_(True), g0.clone().requires_grad_(True)
        cr, rr = c0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        w2r, w3r = w2.clone().requires_grad_(True), w3.clone().requires_grad_(True)
        e2, _ = O.weight_effective(w2r, 1.0, True)
        v = F.conv2d(xr, e2, padding=1)
        if gated:
            e3, _ = O.weight_effective(w3r, 1.0, True)
            clean = xr.reshape(B, 2, T, cin, H, H)[:, 0]
            ctx = torch.cat([torch.ones(B, 2, cin, H, H), clean], 1)
            y3 = F.conv2d(ctx[:, 0:T].reshape(B * T, cin, H, H), e3[:, :, 0], padding=1) + \
                F.conv2d(ctx[:, 1:T + 1].reshape(B * T, cin, H, H), e3[:, :, 1], padding=1)
            y3 = y3.reshape(B, 1, T, cout, H, H).expand(B, 2, T, cout, H, H).reshape(N, cout, H, H)
            v = O.mp_sum(v, y3, gr)
        if epi == "emb_silu":
            yr = O.mp_silu(v * cr[:, :, None, None])
        else:
            # the pass-through mask of clamp is taken from the bf16 result: elements within one bf16 ulp (2.0) of +-256
            # legitimately fall on the other side of the threshold than in fp32, and each such element is an O(1)
            # gradient difference (sqrt(fraction) in relative L2) that says nothing about the kernel
            pre = 0.9 * rr + 0.4 * v
            keep = (nchw(y).abs() < 256).float()
            yr = pre * keep + pre.detach().clamp(-256, 256) * (1 - keep)
        (yr * gy0).sum().backward()
        e = dict(y=rel(nchw(y), yr), dx=rel(nchw(x.grad), xr.grad), dw2=rel(p2.grad, w2r.grad))
        if epi == "emb_silu":
            e["dc"] = rel(cs.grad, cr.grad)
        else:
            e["dres"] = rel(nchw(res.grad), rr.grad)
            assert (float((nchw(y).abs() >= 256).float().mean()) > 0.01) == clipped, "clip (not) exercised"
        if gated:
            e["dw3"], e["dg"] = rel(p3.grad, w3r.grad), rel(gate.grad, gr.grad)
        print("conv epilogue", epi, "gated" if gated else "plain", "private" if private else "", e)
        assert e["y"] < 1e-2 and e["dx"] < 1.5e-2 and e["dw2"] < 2e-2
        assert all(v_ < 3e-2 for v_ in e.values()), e


def test_sampler_update_matches_tensor_expressions():
    """oniris_sampler_update against the reference's tensor expressions (edm2/sampler.py:66-76), sigmas as 0-dim device
    tensors like there: bit-exact (same fp32 operations in the same order), in-place aliasing as the frame loop uses it."""
    from autoregressive_diffusion_amd import ops
    torch.manual_seed(3)
    shape = (2, 1, 8, 16, 16)
    x_hat, x_pred, x_pred2 = (torch.randn(shape, device=DEV) * s for s in (40.0, 1.0, 1.0))
    t_hat, t_next = torch.tensor(37.25, device=DEV), torch.tensor(21.125, device=DEV)
    d_cur = (x_hat - x_pred) / t_hat
    x_e = x_hat + (t_next - t_hat) * d_cur
    d_prime = (x_e - x_pred2) / t_next
    x_n = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
    th, tn = t_hat.item(), t_next.item()
    xh, d, xin, sig = x_hat.clone(), torch.empty_like(x_hat), torch.empty_like(x_hat), torch.zeros(2, 1, device=DEV)
    ops.sampler_update(0, xh, x_pred, d, None, xin, th, tn - th, sig, tn)
    assert torch.equal(d, d_cur) and torch.equal(xin, x_e) and torch.equal(xh, x_hat)
    assert torch.equal(sig, torch.full((2, 1), tn, device=DEV))
    ops.sampler_update(1, xh, x_pred2, d, xin, xin, tn, tn - th)
    assert torch.equal(xh, x_n) and torch.equal(xin, x_n)


def test_eval_side_kernels_match_torch_formulation():
    """oniris_gates / oniris_embed_eval / oniris_precond_out (+ oniris_dart_input without noise) against the torch
    formulation of the same reference lines (conv.py:113-127, networks_edm2.py:204-216, :278-297)."""
    from autoregressive_diffusion_amd import ops
    from edm2.conv import Gating
    from edm2.utils import MPFourier, mp_sum, mp_silu
    torch.manual_seed(11)
    # gates: 5 layers, B = 2, t = 3, frame counters 0 / 7
    B, t, L = 2, 3, 5
    gs = [Gating().to(DEV) for _ in range(L)]
    for g_ in gs:
        for p in g_.parameters():
            p.data.add_(torch.randn_like(p) * 0.5)
    c_noise = torch.randn(B, t, device=DEV)
    nctx = [0, 7, 7, 0, 3]
    params = torch.stack([torch.cat([g_.mult, g_.offset, g_.min_gating.reshape(1), g_.max_gating.reshape(1)]) for g_ in gs]).detach()
    ca, cb = ops.gates_eval(c_noise.contiguous(), params.contiguous(), torch.tensor(nctx, dtype=torch.int32, device=DEV), t)
    for l, g_ in enumerate(gs):
        g_.eval()
        with torch.no_grad():
            gate, _ = g_(c_noise, nctx[l])
        wa, wb = ops.gate_coefs(gate.reshape(-1))
        assert torch.allclose(ca[l], wa, atol=2e-6, rtol=1e-5) and torch.allclose(cb[l], wb, atol=2e-6, rtol=1e-5)
    # embedding: cnoise 16, cemb 64, 4 labels (fp32 kernel vs fp32 torch math on the normalised weights)
    N, cn, cemb, Ld = 6, 16, 64, 4
    four = MPFourier(cn).to(DEV)
    wn, wl = torch.randn(cemb, cn, device=DEV), torch.randn(cemb, Ld, device=DEV)
    cnz, lab = torch.randn(N, device=DEV), torch.randint(0, Ld, (N,), device=DEV)

    def what(w):
        fan = w.shape[1]
        return w / (1e-4 + w.norm(dim=1, keepdim=True) / fan ** 0.5) / fan ** 0.5
    e = four(cnz) @ what(wn).t()
    oh = torch.nn.functional.one_hot(lab, Ld).float() * Ld ** 0.5
    want = mp_silu(mp_sum(e, oh @ what(wl).t(), t=1 / 3))
    got = ops.embed_eval(cnz, lab, four, wn, wl, Ld).reshape(N, cemb).float()
    assert rel(got, want) < 4e-3                                      # (bf16 output rounding)
    got0 = ops.embed_eval(cnz, None, four, wn, None, Ld).reshape(N, cemb).float()
    assert rel(got0, mp_silu(e)) < 4e-3
    # preconditioning around the UNet
    Bx, tx, C, H = 2, 3, 4, 16
    x = torch.randn(Bx, tx, C, H, H, device=DEV)
    sg = (torch.randn(Bx, tx, device=DEV) * 0.8).exp()
    sd = 0.5
    xcl = ops.dart_input(x, None, sg, 1, sd)
    cin = 1 / (sd ** 2 + sg ** 2).sqrt()
    want_in = (cin[:, :, None, None, None] * x).reshape(Bx * tx, C, H, H).permute(0, 2, 3, 1)
    assert rel(xcl[..., :C], want_in) < 4e-3 and torch.equal(xcl[..., C].float(), torch.ones_like(xcl[..., C].float()))
    assert float(xcl[..., C + 1:].abs().max()) == 0.0
    Fcl = torch.randn(Bx * tx, H, H, 8, device=DEV).to(torch.bfloat16)
    og = torch.tensor(0.7, device=DEV)
    D = ops.precond_out(Fcl, x, sg, og, sd)
    den = sg ** 2 + sd ** 2
    Fn = Fcl[..., :C].float().permute(0, 3, 1, 2).reshape(Bx, tx, C, H, H) * og
    wantD = (sd ** 2 / den)[:, :, None, None, None] * x + (sg * sd / den.sqrt())[:, :, None, None, None] * Fn
    assert torch.allclose(D, wantD, atol=1e-5, rtol=1e-5)


def test_prelude_kernels_and_adjoints_match_torch_autograd():
    """oniris_gates_bwd, oniris_emb_scale[_bwd], oniris_embed_pre / _post[_bwd] against torch autograd on the reference
    formulas (conv.py:113-127 training layout, networks_edm2.py:78, :204-212): values and every gradient."""
    from autoregressive_diffusion_amd import ops
    from edm2.utils import MPFourier, mp_sum, mp_silu
    torch.manual_seed(12)
    # ---- gates, training layout: B = 2, 2T = 8 slots, T = 4 positions, 5 layers, frame counters 0 / 3
    B, T, L = 2, 4, 5
    P = (torch.randn(L, 6, device=DEV) * 0.7).requires_grad_(True)         # mult0, mult1, off0, off1, min, max
    c_noise = torch.randn(B, 2 * T, device=DEV)
    nctx = torch.tensor([0, 3, 0, 3, 1], dtype=torch.int32, device=DEV)
    ca, cb = ops.gates_train(c_noise.reshape(-1).contiguous(), P, nctx, T)
    ra, rb = torch.randn_like(ca), torch.randn_like(cb)
    (ca * ra + cb * rb).sum().backward()
    got_g, P.grad = P.grad.clone(), None
    pos = ((torch.arange(B * 2 * T, device=DEV) % T)[None] + nctx[:, None]).float().log1p()
    sv = c_noise.reshape(1, -1) * P[:, 0:1] + P[:, 2:3] + pos * P[:, 1:2] + P[:, 3:4]
    lo, hi = torch.sigmoid(P[:, 4:5]), torch.sigmoid(P[:, 5:6])
    g = lo + (1 - lo) * hi * torch.sigmoid(sv)
    wa, wb = ops.gate_coefs(g)
    assert torch.allclose(ca, wa, atol=2e-6, rtol=1e-5) and torch.allclose(cb, wb, atol=2e-6, rtol=1e-5)
    (wa * ra + wb * rb).sum().backward()
    assert torch.allclose(got_g, P.grad, atol=1e-5, rtol=1e-4), (got_g - P.grad).abs().max()
    # ---- emb scales: 3 blocks of 64 / 128 / 64 columns (64-aligned: no pad columns), N = 7 rows
    N, widths = 7, [64, 128, 64]
    Ct, K = sum(widths), len(widths)
    seg = torch.tensor([k for k, w in enumerate(widths) for _ in range(w)], dtype=torch.int32, device=DEV)
    start = torch.tensor([0, 64, 192, 256], dtype=torch.int32, device=DEV)
    c_all = torch.randn(N, Ct, device=DEV).to(torch.bfloat16).requires_grad_(True)
    gain = torch.randn(K, device=DEV).requires_grad_(True)
    c = ops._EmbScaleFn.apply(c_all, gain, seg, start)
    rc = torch.randn_like(c)
    (c * rc).sum().backward()
    got_dc, got_dg = c_all.grad.clone(), gain.grad.clone()
    c_all.grad = gain.grad = None
    want = 1 + c_all.float() * gain[seg.long()][None]
    assert torch.allclose(c, want, atol=1e-6, rtol=1e-6)
    (want * rc).sum().backward()
    assert rel(got_dc, c_all.grad) < 4e-3 and torch.allclose(got_dg, gain.grad, atol=1e-4, rtol=1e-4)
    # ---- embedding pre / post
    Ne, cn, Ld = 6, 12, 4                                               # (cn, Ld not multiples of 8: padded columns)
    four = MPFourier(cn).to(DEV)
    cnz, lab = torch.randn(Ne, device=DEV), torch.randint(0, Ld, (Ne,), device=DEV)
    f_ = torch.empty(Ne, 16, dtype=torch.bfloat16, device=DEV)
    oh = torch.empty(Ne, 8, dtype=torch.bfloat16, device=DEV)
    from autoregressive_diffusion_amd._lib import lib, check
    check(lib.oniris_embed_pre(cnz.data_ptr(), lab.data_ptr(), four.freqs.data_ptr(), four.phases.data_ptr(), f_.data_ptr(),
                               oh.data_ptr(), Ne, cn, 16, Ld, 8, ops._stream()), "embed_pre")
    assert rel(f_[:, :cn], four(cnz)) < 4e-3 and float(f_[:, cn:].abs().max()) == 0.0
    assert torch.equal(oh[:, :Ld].float(), torch.nn.functional.one_hot(lab, Ld).float() * 2.0) and float(oh[:, Ld:].abs().max()) == 0.0
    for with_e2 in (True, False):
        e1 = torch.randn(Ne, 64, device=DEV).to(torch.bfloat16).requires_grad_(True)
        e2 = torch.randn(Ne, 64, device=DEV).to(torch.bfloat16).requires_grad_(True) if with_e2 else None
        emb = ops._EmbedPostFn.apply(e1, e2, 1 / 3)
        re_ = torch.randn(Ne, 64, device=DEV).to(torch.bfloat16)
        (emb.float() * re_.float()).sum().backward()
        g1, g2 = e1.grad.clone(), (e2.grad.clone() if with_e2 else None)
        e1.grad = None
        if with_e2:
            e2.grad = None
        want = mp_silu(mp_sum(e1.float(), e2.float(), t=1 / 3)) if with_e2 else mp_silu(e1.float())
        assert rel(emb, want) < 4e-3
        (want * re_.float()).sum().backward()
        assert rel(g1, e1.grad) < 6e-3 and (not with_e2 or rel(g2, e2.grad) < 6e-3)
