"""The reference's own self-consistency properties (edm2/consistency_test.py, 11 tests, fp32, std(diff) <= 3e-4) stated
for the HIP modules behind the same `edm2.*` API.  Same constants where the shipped kernels allow (16x16 images = 256
tokens per frame, 8 frames, cut at frame 3); what differs and why:

  * head dimension: the reference's test modules use 4 heads of 16 channels; every BASELINE configuration has
    channels_per_head = 64 (networks_edm2.py:28,39).  Every attention test here runs at both: 4 x 16 (the reference's own
    constants, through the zero-padded-head path) and 4 x 64 (the shipped kernels' native width);
  * the UNet of the reference's test (resolution 16 -> a 2x2 bottom level) is run at resolution 32 (4x4 bottom level, the
    smallest image the conv kernels tile);
  * tolerance: bf16 operands.  Where both sides run the SAME arithmetic (cached vs non-cached: per-token results do not
    depend on how many query tokens a launch carries) the reference's 3e-4 is kept; where the two sides are different
    kernels or an fp32 torch restatement, the bound is 1e-2 of the output's own standard deviation (stated per test).
"""
import math
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.selfcheck]      # (self-consistency properties: not counted as oracle coverage)
DEV = "cuda"
RES, B, CH, T_, CUT, SEED = 16, 4, 16, 8, 3, 42
TIGHT = 3e-4          # consistency_test.py:32


def _std(a, b):
    return (a.float() - b.float()).std().item()


def _bt(x, b):
    return x.reshape(b, -1, *x.shape[1:])


HEAD_CHANNELS = (16, 48, 64)      # consistency_test.py:39,61 (4 * IMG_CHANNELS over 4 heads); networks_edm2.py:28; 48: a width that
                                  # is neither a power of two nor a BASELINE value (every multiple of 8 up to 64 is served, round 5)


@pytest.fixture(scope="module", params=HEAD_CHANNELS, ids=lambda c: f"head{c}")
def video_attention(request):
    from edm2.attention import VideoAttention
    torch.manual_seed(SEED)
    return VideoAttention(channels=4 * request.param, num_heads=4).to(DEV)


def _qkv_split(att, x, b):
    """Normalised q, k, v (b, m, t, hw, c) in fp32 from the module's own qkv conv (attention_modules.py:48-49)."""
    from edm2.utils import normalize
    y = att.attn_qkv(x).float()
    n, _, h, w = y.shape
    y = y.reshape(b, n // b, att.num_heads, -1, 3, h * w).permute(4, 0, 2, 1, 5, 3)       # s b m t hw c
    return normalize(y, dim=-1).unbind(0)


@pytest.mark.parametrize("hc", HEAD_CHANNELS)
def test_frame_attention_matches_manual_softmax(hc):
    """consistency_test.py:41-55"""
    from edm2.attention import FrameAttention
    from edm2.utils import mp_sum
    torch.manual_seed(SEED)
    att = FrameAttention(channels=4 * hc, num_heads=4).to(DEV)
    x = torch.randn(B * 2 * T_, 4 * hc, RES, RES, device=DEV)
    with torch.no_grad():
        y_frame, _ = att(x.clone())
        q, k, v = _qkv_split(att, x, x.shape[0])                       # every image its own "sequence" of one frame
        w = torch.einsum("bmtqc,bmtkc->bmtqk", q, k / math.sqrt(q.shape[-1])).softmax(dim=-1)
        y = torch.einsum("bmtqk,bmtkc->bmtqc", w, v)                    # b m 1 hw c
        y = y[:, :, 0].permute(0, 1, 3, 2).reshape(x.shape)
        y = mp_sum(x, att.attn_proj(y), t=att.attn_balance)
    assert _std(y, y_frame) <= 1e-2 * y.std().item()                   # bf16 kernels vs fp32 formula


def test_video_vs_frame_attention_first_frame(video_attention):
    """consistency_test.py:63-74: with just_2d every frame attends to itself only -- so does frame 0 of each sequence under
    the training mask; later frames must differ."""
    att = video_attention.train()
    x = torch.randn(B * 2 * T_, att.channels, RES, RES, device=DEV)
    with torch.no_grad():
        yv, _ = att(x.clone(), B, just_2d=False)
        yf, _ = att(x.clone(), B, just_2d=True)
    yv, yf = (z.reshape(B, 2, T_, *z.shape[1:]).permute(1, 0, 2, 3, 4, 5).reshape(2 * B, T_, *z.shape[1:]) for z in (yv, yf))
    d = (yv.float() - yf.float()).std(dim=(0, 2, 3, 4))
    assert d[0].item() <= 1e-2 * yf.std().item()                       # video (table kernel) vs frame (dense kernel)
    assert d[1:].mean().item() >= 1e-2


def test_video_attention_matches_masked_sdpa(video_attention):
    """consistency_test.py:79-103: the block-sparse training kernel against dense SDPA under the token-level mask."""
    from oracle import oniris_oracle as O
    from edm2.utils import mp_sum
    att = video_attention.train()
    x = torch.randn(B * 2 * T_, att.channels, RES, RES, device=DEV)
    with torch.no_grad():
        yv, _ = att(x.clone(), B, just_2d=False)
        q, k, v = _qkv_split(att, x, B)                                # b m 2T hw c
        q, k = O.rope_apply(q.cpu(), k.cpu(), att.rope.inv_freq.float().cpu(), att.rope.scale.float().cpu(), True)
        hc = att.channels // att.num_heads
        q, k, v = (z.reshape(B, att.num_heads, -1, hc).to(DEV) for z in (q, k, v.cpu()))
        allowed = torch.from_numpy(O.train_allowed_tokens(T_, RES * RES)).to(DEV)
        y = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=allowed)
        y = y.reshape(B, att.num_heads, 2 * T_, RES, RES, hc).permute(0, 2, 1, 5, 3, 4).reshape(x.shape)
        y = mp_sum(x, att.attn_proj(y), t=att.attn_balance)
    assert _std(y, yv) <= 1e-2 * y.std().item()


def test_video_attention_train_vs_eval(video_attention):
    """consistency_test.py:108-125: clean frames 0..CUT-1 and the noised frame CUT of a training pass equal an eval pass
    over [clean 0..CUT-1, noised CUT]."""
    att = video_attention
    x = torch.randn(B * 2 * T_, att.channels, RES, RES, device=DEV)
    with torch.no_grad():
        y_train, _ = att.train()(x, B)
        xs = _bt(x, B)
        x_eval = torch.cat((xs[:, :CUT], xs[:, CUT + T_].unsqueeze(1)), dim=1).reshape(-1, *x.shape[1:])
        y_eval, _ = att.eval()(x_eval, B)
    y_train, y_eval = _bt(y_train, B), _bt(y_eval, B)
    s = y_train.std().item()
    assert _std(y_train[:, :CUT], y_eval[:, :-1]) <= 1e-2 * s          # training-table kernel vs causal-prefill kernel
    assert _std(y_train[:, CUT + T_], y_eval[:, -1]) <= 1e-2 * s


def test_video_attention_cached_vs_non_cached(video_attention):
    """consistency_test.py:129-146"""
    att = video_attention.eval()
    x = torch.randn(B, T_, att.channels, RES, RES, device=DEV)
    flat = lambda z: z.reshape(-1, *z.shape[2:])
    with torch.no_grad():
        y_full, _ = att(flat(x), B)
        y_ctx, cache = att(flat(x[:, :-1]), B, update_cache=True)
        out, _ = att(flat(x[:, -1:]), B, cache)
    y_full = _bt(y_full, B)
    e = (_std(y_full[:, -1], _bt(out, B)[:, 0]), _std(y_full[:, :-1], _bt(y_ctx, B)), y_full.std().item())
    print("attention cached vs non-cached: std(diff) last frame, context frames, std(y)", e)
    assert e[0] <= 1e-2 * e[2]             # prefill kernel vs decode kernel
    assert e[1] <= 3e-3 * e[2]             # prefill over 8 vs 7 frames: two key streams per workgroup from 2048 keys on


def test_video_attention_cached_vs_non_cached_multistep(video_attention):
    """consistency_test.py:148-172"""
    att = video_attention.eval()
    b = 1
    x = torch.randn(b, T_, att.channels, RES, RES, device=DEV)
    flat = lambda z: z.reshape(-1, *z.shape[2:])
    with torch.no_grad():
        y_full, _ = att(flat(x), b)
        _, cache = att(flat(x[:, :-2]), b, update_cache=True)
        out1, cache = att(flat(x[:, -2:-1]), b, cache, update_cache=True)
        out2, _ = att(flat(x[:, -1:]), b, cache)
    y_full = _bt(y_full, b)
    got = torch.cat((_bt(out1, b), _bt(out2, b)), dim=1)
    assert _std(y_full[:, -2:], got) <= 1e-2 * y_full.std().item()


@pytest.fixture(scope="module")
def unet():
    from edm2.networks_edm2 import UNet
    torch.manual_seed(SEED)
    net = UNet(img_resolution=32, img_channels=CH, label_dim=0, model_channels=32, channel_mult=[1, 2, 2, 4],
               channel_mult_noise=None, channel_mult_emb=None, num_blocks=3, video_attn_resolutions=[16, 8]).to(DEV)
    with torch.no_grad():                  # (out_gain is initialised to 0, the gates almost closed: give the output a scale
        net.out_gain.fill_(1.0)            #  and the temporal paths a weight)
        for n, p in net.named_parameters():
            if n.endswith("gating.max_gating"):
                p.fill_(1.0)
            elif n.endswith("gating.min_gating"):
                p.fill_(-1.0)
    return net


def test_unet_train_vs_eval(unet):
    """consistency_test.py:196-211"""
    x = torch.randn(2, 2 * T_, CH, 32, 32, device=DEV)
    noise = torch.zeros(x.shape[:2], device=DEV)
    with torch.no_grad():
        y_train, _ = unet.train()(x, noise, conditioning=None)
        x_eval = torch.cat((x[:, :CUT], x[:, CUT + T_].unsqueeze(1)), dim=1)
        n_eval = torch.cat((noise[:, :CUT], noise[:, CUT + T_].unsqueeze(1)), dim=1)
        y_eval, _ = unet.eval()(x_eval, n_eval, conditioning=None)
    s = y_train.std().item()
    assert s > 0
    assert _std(y_train[:, :CUT], y_eval[:, :-1]) <= 2e-2 * s          # 27 blocks deep, different kernels on both sides
    assert _std(y_train[:, CUT + T_], y_eval[:, -1]) <= 2e-2 * s


def test_unet_causality(unet):
    """consistency_test.py:214-228: a change in clean frame CUT reaches neither the clean nor the noised frames before it."""
    x = torch.zeros(2, T_, CH, 32, 32, device=DEV)
    r = torch.randn(2, T_, CH, 32, 32, device=DEV)
    a = torch.cat((x, r), dim=1)
    x[:, CUT] = torch.randn(2, CH, 32, 32, device=DEV)
    x = torch.cat((x, r), dim=1)
    noise = torch.zeros(x.shape[:2], device=DEV)
    with torch.no_grad():
        unet.train()
        # a training-mode forward re-normalises the stored weights (conv.py:16-18): two calls on the SAME input differ by
        # a few bf16 roundings until the fixed point is reached -- that difference is the floor the causal frames are held to
        unet(a, noise, None)
        y0, yb = unet(a, noise, None)[0], unet(a, noise, None)[0]
        ya = unet(x, noise, None)[0]
    floor = (y0 - yb).float().std().item()
    y = (ya - yb).float()
    scale = ya.std().item()
    e = (y[:, :CUT].std().item(), y[:, CUT:T_].std().item(), y[:, T_:T_ + CUT].std().item(), y[:, T_ + CUT:].std().item(), scale, floor)
    print("unet causality: std(diff) clean before / clean from / noised before / noised from the cut, std(y), floor", e)
    bound = max(2.0 * floor, 1e-3 * scale)
    assert e[0] <= bound and e[2] <= bound
    assert e[1] > 0.3 * scale                                          # the changed clean frame and its successors
    assert e[3] > 5.0 * bound                                          # noised frames that see the changed clean frame


@pytest.fixture(scope="module")
def conv3d():
    from edm2.conv import MPCausal3DGatedConv
    torch.manual_seed(SEED)
    conv = MPCausal3DGatedConv(CH, CH, kernel=(3, 3, 3)).to(DEV)
    with torch.no_grad():                  # (gates are initialised almost closed: open them so the context path counts)
        conv.gating.max_gating.fill_(1.0)
        conv.gating.min_gating.fill_(-1.0)
    return conv


def test_conv_train_vs_eval(conv3d):
    """consistency_test.py:239-259"""
    x = torch.randn(B * 2 * T_, CH, RES, RES, device=DEV)
    c_noise = torch.randn(B, 2 * T_, device=DEV)
    with torch.no_grad():
        y_train, _ = conv3d.train()(x, None, B, c_noise)
        xs = _bt(x, B)
        x_eval = torch.cat((xs[:, :CUT], xs[:, CUT + T_].unsqueeze(1)), dim=1).reshape(-1, *x.shape[1:])
        cn = torch.cat((c_noise[:, :CUT], c_noise[:, CUT + T_].unsqueeze(1)), dim=1)
        y_eval, _ = conv3d.eval()(x_eval, None, B, cn)
    y_train, y_eval = _bt(y_train, B), _bt(y_eval, B)
    s = y_train.std().item()
    assert _std(y_train[:, :CUT], y_eval[:, :-1]) <= 1e-2 * s          # DART-layout kernel vs eval kernel
    assert _std(y_train[:, CUT + T_], y_eval[:, -1]) <= 1e-2 * s


def test_conv_cached_vs_non_cached(conv3d):
    """consistency_test.py:261-280"""
    conv3d.eval()
    x = torch.randn(B, T_, CH, RES, RES, device=DEV)
    c_noise = torch.randn(B, T_, device=DEV)
    flat = lambda z: z.reshape(-1, *z.shape[2:])
    with torch.no_grad():
        y_full, _ = conv3d(flat(x), None, B, c_noise)
        y_ctx, cache = conv3d(flat(x[:, :-1]), None, B, c_noise[:, :-1], update_cache=True)
        out, _ = conv3d(flat(x[:, -1:]), None, B, c_noise[:, -1:], cache=cache)
    got = torch.cat((_bt(y_ctx, B), _bt(out, B)), dim=1)
    assert _std(_bt(y_full, B), got) <= TIGHT * max(1.0, y_full.std().item())


def test_conv_cached_vs_non_cached_multistep(conv3d):
    """consistency_test.py:282-307"""
    conv3d.eval()
    x = torch.randn(B, T_, CH, RES, RES, device=DEV)
    c_noise = torch.randn(B, T_, device=DEV)
    flat = lambda z: z.reshape(-1, *z.shape[2:])
    with torch.no_grad():
        y_full, _ = conv3d(flat(x), None, B, c_noise)
        y_ctx, cache = conv3d(flat(x[:, :-2]), None, B, c_noise[:, :-2], update_cache=True)
        out1, cache = conv3d(flat(x[:, -2:-1]), None, B, c_noise[:, -2:-1], cache=cache, update_cache=True)
        out2, _ = conv3d(flat(x[:, -1:]), None, B, c_noise[:, -1:], cache=cache)
    got = torch.cat((_bt(y_ctx, B), _bt(out1, B), _bt(out2, B)), dim=1)
    assert _std(_bt(y_full, B), got) <= TIGHT * max(1.0, y_full.std().item())
