"""Torch-autograd formulation of the conditioning prelude (all Gating modules, the noise / label embedding, the per-Block
emb scales): TEST INFRASTRUCTURE.  The product runs these on fused HIP kernels with hand-written adjoints
(oniris_gates[_bwd], oniris_embed_pre / _post[_bwd], oniris_emb_scale[_bwd]); the parity test
(tests/test_model_gpu.py::test_fused_prelude_matches_torch_formulation) installs this module as
`ops.prelude_reference` and sets `ops.FUSED_PRELUDE = 0` to run the same net through torch ops instead.
Same math as the reference: edm2/conv.py:113-127 (Gating), networks_edm2.py:204-216 (embedding), :78 (emb scales)."""
import math
import torch
import torch.nn.functional as F


def batched_gates(convs, c_noise, caches, training, n_ctx, T, nctx_tensor):
    from autoregressive_diffusion_amd import ops
    B, tt = c_noise.shape
    dev = c_noise.device
    mult = torch.stack([m.gating.mult for m in convs])            # (L,2)
    off = torch.stack([m.gating.offset for m in convs])           # (L,2)
    lo = torch.sigmoid(torch.stack([m.gating.min_gating for m in convs]))[:, None, None]
    hi = torch.sigmoid(torch.stack([m.gating.max_gating for m in convs]))[:, None, None]
    base = (torch.arange(B * tt, device=dev) % T).reshape(1, B, tt)
    if any(n_ctx):
        base = base + nctx_tensor(n_ctx, dev)
    pos = base.to(c_noise.dtype).log1p()
    sv = c_noise[None] * mult[:, 0, None, None] + off[:, 0, None, None] + pos * mult[:, 1, None, None] + off[:, 1, None, None]
    g = (lo + (1 - lo) * hi * torch.sigmoid(sv)).reshape(len(convs), -1)
    ca, cb = ops.gate_coefs(g)
    return [(a, b, n + T) for a, b, n in zip(ca.unbind(0), cb.unbind(0), n_ctx)]


def embedding(unet, cn, conditioning):
    from autoregressive_diffusion_amd.edm2.utils import mp_silu, mp_sum, BF16
    emb = unet.emb_noise.forward(unet.emb_fourier_sigma(cn))
    if unet.emb_label is not None and conditioning is not None:
        oh = F.one_hot(conditioning.reshape(-1), num_classes=unet.label_dim).to(cn.dtype) * math.sqrt(unet.label_dim)
        emb = mp_sum(emb, unet.emb_label.forward(oh), t=1 / 3)
    emb = mp_silu(emb)
    return emb.to(BF16)[:, None, None, :].contiguous()


def emb_scales(c_all, gpw, gains, split_cols):
    from autoregressive_diffusion_amd.ops import roundup
    dev = c_all.device
    sizes, seg = [], []
    for k, m in enumerate(gpw.members):
        sizes.append(m.cout)
        seg += [k] * m.cout
        pad = roundup(m.cout, 64) - m.cout
        if pad:
            sizes.append(pad)
            seg += [k] * pad
    cache = torch.tensor(seg, dtype=torch.int64, device=dev)
    g_col = torch.stack(list(gains)).float().index_select(0, cache)                 # (Ctot,)
    c = torch.addcmul(torch.ones((), dtype=torch.float32, device=dev), c_all.float(), g_col)
    outs = split_cols.apply(c, tuple(sizes))
    res, j = [], 0
    for m in gpw.members:
        res.append(outs[j])
        j += 2 if roundup(m.cout, 64) != m.cout else 1
    return res
