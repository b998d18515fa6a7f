"""Deterministic parameter sets for fixtures: the same tensors are loaded into the reference modules (by
make_golden.py, strict=True -> pins state_dict key names and shapes) and handed to the oracle / the HIP path
(by the tests).  Only torch CPU RNG is used, so fixtures need not store multi-MB state dicts."""
import math
import torch
from oracle import oniris_oracle as O


def _conv_keys(prefix, cin, cout):
    return {
        prefix + "last_frame_conv.weight.weight": (cout, cin, 3, 3),
        prefix + "weight.weight": (cout, cin, 2, 3, 3),
        prefix + "gating.offset": (2,), prefix + "gating.mult": (2,),
        prefix + "gating.max_gating": (), prefix + "gating.min_gating": (),
    }


def unet_param_shapes(cfg, d_head=64):
    lay = O.unet_layout(**cfg)
    sh = {"out_res.offset": (2,), "out_res.mult": (2,), "out_res.max_gating": (), "out_res.min_gating": (),
          "out_gain": (),
          "emb_fourier_sigma.freqs": (lay["cnoise"],), "emb_fourier_sigma.phases": (lay["cnoise"],),
          "emb_noise.weight.weight": (lay["cemb"], lay["cnoise"]),
          "emb_fourier_time.freqs": (lay["cnoise"],), "emb_fourier_time.phases": (lay["cnoise"],),
          "emb_time.weight.weight": (lay["cemb"], lay["cnoise"])}
    if lay["label_dim"]:
        sh["emb_label.weight.weight"] = (lay["cemb"], lay["label_dim"])
    for side in ("enc", "dec"):
        for e in lay[side]:
            pre = f"{side}.{e['name']}."
            if e["kind"] == "conv":
                sh.update(_conv_keys(pre, e["cin"], e["cout"]))
                continue
            sh[pre + "emb_gain"] = ()
            sh[pre + "emb_linear.weight.weight"] = (e["cout"], lay["cemb"])
            c0 = e["cout"] if e["flavor"] == "enc" else e["cin"]
            sh.update(_conv_keys(pre + "conv_res0.", c0, e["cout"]))
            sh.update(_conv_keys(pre + "conv_res1.", e["cout"], e["cout"]))
            if e["cin"] != e["cout"]:
                sh[pre + "conv_skip.weight.weight"] = (e["cout"], e["cin"], 1, 1)
            if e["heads"]:
                C = e["cout"]
                sh[pre + "attn.attn_qkv.weight.weight"] = (3 * C, C, 1, 1)
                sh[pre + "attn.attn_proj.weight.weight"] = (C, C, 1, 1)
                if e["attention"] == "video":
                    sh[pre + "attn.rope.inv_freq"] = (d_head // 2,)
                    sh[pre + "attn.rope.scale"] = (d_head // 2,)
    sh.update(_conv_keys("out_conv.", lay["cout"], lay["img_channels"]))
    return sh


def fill(shapes, seed, d_head=64):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k in sorted(shapes):
        s = shapes[k]
        if k.endswith("rope.inv_freq"):
            p[k] = 1.0 / (10000 ** (torch.arange(0, d_head, 2).float() / d_head))
        elif k.endswith("rope.scale"):
            p[k] = (torch.arange(0, d_head, 2) + 0.4 * d_head) / (1.4 * d_head)
        elif k.endswith(".freqs"):
            p[k] = 2 * math.pi * torch.randn(s, generator=g)
        elif k.endswith(".phases"):
            p[k] = 2 * math.pi * torch.rand(s, generator=g)
        elif k.endswith("gating.mult") or k.endswith("out_res.mult"):
            p[k] = torch.tensor([1.5, -0.5]) + 0.2 * torch.randn(2, generator=g)
        elif k.endswith("offset"):
            p[k] = 0.3 * torch.randn(2, generator=g)
        elif k.endswith("max_gating"):
            p[k] = -1.0 + 0.3 * torch.randn((), generator=g)
        elif k.endswith("min_gating"):
            p[k] = -3.0 + 0.3 * torch.randn((), generator=g)
        elif k.endswith("emb_gain"):
            p[k] = 0.5 + 0.2 * torch.randn((), generator=g)
        elif k == "out_gain":
            p[k] = torch.tensor(1.3)
        else:
            p[k] = torch.randn(s, generator=g)
    return p


def unet_params(cfg, seed):
    return fill(unet_param_shapes(cfg), seed)


def precond_params(cfg, seed):
    p = {"unet." + k: v for k, v in unet_params(cfg, seed).items()}
    g = torch.Generator().manual_seed(seed + 1)
    p["noise_weight.fourier_approximator.coefficients"] = 0.1 * torch.randn(7, 1, generator=g)
    return p


def prenormalise(p):
    """Apply the forced weight normalisation twice (fixed point to fp32 precision, SURVEY 8c caveat) so that a
    training-mode forward leaves the stored weights (numerically) unchanged."""
    q = dict(p)
    for k, v in p.items():
        if k.endswith("weight.weight"):
            q[k] = O.normalize(O.normalize(v))
    return q
