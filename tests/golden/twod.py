"""A stand-in for NVIDIA's 2-D EDM2 UNet (the object `UNet.load_from_2d(unet2d)` imports from, reference
networks_edm2.py:96-110,238-258 / test.py:28): only its module tree and state_dict key layout matter -- per block
`emb_gain, conv_res0.weight, emb_linear.weight, conv_res1.weight [, conv_skip.weight] [, attn_qkv.weight,
attn_proj.weight]`, at the top `emb_fourier.{freqs,phases}, emb_noise.weight, emb_label.weight, out_conv.weight,
out_gain`, enc/dec ModuleDicts in the same order as the 3-D net.  Built from a paramgen parameter set (3-D key names)."""
import torch
from torch import nn


class _W(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.weight = nn.Parameter(w.clone())


class _Fourier(nn.Module):
    def __init__(self, freqs, phases):
        super().__init__()
        self.register_buffer("freqs", freqs.clone())
        self.register_buffer("phases", phases.clone())


class _Block2D(nn.Module):
    def __init__(self, p, pre):
        super().__init__()
        self.emb_gain = nn.Parameter(p[pre + "emb_gain"].clone())
        self.conv_res0 = _W(p[pre + "conv_res0.last_frame_conv.weight.weight"])
        self.emb_linear = _W(p[pre + "emb_linear.weight.weight"])
        self.conv_res1 = _W(p[pre + "conv_res1.last_frame_conv.weight.weight"])
        if pre + "conv_skip.weight.weight" in p:
            self.conv_skip = _W(p[pre + "conv_skip.weight.weight"])
        if pre + "attn.attn_qkv.weight.weight" in p:
            self.attn_qkv = _W(p[pre + "attn.attn_qkv.weight.weight"])
            self.attn_proj = _W(p[pre + "attn.attn_proj.weight.weight"])


class Net2D(nn.Module):
    def __init__(self, p, enc_names, dec_names):
        """p: paramgen.unet_params(...) (3-D key names, no 'unet.' prefix); enc_names / dec_names: block order."""
        super().__init__()
        self.enc, self.dec = nn.ModuleDict(), nn.ModuleDict()
        for side, names, md in (("enc", enc_names, self.enc), ("dec", dec_names, self.dec)):
            for n in names:
                pre = f"{side}.{n}."
                md[n] = _W(p[pre + "last_frame_conv.weight.weight"]) if pre + "emb_gain" not in p else _Block2D(p, pre)
        self.emb_fourier = _Fourier(p["emb_fourier_sigma.freqs"], p["emb_fourier_sigma.phases"])
        self.emb_noise = _W(p["emb_noise.weight.weight"])
        if "emb_label.weight.weight" in p:
            self.emb_label = _W(p["emb_label.weight.weight"])
        self.out_conv = _W(p["out_conv.last_frame_conv.weight.weight"])
        self.out_gain = nn.Parameter(p["out_gain"].clone())


def state_sums(sd):
    """(sorted keys, float64 sums, float64 abs-sums) of a state dict: the fixture content."""
    keys = sorted(k for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point())
    return keys, [float(sd[k].double().sum()) for k in keys], [float(sd[k].double().abs().sum()) for k in keys]
