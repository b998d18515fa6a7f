"""UNet.load_from_2d against the reference (fixture G10: per-key sums of the reference's state_dict after the same
import from the same 2-D stand-in).  Pure parameter copies: runs on CPU, no HIP call."""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import paramgen   # noqa: E402
import twod       # noqa: E402

SMALL_CFG = dict(img_resolution=32, img_channels=4, label_dim=4, model_channels=16, channel_mult=[1, 4, 4],
                 num_blocks=1, video_attn_resolutions=[8], frame_attn_resolutions=[16])
C1_CFG = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=16, channel_mult=[1, 2, 4, 8],
              num_blocks=1, video_attn_resolutions=[8], frame_attn_resolutions=[16])


def test_load_from_2d_matches_reference():
    from edm2.networks_edm2 import UNet
    z = np.load(os.path.join(HERE, "golden", "g10_load2d.npz"))
    for tag, cfg in (("small", SMALL_CFG), ("c1", C1_CFG)):
        sa, sb = (int(v) for v in z[tag + "_seeds"])
        pa, pb = paramgen.unet_params(cfg, sa), paramgen.unet_params(cfg, sb)
        unet = UNet(**cfg)
        unet.load_state_dict({k: v.clone() for k, v in pa.items()}, strict=True)
        unet.load_from_2d(twod.Net2D(pb, list(unet.enc.keys()), list(unet.dec.keys())))
        sd = unet.state_dict()
        keys, sums, asums = twod.state_sums(sd)
        assert keys == [str(k) for k in z[tag + "_keys"]], "state_dict keys differ from the reference's"
        assert np.array_equal(np.array(sums), z[tag + "_sums"]) and np.array_equal(np.array(asums), z[tag + "_abs"])
        # and the semantics spelled out: 2-D convs land in the own-frame path, context weights / gates / emb_time stay
        for k, v in sd.items():
            if not (torch.is_tensor(v) and v.is_floating_point()):
                continue
            if k.endswith("last_frame_conv.weight.weight") or k.endswith("emb_gain") or k == "out_gain" \
                    or "emb_linear" in k or "conv_skip" in k or "attn_qkv" in k or "attn_proj" in k \
                    or k.startswith("emb_noise") or k.startswith("emb_label"):
                assert torch.equal(v, pb[k]), k
            elif k.startswith("emb_fourier_"):
                assert torch.equal(v, pb["emb_fourier_sigma." + k.split(".")[-1]]), k
            else:
                assert torch.equal(v, pa[k]), k
