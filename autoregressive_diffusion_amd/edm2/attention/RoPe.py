"""RotaryEmbedding with the reference's buffers/signature (edm2/attention/RoPe.py:5-74).  The hot path applies
the rotation inside the HIP rope kernel (ops.rope_tables builds the fp16-rounded tables); this module keeps the
state_dict entries (`inv_freq`, `scale`) and offers the standalone forward for API compatibility."""
import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, scale_base=64):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim)))
        self.scale_base = scale_base
        self.register_buffer("scale", (torch.arange(0, dim, 2) + 0.4 * dim) / (1.4 * dim))
        self.register_buffer("pos_emb", None, persistent=False)
        self.register_buffer("pos_emb_scale", None, persistent=False)

    def make_rotary_embedding(self, seq_len):
        t = torch.arange(seq_len, device=self.inv_freq.device).type_as(self.inv_freq)
        ang = torch.outer(t, self.inv_freq)
        ang = torch.cat((ang, ang), dim=-1).to(torch.float16)
        scale = self.scale[None, :] ** ((t - (seq_len // 2)) / self.scale_base)[:, None]
        scale = torch.cat((scale, scale), dim=-1).to(torch.float16)
        return ang.unsqueeze(1), scale.unsqueeze(1)

    def forward(self, q, k):
        """q, k: (b, m, frames, hw, c) -> rotated (b, m, frames*hw, c); train mode: frames = 2T (clean|noisy)."""
        nk = k.shape[-3] // 2 if self.training else k.shape[-3]
        pos, scale = self.make_rotary_embedding(nk)
        cos, sin = pos.cos(), pos.sin()
        if self.training:
            cos, sin, scale = (torch.cat((z, z), dim=0) for z in (cos, sin, scale))
        k = (k * cos + rotate_half(k) * sin) / scale
        nq = q.shape[-3]
        q = (q * cos[-nq:] + rotate_half(q) * sin[-nq:]) * scale[-nq:]
        return q.flatten(-3, -2), k.flatten(-3, -2)


def rotate_half(x):
    a, b = x.chunk(2, dim=-1)
    return torch.cat((-b, a), dim=-1)
