"""VideoAttention / FrameAttention with the reference's signatures (edm2/attention/attention_modules.py:15-119).
qkv 1x1 conv -> per-head pixel norm -> RoPE over the frame index -> block-sparse DART attention (train) /
causal prefill / KV-cached decode (eval) -> proj 1x1 conv fused with mp_sum(x, y, attn_balance)."""
import math
import torch
from torch import nn

from ... import ops
from ... import fp32 as _fp32
from ..conv import MPConv, weights_ready
from ..utils import to_cl, from_cl
from .RoPe import RotaryEmbedding


class _AttentionBase(nn.Module):
    def __init__(self, channels, num_heads, attn_balance=0.3):
        super().__init__()
        self.channels, self.num_heads, self.attn_balance = channels, num_heads, attn_balance
        if num_heads == 0:
            return
        # every configuration of the reference builds its attention with channels_per_head = 64 (networks_edm2.py:28,39) and
        # the attention kernels are written for that; heads of 8 / 16 / 32 channels (the reference's own consistency tests
        # use 16) run through the same kernels zero-padded to 64 (ops._AttentionHdFn, csrc/attention_hd.hip)
        ops._head_dim(channels, num_heads)
        self.attn_qkv = MPConv(channels, channels * 3, kernel=[1, 1])
        self.attn_qkv.weight.perm3 = True          # packed rows (m c s) -> (s m c): q | k | v contiguous
        self.attn_proj = MPConv(channels, channels, kernel=[1, 1])

    def _proj(self, x, o, clip, slot=None):
        t = self.attn_balance
        den = 1.0 / math.sqrt((1 - t) ** 2 + t ** 2)
        # proj conv with fused epilogue  out = clip((1-t)/den' * x + t/den' * conv(o))
        return self.attn_proj._cl(o, res=x, ta=(1 - t) * den, tb=t * den, clip=clip, res_slot=slot)

    @staticmethod
    def _slot(x):
        """x has two consumers here (attn_qkv and the residual of attn_proj).  The residual gradient comes first in
        backward: it is parked in a GradSlot and the qkv dgrad adds it in its epilogue (no torch add of the two)."""
        return ops.GradSlot() if (torch.is_grad_enabled() and x.requires_grad) else None

    def _frame_cl(self, x, clip):
        N, H, W, C = x.shape
        if not torch.is_grad_enabled() and x.is_cuda:          # eval: qkv convolution + normalisation in one launch
            o = ops.frame_attention_eval(x, self.attn_qkv.weight.pw, self.num_heads)
            return self._proj(x, o.reshape(N, H, W, C), clip)
        slot = self._slot(x)
        qkv = self.attn_qkv._cl(x, in_slot=slot).reshape(N, H * W, 3 * C)
        o = ops.attention_train(qkv, "frame", N, 1, self.num_heads)
        return self._proj(x, o.reshape(N, H, W, C), clip, slot)


class VideoAttention(_AttentionBase):
    def __init__(self, channels, num_heads, attn_balance=0.3):
        super().__init__(channels, num_heads, attn_balance)
        if num_heads == 0:
            return
        self.rope = RotaryEmbedding(channels // num_heads)
        self.train_mask = None

    def _cl(self, x, batch_size, cache=None, update_cache=False, just_2d=False, clip=0.0):
        """x (B*t, H, W, C) bf16 -> (mp_sum(x, attention(x)) [clipped], cache)."""
        if self.num_heads == 0:
            return (x.clamp(-clip, clip) if clip > 0 else x), None
        if just_2d:
            return self._frame_cl(x, clip), cache
        N, H, W, C = x.shape
        P = H * W
        self.__dict__["_tokens_per_frame"] = P               # (UNet.prewarm_eval sizes the next RoPE table from it)
        rope_bufs = (self.rope.inv_freq, self.rope.scale)
        if not self.training and not torch.is_grad_enabled() and x.is_cuda:
            o, cache = ops.attention_eval_x(x, self.attn_qkv.weight.pw, batch_size, self.num_heads, rope_bufs, cache,
                                            update_cache, P)
            return self._proj(x, o.reshape(N, H, W, C), clip), cache
        slot = self._slot(x)
        qkv = self.attn_qkv._cl(x, in_slot=slot).reshape(N, P, 3 * C)
        if self.training:
            T = N // (2 * batch_size)
            o = ops.attention_train(qkv, "video", batch_size, T, self.num_heads, rope_bufs)
        else:
            o, cache = ops.attention_eval(qkv, batch_size, self.num_heads, rope_bufs, cache, update_cache, P)
        return self._proj(x, o.reshape(N, H, W, C), clip, slot), cache

    def forward(self, x, batch_size, cache=None, update_cache=False, just_2d=False):
        if self.num_heads == 0:
            return x, None
        if _fp32.active():
            return _fp32.video_attention(self, x, batch_size, cache, update_cache, just_2d)
        with weights_ready(self):
            y, cache = self._cl(to_cl(x), batch_size, cache, update_cache, just_2d)
            return from_cl(y, x.dtype), cache


class FrameAttention(_AttentionBase):
    def _cl(self, x, batch_size=None, cache=None, update_cache=False, just_2d=True, clip=0.0):
        if self.num_heads == 0:
            return (x.clamp(-clip, clip) if clip > 0 else x), None
        return self._frame_cl(x, clip), None

    def forward(self, x, batch_size=None, cache=None, update_cache=False, just_2d=True):
        if self.num_heads == 0:
            return x, None
        if _fp32.active():
            return _fp32.frame_attention(self, x), None
        with weights_ready(self):
            return from_cl(self._frame_cl(to_cl(x), 0.0), x.dtype), None
