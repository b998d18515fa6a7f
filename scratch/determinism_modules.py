"""Which module makes two training-mode forwards on the same input differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autoregressive_diffusion_amd  # noqa: F401
from autoregressive_diffusion_amd import ops
from edm2.attention import VideoAttention, FrameAttention
from edm2.conv import MPCausal3DGatedConv, MPConv
from edm2.networks_edm2 import Block
torch.manual_seed(0)
B, T, H = 2, 8, 16
def rep(name, f, n=4):
    with torch.no_grad():
        ys = [f().clone() for _ in range(n)]
    print(f"{name:40s}", [f"{(y.float() - ys[1].float()).std().item():.2e}" for y in ys[2:]], "(vs call 2; call 1 may precede the weight fixed point)", f"std {ys[1].float().std().item():.3f}")
att = VideoAttention(256, 4).cuda().train()
x = torch.randn(B * 2 * T, 256, H, H, device="cuda")
rep("VideoAttention train (16x16)", lambda: att(x, B)[0])
for pers in (0, 1):
    ops.ATTN_PERSISTENT = pers
    rep(f"VideoAttention train persistent={pers}", lambda: att(x, B)[0])
ops.ATTN_PERSISTENT = 1
rep("VideoAttention train just_2d", lambda: att(x, B, just_2d=True)[0])
conv = MPCausal3DGatedConv(64, 64, (3, 3, 3)).cuda().train()
xc = torch.randn(B * 2 * T, 64, 32, 32, device="cuda")
cn = torch.randn(B, 2 * T, device="cuda")
rep("MPCausal3DGatedConv train (32x32x64)", lambda: conv(xc, None, B, cn)[0])
rep("MPCausal3DGatedConv train just_2d", lambda: conv(xc, None, B, cn, just_2d=True)[0])
c1 = MPConv(64, 128, [1, 1]).cuda().train()
rep("MPConv 1x1 train", lambda: c1(xc))
blk = Block(64, 64, 256, flavor="enc", attention="video").cuda().train()
emb = torch.randn(B * 2 * T, 256, device="cuda")
x8 = torch.randn(B * 2 * T, 64, 8, 8, device="cuda")
rep("Block enc video-attn 8x8 train", lambda: blk(x8, emb, B, cn)[0])
