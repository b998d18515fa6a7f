"""1x1 convs of the gym net at B = 8 (1024 frames): time and algorithmic TB/s per shape (forward = the dgrad of the transposed shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops
dev = "cuda"
N = 1024
for (hw, C, Co) in [(64, 64, 32), (64, 96, 32), (64, 32, 64), (64, 32, 96), (32, 64, 128), (32, 128, 64), (32, 192, 64), (32, 64, 192),
                    (16, 128, 384), (16, 128, 128), (16, 384, 128), (16, 256, 128), (16, 128, 256), (8, 256, 768), (8, 256, 256), (8, 512, 256), (8, 256, 512)]:
    p = torch.nn.Parameter(torch.randn(Co, C, 1, 1, device=dev))
    bank = ops.WeightBank(); pw = bank.add(p); bank.prepare(True)
    x = torch.randn(N, hw, hw, C, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(3): y = ops.conv(x, pw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = ops.conv(x, pw)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    mb = (x.numel() + y.numel()) * 2 / 1e6
    fl = 2.0 * N * hw * hw * C * Co
    print(f"{hw:2d}x{hw:<2d} {C:4d}->{Co:4d}  {mb:7.1f} MB {us:7.1f} us  {mb / us:5.2f} TB/s  {fl / us / 1e6:5.0f} TF  t_min {max(mb / 6.3, fl / 2.5e9):6.1f} us")
    del x, y
