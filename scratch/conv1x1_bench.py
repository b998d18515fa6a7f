"""A/B of the 1x1 conv kernels (ops.BIG_TILE 0 = register-staged, 4 = LDS-DMA GEMM) at the gym net's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops
dev = "cuda"
for (M_hw, N, C, Co) in [(32, 256, 64, 64), (32, 256, 64, 128), (16, 256, 128, 384), (16, 256, 128, 128), (16, 256, 256, 128),
                         (8, 256, 256, 768), (8, 256, 256, 256), (8, 256, 512, 256)]:
    p = torch.nn.Parameter(torch.randn(Co, C, 1, 1, device=dev))
    bank = ops.WeightBank(); pw = bank.add(p); bank.prepare(True)
    x = torch.randn(N, M_hw, M_hw, C, device=dev).to(torch.bfloat16)
    res = {}
    for v in (0, 4):
        ops.BIG_TILE = v
        with torch.no_grad():
            for _ in range(3): y = ops.conv(x, pw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): y = ops.conv(x, pw)
            e1.record(); torch.cuda.synchronize()
        res[v] = (e0.elapsed_time(e1) / 20 * 1e3, y.float())
    mb = (x.numel() + res[0][1].numel()) * 2 / 1e6
    print(f"M={N*M_hw*M_hw:7d} {C:4d}->{Co:4d}  {mb:6.1f} MB  old {res[0][0]:6.1f} us ({mb/res[0][0]*1e-6*1e6/1e6*1e0:.2f} TB/s)  new {res[4][0]:6.1f} us ({mb/res[4][0]:.2f} TB/s)  maxdiff {float((res[0][1]-res[4][1]).abs().max()):.3g}")
