"""Time the dominant gated-conv launches (forward and dgrad) at the gym net's MFMA-bound levels; run under different
ONIRIS_LIB_NAME builds to compare kernel variants.  usage: python scratch/mid_bench.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoregressive_diffusion_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = 64
dev = "cuda"
tot = 0.0
line = []
for (H, C, Cout) in [(32, 64, 64), (16, 128, 128), (8, 256, 256), (16, 256, 128), (32, 128, 64)]:
    torch.manual_seed(0)
    p2 = torch.nn.Parameter(torch.randn(Cout, C, 3, 3, device=dev)); p3 = torch.nn.Parameter(torch.randn(Cout, C, 2, 3, 3, device=dev))
    bank = ops.WeightBank(); pw2 = bank.add(p2); pw3 = bank.add(p3); bank.prepare(True)
    N = B * 2 * T
    x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    c3 = torch.randn(B * T, H, H, C, device=dev).to(torch.bfloat16)
    ca = torch.rand(N, device=dev) + 0.5; cb = torch.rand(N, device=dev) * 0.3
    out = torch.zeros(N, H, H, Cout, device=dev, dtype=torch.bfloat16); y3 = torch.zeros(B * T, H, H, Cout, device=dev, dtype=torch.bfloat16)
    for dgrad in (False, True):
        def go():
            if dgrad:
                ops._conv_launch(x, c3, pw2.wf, pw3.wf, out, ca, cb, B, 2, T, H, H, C, pw2.CinP, Cout, pw2.CoutP, 9,
                                 ctx_bstride=T, ctx_T=T, coff=(2, 1), ctx_fill=0.0)
            else:
                ops._conv_launch(x, x, pw2.wf, pw3.wf, out, ca, cb, B, 2, T, H, H, C, pw2.CinP, Cout, pw2.CoutP, 9,
                                 ctx_bstride=2 * T, ctx_T=T, coff=(-2, -1), ctx_fill=1.0, ctx_out=y3)
        for _ in range(3): go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        fl = 2.0 * N * H * H * Cout * C * 9 * 2
        tot += us
        line.append(f"{H}x{C}>{Cout}{'d' if dgrad else 'f'} {us:6.1f}us {fl / us / 1e6:4.0f}TF")
print(os.environ.get("ONIRIS_LIB_NAME", "default"), f"sum {tot:7.1f} us |", " | ".join(line))
