"""Micro-benchmark of the gated conv at the 64x64 level (C = 32): forward, dgrad and weight-gradient launches through
KernelProfile (HIP events of the launches themselves).  usage: python scratch/c32_bench.py [B] [cin] [cout]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoregressive_diffusion_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 32
T, H = 64, 64
dev = torch.device("cuda")
torch.manual_seed(0)
p2 = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev))
p3 = torch.nn.Parameter(torch.randn(cout, cin, 2, 3, 3, device=dev))
bank = ops.WeightBank()
pw2, pw3 = bank.add(p2), bank.add(p3)
pw2.bank = pw3.bank = bank
bank.prepare(True)
N = B * 2 * T
x = torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16).requires_grad_(True)
g = (torch.rand(N, device=dev) * 0.6 + 0.05).requires_grad_(True)
gy = torch.randn(N, H, H, cout, device=dev).to(torch.bfloat16)
cs = torch.rand(N, cout, device=dev) + 0.5
res = torch.randn(N, H, H, cout, device=dev).to(torch.bfloat16)
for mode in ("none", "emb_silu", "mpsum"):
    kw = dict(cscale=cs) if mode == "emb_silu" else dict(res=res, ta=0.9, tb=0.4, clip=256.0) if mode == "mpsum" else {}
    for it in range(3):
        if it == 2:
            ops.KernelProfile.start()
        y = ops.gated_conv_train(x, g, pw2, pw3, B, T, **kw)
        y.backward(gy)
        bank.backward()
        x.grad = None
    agg = ops.KernelProfile.stop()
    for k, v in agg.items():
        print(f"{mode:9s} {k:60s} n={v['launches']} {v['ms'] / v['launches'] * 1e3:8.1f} us  {v['flops'] / (v['ms'] * 1e-3) / 1e12:7.1f} TF/s")
