"""Persistent vs grid VideoAttention kernels at the C2 shape: equality of results and time per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops
B = int(os.environ.get("B", 2)); T = int(os.environ.get("T", 64)); P = int(os.environ.get("P", 64)); m = int(os.environ.get("M", 4))
C, N = 64 * m, B * 2 * T
torch.manual_seed(0)
x = torch.randn(N, P, 3 * C, device="cuda").to(torch.bfloat16)
inv = (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).cuda()
sc = ((torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)).cuda()
g = torch.randn(N, P, C, device="cuda").to(torch.bfloat16)
res = {}
for mode in (0, 1):
    ops.ATTN_PERSISTENT = mode
    xx = x.clone().requires_grad_(True)
    out = ops.attention_train(xx, "video", B, T, m, (inv, sc))
    out.backward(g)
    torch.cuda.synchronize()
    res[mode] = (out.detach().float(), xx.grad.float())
    ops.KernelProfile.start()
    for _ in range(20):
        xx.grad = None
        out = ops.attention_train(xx, "video", B, T, m, (inv, sc))
        out.backward(g)
    agg = ops.KernelProfile.stop()
    for k, v in agg.items():
        print(f"persistent={mode} {k}: {v['ms'] / v['launches'] * 1e3:.1f} us  {v['flops'] / (v['ms'] * 1e-3) / 1e12:.0f} TFLOP/s ({v['flops'] / (v['ms'] * 1e-3) / 2.5e15 * 100:.1f} % of peak)")
for i, name in enumerate(("out", "dqkv")):
    a, b = res[1][i], res[0][i]
    print(name, "rel L2 persistent vs grid:", ((a - b).norm() / b.norm()).item(), "max abs", (a - b).abs().max().item(), "finite", bool(torch.isfinite(b).all()))
