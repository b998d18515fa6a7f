"""VideoAttention backward at the C2 shape: persistent dQ kernel vs the grid kernel (per-kernel times, gradient agreement)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops
B, T, P, m = 2, 64, 64, 4
C, N = 64 * m, B * 2 * T
torch.manual_seed(0)
x = torch.randn(N, P, 3 * C, device="cuda").to(torch.bfloat16).requires_grad_(True)
inv = (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).cuda()
sc = ((torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)).cuda()
g = torch.randn(N, P, C, device="cuda").to(torch.bfloat16)
res = {}
for pers in (1, 0, 1):
    ops.ATTN_DQ_PERSISTENT = pers
    for _ in range(3):
        out = ops.attention_train(x, "video", B, T, m, (inv, sc)); out.backward(g)
    torch.cuda.synchronize()
    ops.KernelProfile.start()
    for _ in range(20):
        x.grad = None
        out = ops.attention_train(x, "video", B, T, m, (inv, sc)); out.backward(g)
    agg = ops.KernelProfile.stop()
    for k, v in agg.items():
        if "bwd" in k:
            print(f"dq_persistent={pers} {k:40s} {v['ms'] / v['launches'] * 1e3:8.1f} us  {v['flops'] / (v['ms'] * 1e-3) / 1e12:7.1f} TFLOP/s (algorithmic)")
    res[pers] = x.grad.float().clone()
print("finite:", bool(torch.isfinite(res[1]).all()), " dqkv rel diff persistent vs grid:", float((res[1] - res[0]).norm() / res[0].norm()))
