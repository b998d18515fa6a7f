"""Video-attention micro-benchmark at the C2 shape (B=2 [argv 2], T=64, P=64, 4 heads): forward + backward launches.  argv 1: repetitions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops
B, T, P, m = (int(sys.argv[2]) if len(sys.argv) > 2 else 2), 64, 64, 4
C, N = 64 * m, B * 2 * T
torch.manual_seed(0)
x = torch.randn(N, P, 3 * C, device="cuda").to(torch.bfloat16).requires_grad_(True)
inv = (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).cuda()
sc = ((torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)).cuda()
g = torch.randn(N, P, C, device="cuda").to(torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    out = ops.attention_train(x, "video", B, T, m, (inv, sc))
    out.backward(g)
torch.cuda.synchronize()
print("done", float(out.float().abs().mean()))
