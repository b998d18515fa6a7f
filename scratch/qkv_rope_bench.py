"""qkv_norm_rope forward / backward at the C2 shape: time per launch (events around 50 launches)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoregressive_diffusion_amd import ops
B, T, P, m = 2, 64, 64, 4
C, N = 64 * m, B * 2 * T
torch.manual_seed(0)
x = torch.randn(N, P, 3 * C, device="cuda").to(torch.bfloat16)
inv = (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).cuda()
sc = ((torch.arange(0, 64, 2) + 0.4 * 64) / (1.4 * 64)).cuda()
cs_, sn_, sc_ = ops.rope_tables(inv, sc, T, x.device)
q, k, v = (torch.empty(N, P, C, device="cuda", dtype=torch.bfloat16) for _ in range(3))
dq, dk, dv = (torch.randn(N, P, C, device="cuda").to(torch.bfloat16) for _ in range(3))
dqkv = torch.empty_like(x)
p, lib, st = ops._p, ops.lib, ops._stream
def fwd(): ops.check(lib.oniris_qkv_norm_rope(p(x), p(q), p(k), p(v), p(cs_), p(sn_), p(sc_), N * P, C, P, T, st()), "f")
def bwd(): ops.check(lib.oniris_qkv_norm_rope_bwd(p(x), p(dq), p(dk), p(dv), p(dqkv), p(cs_), p(sn_), p(sc_), N * P, C, P, T, st()), "b")
for name, f, mb in (("fwd", fwd, 2 * x.numel() * 2 / 1e6), ("bwd", bwd, 3 * x.numel() * 2 / 1e6)):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"qkv_norm_rope {name}: {us:.1f} us per launch, {mb / us:.2f} TB/s of {mb:.0f} MB", float(q.float().abs().mean()), float(dqkv.float().abs().mean()))
