"""Per-phase cycle sums of the persistent attention forward (diagnostic build: make stamp; ONIRIS_LIB_NAME=liboniris_hip_stamp.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops, _lib
from autoregressive_diffusion_amd._lib import lib, check
B, T, P, m = 2, 64, 64, 4
C, N, L = 64 * m, B * 2 * T, 2 * T * P
torch.manual_seed(0)
q = torch.randn(B, L, C, device="cuda").to(torch.bfloat16)
q = (q.view(B, L, m, 64) / q.view(B, L, m, 64).float().norm(dim=-1, keepdim=True).to(torch.bfloat16) * 8).view(B, L, C).contiguous()
k, v = q.roll(1, 1).contiguous(), q.roll(2, 1).contiguous()
out = torch.empty_like(q); lse = torch.empty(B, m, L, device="cuda")
tabs = ops.device_tables("train", T, P, q.device)
a = ops._attn_args(q, k, v, None, None, None, out, lse, tabs, B, m, L, L, C, 2, P, T)
sched = ops._train_sched(T, P, B * m, q.device, "fwd")
a.sched, a.sched_wgs, a.sched_slots = sched[0].data_ptr(), sched[1], sched[2]
stamps = torch.zeros(8 * 8, dtype=torch.int64, device="cuda")
a.dkv_part = stamps.data_ptr()
a.dkv_chunks = int(os.environ.get('DBG', 0))
for _ in range(3):
    check(lib.oniris_attn_fwd(ctypes.byref(a), ops._stream()), "attn_fwd")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    check(lib.oniris_attn_fwd(ctypes.byref(a), ops._stream()), "attn_fwd")
e1.record(); torch.cuda.synchronize()
print("us per launch (stamped build)", e0.elapsed_time(e1) * 100)
names = ["B0 wait", "fill", "-", "steady blocks", "last 2 blocks", "E1 wait", "epilogue", "-"]
s = stamps.view(8, 8).cpu()
print("workgroup 0: items", [(int(e) >> 16, int(e) & 0xffff) for e in sched[0][0].cpu() if e >= 0], " (100 MHz ticks x 1)")
for w in range(8):
    tot = int(s[w].sum())
    print(f"wave {w}: total {tot:7d} | " + " ".join(f"{n} {int(c)}" for n, c in zip(names, s[w]) if n != "-"))
