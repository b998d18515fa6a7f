"""A/B: FrameAttention core (forward + backward) at the bench shape (gym, B = 8, T = 64: N = 1024 frames of 16x16 tokens, 2 heads)
through the persistent work lists (ONIRIS_FRAME_KERNEL=1) vs the grid kernels (0)."""
import sys, time, torch
sys.path.insert(0, ".")
from autoregressive_diffusion_amd import ops
N, P, m = 1024, 256, 2
C = 64 * m
torch.manual_seed(0)
x = torch.randn(N, P, 3 * C, device="cuda", dtype=torch.bfloat16).requires_grad_(True)
go = torch.randn(N, P, C, device="cuda", dtype=torch.bfloat16)
res = {}
for ws in (1, 0, 1, 0):
    ops.FRAME_KERNEL = ws
    ops.FRAME_QKV_FUSED = ws
    for it in range(3):
        x.grad = None
        out = ops.attention_train(x, "frame", N, 1, m)
        out.backward(go)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fw = bw = 0.0
    for it in range(10):
        x.grad = None
        e0.record()
        out = ops.attention_train(x, "frame", N, 1, m)
        e1.record()
        out.backward(go)
        e2.record()
        torch.cuda.synchronize()
        fw += e0.elapsed_time(e1); bw += e1.elapsed_time(e2)
    res.setdefault(ws, []).append((fw / 10, bw / 10))
    if ws == 1:
        keep = (out.detach().float().clone(), x.grad.detach().float().clone())
    else:
        a, b = out.detach().float(), x.grad.detach().float()
        print("ws vs grid: rel out", ((keep[0] - a).norm() / a.norm()).item(), "rel dqkv", ((keep[1] - b).norm() / b.norm()).item())
fl = 4.0 * 64 * m * N * P * P
for ws, v in res.items():
    for fw, bw in v:
        print(f"FRAME_KERNEL={ws}: forward (incl. qkv norm) {fw*1e3:.0f} us, backward {bw*1e3:.0f} us; attention FLOPs fwd {fl/1e9:.1f} GF")
# which kernels, and their own durations (launch events: ops.KernelProfile)
for fk in (3, 1, 2, 0):
    ops.FRAME_KERNEL = min(fk, 1)
    ops.FRAME_BWD_FUSED = int(fk in (1, 3))
    ops.FRAME_QKV_FUSED = int(fk == 3)
    ops.census_start()
    ops.KernelProfile.start()
    for it in range(5):
        x.grad = None
        out = ops.attention_train(x, "frame", N, 1, m)
        out.backward(go)
    agg = ops.KernelProfile.stop()
    seen = ops.census_stop()
    print(f"FRAME_KERNEL={fk}:", {k: v for k, v in seen.items() if "attn" in k or "qkv" in k})
    for k, a in agg.items():
        print(f"    {k}: {a['ms'] / a['launches'] * 1e3:.1f} us per launch, {a['flops'] / a['ms'] / 1e9:.0f} TFLOP/s")
