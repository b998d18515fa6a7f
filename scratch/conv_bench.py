"""Micro-benchmark of single conv launches (used for rocprofv3 PMC runs and A/B of kernel variants)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops

def bench_gated(B, T, H, C, Cout, iters=20, wgrad=False):
    dev = "cuda"
    p2 = torch.nn.Parameter(torch.randn(Cout, C, 3, 3, device=dev))
    p3 = torch.nn.Parameter(torch.randn(Cout, C, 2, 3, 3, device=dev))
    bank = ops.WeightBank(); pw2 = bank.add(p2); pw3 = bank.add(p3); bank.prepare(True)
    N = B * 2 * T
    x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    ca = torch.rand(N, device=dev) + 0.5; cb = torch.rand(N, device=dev) * 0.3
    out = torch.empty(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
    y3 = torch.empty(B * T, H, H, Cout, device=dev, dtype=torch.bfloat16)
    def run():
        if wgrad:
            ops._wgrad_launch(x, out, pw2, ca, 1, N, H, H, C, pw2.CinP, Cout, pw2.CoutP, 9, N, N, 0, 0.0)
        else:
            ops._conv_launch(x, x, pw2.wf, pw3.wf, out, ca, cb, B, 2, T, H, H, C, pw2.CinP, Cout, pw2.CoutP, 9,
                             ctx_bstride=2 * T, ctx_T=T, coff=(-2, -1), ctx_fill=1.0, ctx_out=y3)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * N * H * H * Cout * C * 9 * (1 if wgrad else 2)
    print(f"{'wgrad' if wgrad else 'gconv'} B={B} T={T} H={H} C={C}->{Cout}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    shapes = [(2, 64, 64, 32, 32), (2, 64, 32, 64, 64), (2, 64, 16, 128, 128), (2, 64, 8, 256, 256)]
    for s in shapes:
        if which in ("all", "fwd"): bench_gated(*s, iters=iters)
        if which in ("all", "wgrad"): bench_gated(*s, iters=iters, wgrad=True)
