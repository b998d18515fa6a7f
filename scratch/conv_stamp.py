"""Per-phase cycle sums of the LDS-DMA gated conv kernel (diagnostic build: make stamp; ONIRIS_LIB_NAME=liboniris_hip_stamp.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops, _lib
from autoregressive_diffusion_amd._lib import lib, check
names = ["tile-top wait", "prologue", "own: dma issue", "mfma", "dma wait", "barrier", "epilogue", "ctx: dma issue", "mfma", "dma wait", "barrier", "-"]
def run(B, T, H, C, Cout, dgrad=False):
    dev = "cuda"
    torch.manual_seed(0)
    p2 = torch.nn.Parameter(torch.randn(Cout, C, 3, 3, device=dev)); p3 = torch.nn.Parameter(torch.randn(Cout, C, 2, 3, 3, device=dev))
    bank = ops.WeightBank(); pw2 = bank.add(p2); pw3 = bank.add(p3); bank.prepare(True)
    N = B * 2 * T
    x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    c3 = torch.randn(B * T, H, H, C, device=dev).to(torch.bfloat16)
    ca = torch.rand(N, device=dev) + 0.5; cb = torch.rand(N, device=dev) * 0.3
    out = torch.zeros(N, H, H, Cout, device=dev, dtype=torch.bfloat16); y3 = torch.zeros(B * T, H, H, Cout, device=dev, dtype=torch.bfloat16)
    stamps = torch.zeros(96, dtype=torch.int64, device=dev)
    a = _lib.ConvArgs()
    a.x, a.ctx, a.w_own, a.w_ctx, a.out = x.data_ptr(), (c3 if dgrad else x).data_ptr(), pw2.wf.data_ptr(), pw3.wf.data_ptr(), out.data_ptr()
    a.coef_own, a.coef_ctx = ca.data_ptr(), cb.data_ptr()
    a.B, a.S, a.T, a.H, a.W = B, 2, T, H, H
    a.Cin, a.CinP, a.Cout, a.CoutP, a.taps = C, pw2.CinP, Cout, pw2.CoutP, 9
    if dgrad:
        a.ctx_bstride, a.ctx_T, a.coff0, a.coff1, a.ctx_fill = T, T, 2, 1, 0.0
    else:
        a.ctx_bstride, a.ctx_T, a.coff0, a.coff1, a.ctx_fill = 2 * T, T, -2, -1, 1.0
        a.ctx_out = y3.data_ptr()
    a.big_tile = (7 if os.environ.get("NORES") else 4) | (64 if os.environ.get("NOW") else 0)
    a.splitk_ws, a.splitk_ws_bytes = stamps.data_ptr(), 0
    for _ in range(3): check(lib.oniris_conv_fwd(ctypes.byref(a), ops._stream()), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): check(lib.oniris_conv_fwd(ctypes.byref(a), ops._stream()), "conv")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * N * H * H * Cout * C * 9 * 2
    print(f"{'dgrad' if dgrad else 'fwd'} B={B} T={T} H={H} {C}->{Cout}: {us:.1f} us (stamped build) {fl / us / 1e6:.0f} TFLOP/s")
    s = stamps.view(8, 12).cpu()
    for w in (0, 3, 4, 7):
        tot = int(s[w].sum())
        print(f"  wave {w}: total {tot:7d} | " + " ".join(f"{n} {int(c)}" for n, c in zip(names, s[w]) if n != "-"))
for shp in [(int(os.environ.get("SB", "2")), 64, 32, 64, 64), (int(os.environ.get("SB", "2")), 64, 16, 128, 128)]:
    run(*shp)
