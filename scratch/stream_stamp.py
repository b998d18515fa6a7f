"""Per-phase cycle sums of conv_stream_kernel (diagnostic build: make stamp; ONIRIS_LIB_NAME=liboniris_hip_stamp.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops, _lib
from autoregressive_diffusion_amd._lib import lib, check
names = ["vm wait", "barrier1", "issue", "mfma", "xch write", "barrier2", "epilogue", "loop"]
def run(B, T, H, C, Cout, dgrad=False, epi=0):
    dev = "cuda"
    torch.manual_seed(0)
    p2 = torch.nn.Parameter(torch.randn(Cout, C, 3, 3, device=dev)); p3 = torch.nn.Parameter(torch.randn(Cout, C, 2, 3, 3, device=dev))
    bank = ops.WeightBank(); pw2 = bank.add(p2); pw3 = bank.add(p3); bank.prepare(True)
    N = B * 2 * T
    x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    c3 = torch.randn(B * T, H, H, C, device=dev).to(torch.bfloat16)
    ca = torch.rand(N, device=dev) + 0.5; cb = torch.rand(N, device=dev) * 0.3
    out = torch.zeros(N, H, H, Cout, device=dev, dtype=torch.bfloat16); y3 = torch.zeros(B * T, H, H, Cout, device=dev, dtype=torch.bfloat16)
    out2 = torch.zeros_like(out); res = torch.randn_like(out); esc = torch.rand(N, Cout, device=dev) + 0.5
    stamps = torch.zeros(64, dtype=torch.int64, device=dev)
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
    a.epi = epi
    if epi == _lib.EPI_EMB_SILU:
        a.escale, a.out2 = esc.data_ptr(), out2.data_ptr()
    if epi == _lib.EPI_MPSUM:
        a.res, a.out2, a.ta, a.tb, a.clip = res.data_ptr(), out2.data_ptr(), 0.9, 0.4, 256.0
    a.big_tile = 4
    a.splitk_ws, a.splitk_ws_bytes = stamps.data_ptr(), 0
    for _ in range(3): check(lib.oniris_conv_fwd(ctypes.byref(a), ops._stream()), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): check(lib.oniris_conv_fwd(ctypes.byref(a), ops._stream()), "conv")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * N * H * H * Cout * C * 9 * 2
    print(f"{'dgrad' if dgrad else 'fwd'} epi={epi} B={B} T={T} H={H} {C}->{Cout}: {us:.1f} us (stamped build) {fl / us / 1e6:.0f} TFLOP/s")
    s = stamps.view(8, 8).cpu()
    for w in (0, 1, 2, 3):
        tot = int(s[w].sum())
        print(f"  wave {w}: total {tot:7d} | " + " ".join(f"{n} {int(c)}" for n, c in zip(names, s[w])))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
run(B, 64, 64, 32, 32)
run(B, 64, 64, 32, 32, epi=_lib.EPI_EMB_SILU)
run(B, 64, 64, 32, 32, epi=_lib.EPI_MPSUM)
run(B, 64, 64, 32, 32, dgrad=True)
