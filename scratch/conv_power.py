"""Is the gated conv kernel power/clock-limited?  Same launch on random vs all-zero operands (MI355X_MICROARCH: DVFS give-back)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops
def run(B, T, H, C, Cout, zero):
    dev = "cuda"; torch.manual_seed(0)
    p2 = torch.nn.Parameter(torch.randn(Cout, C, 3, 3, device=dev)); p3 = torch.nn.Parameter(torch.randn(Cout, C, 2, 3, 3, device=dev))
    bank = ops.WeightBank(); pw2 = bank.add(p2); pw3 = bank.add(p3); bank.prepare(True)
    N = B * 2 * T
    x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    if zero:
        x.zero_(); pw2.wf.zero_(); pw3.wf.zero_()
    ca = torch.rand(N, device=dev) + 0.5; cb = torch.rand(N, device=dev) * 0.3
    out = torch.zeros(N, H, H, Cout, device=dev, dtype=torch.bfloat16); y3 = torch.zeros(B * T, H, H, Cout, device=dev, dtype=torch.bfloat16)
    def go():
        ops._conv_launch(x, x, pw2.wf, pw3.wf, out, ca, cb, B, 2, T, H, H, C, pw2.CinP, Cout, pw2.CoutP, 9, ctx_bstride=2 * T, ctx_T=T, coff=(-2, -1), ctx_fill=1.0, ctx_out=y3)
    for _ in range(200): go()          # (long enough for the clock to settle)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 5
    print(f"H={H} {C}->{Cout} {'zeros ' if zero else 'random'}: {us:.1f} us  {2.0 * N * H * H * Cout * C * 18 / us / 1e6:.0f} TFLOP/s", flush=True)
for shp in [(2, 64, 32, 64, 64), (2, 64, 16, 128, 128)]:
    for z in (False, True): run(*shp, z)
