"""A/B of the conv variants (ops.BIG_TILE): forward and dgrad-shaped launches, bit-compare against big_tile=0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoregressive_diffusion_amd import ops

def run(B, T, H, C, Cout, variants, iters=10, dgrad=False, epi=0):
    dev = "cuda"
    torch.manual_seed(0)
    p2 = torch.nn.Parameter(torch.randn(Cout, C, 3, 3, device=dev))
    p3 = torch.nn.Parameter(torch.randn(Cout, C, 2, 3, 3, device=dev))
    bank = ops.WeightBank(); pw2 = bank.add(p2); pw3 = bank.add(p3); bank.prepare(True)
    N = B * 2 * T
    x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    c3 = torch.randn(B * T, H, H, C, device=dev).to(torch.bfloat16)
    ca = torch.rand(N, device=dev) + 0.5; cb = torch.rand(N, device=dev) * 0.3
    res = {}
    for v in variants:
        ops.BIG_TILE = v
        out = torch.zeros(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
        y3 = torch.zeros(B * T, H, H, Cout, device=dev, dtype=torch.bfloat16)
        def go():
            if dgrad:
                ops._conv_launch(x, c3, pw2.wf, pw3.wf, out, ca, cb, B, 2, T, H, H, C, pw2.CinP, Cout, pw2.CoutP, 9,
                                 ctx_bstride=T, ctx_T=T, coff=(2, 1), ctx_fill=0.0)
            else:
                ops._conv_launch(x, x, pw2.wf, pw3.wf, out, ca, cb, B, 2, T, H, H, C, pw2.CinP, Cout, pw2.CoutP, 9,
                                 ctx_bstride=2 * T, ctx_T=T, coff=(-2, -1), ctx_fill=1.0, ctx_out=y3)
        for _ in range(2): go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): go()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * N * H * H * Cout * C * 9 * 2
        res[v] = (out.clone(), y3.clone())
        ok = ""
        if v != variants[0]:
            o0, y0 = res[variants[0]]
            ok = f"out_equal={torch.equal(o0, out)} y3_equal={torch.equal(y0, y3)} maxdiff={(o0.float()-out.float()).abs().max().item():.3g}"
        print(f"{'dgrad' if dgrad else 'fwd  '} B={B} T={T} H={H} C={C}->{Cout} big_tile={v}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF  {ok}", flush=True)

if __name__ == "__main__":
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,3,4").split(",")]
    for shp in [(2, 64, 32, 64, 64), (2, 64, 16, 128, 128), (2, 64, 64, 32, 32), (1, 5, 16, 64, 96), (2, 64, 8, 256, 256),
                (1, 5, 8, 64, 128), (3, 1, 8, 32, 64)]:
        run(*shp, variants)
        run(*shp, variants, dgrad=True)
