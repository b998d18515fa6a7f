"""In-kernel stamps of conv_eval1_kernel (diagnostic build: make variant VSRC=conv_fwd_s1ctx VNAME=e1stamp VDEF=-DEVAL1_STAMP;
ONIRIS_LIB_NAME=liboniris_hip_e1stamp.so): where the ~0.5 us per 32-channel phase of a one-frame gated conv go.  The stamps land in the
reserved emb_gain pointer of OnirisConvArgs ([workgroup][40] int64: compute wave 0 at [0..15], loader wave 4 at [16..31], s_memrealtime at 32, 33)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from autoregressive_diffusion_amd import ops
DEV = "cuda"
nhwc = lambda x: x.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16)
bfr = lambda x: x.to(torch.bfloat16).float()
def make_bank(params):
    bank = ops.WeightBank()
    return bank, [bank.add(p) for p in params]
for (H, cin, cout) in [(8, 256, 256), (16, 128, 128), (32, 64, 64)]:
    torch.manual_seed(0)
    B = 1
    p2 = torch.nn.Parameter(torch.randn(cout, cin, 3, 3).to(DEV)); p3 = torch.nn.Parameter(torch.randn(cout, cin, 2, 3, 3).to(DEV))
    bank, (pw2, pw3) = make_bank([p2, p3]); bank.prepare(training=False)
    x = nhwc(bfr(torch.randn(B, cin, H, H)))
    pad = bfr(torch.randn(B, cin, 2, H, H)).permute(0, 2, 3, 4, 1).to(DEV, torch.bfloat16).contiguous()
    g = (torch.rand(B) * 0.6 + 0.05).to(DEV)
    res = nhwc(bfr(torch.randn(B, cout, H, H)))
    kept = torch.zeros(B, H, H, cout, device=DEV)
    ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, ctx_prod=kept, ctx_prod_mode=1, res=res, ta=0.7, tb=0.5, clip=2.0)
    co = 16 if ((cout // 32) * (H // 8) ** 2 <= 128 and not (ops.BIG_TILE & 256)) else 32      # launch_conv_eval1's rule
    nwg = (H // 8) ** 2 * (cout // co)
    st = torch.zeros(nwg, 40, dtype=torch.int64, device=DEV)
    real = ops._conv_launch
    def launch(*a, **k):
        k["emb_gain"] = st
        return real(*a, **k)
    ops._conv_launch = launch
    # cold-ish conditions like the rollout: other kernels in between (a 64 MB fill evicts L2)
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    rows = []
    for it in range(6):
        junk.fill_(it)
        ops.gated_conv_eval(x, g, pw2, pw3, B, 1, pad, ctx_T=2, ctx_prod=kept, ctx_prod_mode=2, res=res, ta=0.7, tb=0.5, clip=2.0)
        torch.cuda.synchronize()
        rows.append(st.cpu().numpy().astype(np.float64).copy())
    ops._conv_launch = real
    s = np.median(np.stack(rows[2:]), axis=0)                     # [wg][40]
    clk = (s[:, 11] - s[:, 0]) / np.maximum(s[:, 33] - s[:, 32], 1) * 100.0     # MHz
    NP = cin // 32
    c = s[:, :16] - s[:, 0:1]; l = s[:, 16:32] - s[:, 0:1]
    print(f"=== {H}x{H} {cin}->{cout}: {nwg} workgroups of {co} output channels, {NP} phases; in-kernel clock {np.median(clk):.0f} MHz; cycles since the workgroup's first stamp (median over workgroups)")
    print("  loader : issue(0) out", int(np.median(l[:, 1])), " phase p landed:", [int(np.median(l[:, 2 + p])) for p in range(min(NP, 8))])
    print("  loader : own first stamp", int(np.median(l[:, 0])), " descriptors", int(np.median(l[:, 12])), " resources", int(np.median(l[:, 13])),)
    print("  compute: barrier_0 passed ", int(np.median(c[:, 1])), " phase p done  :", [int(np.median(c[:, 2 + p])) for p in range(min(NP, 8))])
    print("  compute, phase 2: barrier passed", int(np.median(c[:, 12])), " reads issued", int(np.median(c[:, 13])), " phase done", int(np.median(c[:, 4])))
    print("  E2 passed", int(np.median(c[:, 10])), " end", int(np.median(c[:, 11])), f" = {np.median(c[:, 11]) / np.median(clk):.2f} us")
