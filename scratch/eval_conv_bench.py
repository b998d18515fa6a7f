"""One-frame (rollout) gated convs at the four gym levels; run under rocprofv3 --kernel-trace and summarise with
ktrace_sum.py.  Also checks split-K against the unsplit launch (ONIRIS_SPLITK=0 semantics) per shape."""
import sys, torch
sys.path.insert(0, ".")
from autoregressive_diffusion_amd import ops
from edm2.conv import MPCausal3DGatedConv
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (H, C) in [(64, 32), (32, 64), (16, 128), (8, 256)]:
    m = MPCausal3DGatedConv(C, C, [3, 3, 3]).to(dev).eval()
    x = torch.randn(1, C, H, H, device=dev)
    cn = torch.zeros(1, 1, device=dev)
    cache = {"activations": torch.randn(1, 2, H, H, C, device=dev).to(torch.bfloat16), "n_context_frames": 4}
    outs = {}
    with torch.no_grad():
        for sk in (0, 1):
            ops.SPLITK = sk
            for _ in range(10):
                y, _ = m(x, None, 1, cn, cache=dict(cache), update_cache=False)
            outs[sk] = y.float()
    torch.cuda.synchronize()
    d = (outs[0] - outs[1]).abs().max().item()
    print(f"H={H} C={C}: max |split - unsplit| = {d:.3e}, |y| max {outs[0].abs().max().item():.3f}", flush=True)
