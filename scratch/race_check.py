"""Race detector for the optional second stream of the weight gradients: forward + backward of the gym net on ONE input,
repeated; the conv-weight gradients (no atomics anywhere on their path) must be bit-identical from repeat to repeat."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autoregressive_diffusion_amd  # noqa: F401
from autoregressive_diffusion_amd import ops
from autoregressive_diffusion_amd.parallel import FlatParams
from edm2.networks_edm2 import UNet, Precond
from edm2.loss import EDM2Loss
from edm2.conv import NormalizedWeight
torch.manual_seed(0)
B, T = int(os.environ.get("RB", 2)), int(os.environ.get("RT", 16))
unet = UNet(64, 8, 4, 32, [1, 2, 4, 8], None, None, 2, [8], [16]).cuda()
net = Precond(unet, use_fp16=True, sigma_data=1.0).cuda().train()
with torch.no_grad():
    unet.out_gain.fill_(0.5)
flat = FlatParams(unet, lazy_small=True)
owned = [m.weight for m in unet.modules() if isinstance(m, NormalizedWeight)]
loss_fn = EDM2Loss(sigma_data=1.0)
g = torch.Generator().manual_seed(1)
images = torch.randn(B, T, 8, 64, 64, generator=g).cuda()
labels = torch.randint(0, 4, (B, T), generator=g).cuda()
sigma3 = (torch.randn(B, 2 * T, generator=g) * 1.0 + 0.4).exp().cuda()
sigma2 = sigma3[:, :T].contiguous()
eps3 = torch.randn(B, 2 * T, 8, 64, 64, generator=g).cuda()
eps2 = eps3[:, :T].contiguous()
def run(j2d):
    flat.zero_grad()
    loss, _ = loss_fn(net, images, labels, sigma=sigma2 if j2d else sigma3, just_2d=j2d, noise=eps2 if j2d else eps3, sync=False)
    loss.backward()
    flat.gather()
    torch.cuda.synchronize()
    return loss.item(), torch.cat([flat.slice_of(flat.grad, p).reshape(-1) for p in owned]).clone()
for _ in range(2):
    run(False); run(True)                      # weights reach the fixed point of the forced normalisation
ref = {j: run(j) for j in (False, True)}
bad = 0
n = int(os.environ.get("REPS", 30))
for i in range(n):
    j = (i % 4 == 0)
    l, gr = run(j)
    same = torch.equal(gr, ref[j][1])
    if not same or l != ref[j][0]:
        bad += 1
        d = (gr - ref[j][1]).abs()
        print(f"repeat {i} (2-D={j}): loss {l} vs {ref[j][0]}, {int((d > 0).sum())} gradient elements differ, max |diff| {d.max().item():.3e}")
print(f"ONIRIS_WGRAD_STREAM={ops.WGRAD_SIDE_STREAM}: {bad} of {n} repeats differ")
if os.environ.get("GRAPH"):
    # the same through graphs.GraphedStep (eager warm-up calls on the capture stream, then capture, then replays)
    from autoregressive_diffusion_amd.graphs import GraphedStep
    def body(j2d):
        flat.zero_grad()
        loss, _ = loss_fn(net, images, labels, sigma=sigma2 if j2d else sigma3, just_2d=j2d, noise=eps2 if j2d else eps3, sync=False)
        loss.backward()
        return loss
    gs = {j: GraphedStep(lambda j=j: body(j), params=flat.params, flat=flat, warmup=2) for j in (False, True)}
    bad = 0
    for i in range(n):
        j = (i % 4 == 0)
        l = gs[j]()
        flat.gather()
        torch.cuda.synchronize()
        gr = torch.cat([flat.slice_of(flat.grad, p).reshape(-1) for p in owned])
        if not torch.equal(gr, ref[j][1]):
            bad += 1
            d = (gr - ref[j][1]).abs()
            print(f"graphed call {i} (2-D={j}, replay={gs[j].graph is not None}): {int((d > 0).sum())} elements differ, max {d.max().item():.3e}")
    print(f"GraphedStep, ONIRIS_WGRAD_STREAM={ops.WGRAD_SIDE_STREAM}: {bad} of {n} calls differ from the eager reference")
