import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import torch
import test_model_gpu as M
from edm2.loss import EDM2Loss
from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
from autoregressive_diffusion_amd.graphs import GraphedStep
DEV = "cuda"
g = torch.Generator().manual_seed(79)
images = torch.randn(1, 4, 4, 32, 32, generator=g).to(DEV); labels = torch.randint(0, 4, (1, 4), generator=g).to(DEV)
net = M.build_precond(M.SMALL_CFG, 57, 1.0).train(); unet = net.unet
flat = FlatParams(unet, lazy_small=True); opt = FlatAdamW(flat, lr=1e-3, weight_decay=0.1); loss_fn = EDM2Loss(sigma_data=1.0)
dbg = {}
orig_fwd = unet.emb_noise.forward
def patched(x, gain=1):
    key = tuple(x.shape)
    if ("x",) + key not in dbg:
        dbg[("x",) + key] = torch.zeros_like(x); dbg[("dy",) + key] = torch.zeros(x.shape[0], 64, device=x.device)
    dbg[("x",) + key].copy_(x)
    y = orig_fwd(x, gain)
    if y.requires_grad:
        y.register_hook(lambda g, k=key: dbg[("dy",) + k].copy_(g))
    return y
unet.emb_noise.forward = patched
def fwd_bwd(j2d):
    opt.zero_grad(); loss, _ = loss_fn(net, images, labels, just_2d=j2d, sync=False); loss.backward(); return loss
graph = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = {j: (GraphedStep(lambda j=j: fwd_bwd(j), params=flat.params, flat=flat, warmup=1) if graph else (lambda j=j: fwd_bwd(j))) for j in (True, False)}
names = {id(p): n for n, p in unet.named_parameters()}
for k, j2d in enumerate([True, True, True, False, False, False, True]):
    loss = steps[j2d]()
    flat.gather(); torch.cuda.synchronize()
    bad = [names[id(p)] for p in flat.params if not torch.isfinite(flat.slice_of(flat.grad, p)).all()]
    bank = unet.__dict__["_oniris_bank"]
    for i, (w, _) in enumerate(bank.items):
        if w.param is unet.emb_noise.weight.weight:
            slab = w.taps * w.CoutP * w.CinP
            d0 = w.dwp[:slab]; d1 = w.dwp[slab:2 * slab]
            print("   emb_noise desc", i, "nsplit_cap", w.nsplit_cap, "slab0 max", float(d0.abs().max()), "slab1 max", float(d1.abs().max()) if w.nsplit_cap > 1 else None, "grouped", w.group is not None, "taps", w.taps, "cin", w.cin, "cout", w.cout)
    print("   dbg", {str(k): float(v.abs().max()) for k, v in dbg.items()})
    nf = (~torch.isfinite(flat.grad)).nonzero().flatten().tolist()
    big = (flat.grad.abs() > 1e6).nonzero().flatten().tolist()
    if nf or big:
        import bisect
        for idx in (nf + big)[:5]:
            i = bisect.bisect_right(flat.offsets, idx) - 1
            print("   bad element", idx, "value", float(flat.grad[idx]), "in", names[id(flat.params[i])], "offset", flat.offsets[i], "numel", flat.params[i].numel())
    print(k, "2d" if j2d else "3d", "loss", float(loss), "grad norm", float(flat.grad.norm()), "nonfinite grads:", bad[:6], len(bad))
    if os.environ.get('NOSTEP') is None: opt.step(max_norm=0.1)
    else: flat.take_active()
    torch.cuda.synchronize()
    badp = [names[id(p)] for p in flat.params if not torch.isfinite(p).all()]
    badv = [names[id(p)] for p in flat.params if not torch.isfinite(flat.slice_of(opt.v, p)).all()]
    print("   after step: nonfinite params", badp[:4], len(badp), "nonfinite v", badv[:4], len(badv), "runs steps", sorted(set(opt.param_steps)))
