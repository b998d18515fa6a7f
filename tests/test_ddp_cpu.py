"""World-size-2 gloo tests (CPU) of the data-parallel host logic: flat buffers, end-of-backward all-reduce,
no_sync accumulation, parameters without gradient, fused-optimizer fallback math."""
import os
import socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

import cpu_reference_optimizer      # the product's FlatAdamW has no CPU arithmetic: the tests bring their own (also in the
cpu_reference_optimizer.install()   # spawned workers, which import this module)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 8)
        self.b = nn.Linear(8, 3)
        self.unused = nn.Linear(4, 4)          # never used in forward (cs_train.py:54 find_unused_parameters=True)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x))), None


class StagedNet(Net):
    """Same maths, plus the two hooks a module offers OnirisDDP for the early (overlapped) exchange: the parameters
    of `b` are final once the gradient of the hidden activation exists (UNet._oniris_overlap_plan / stage hook)."""
    stage_calls = 0

    def _oniris_overlap_plan(self):
        return "hidden", list(self.b.parameters())

    def forward(self, x):
        h = torch.tanh(self.a(x))
        cb = self.__dict__.get("_oniris_stage_cb")
        if cb is not None and h.requires_grad:
            def hook(g):
                StagedNet.stage_calls += 1
                return cb(g)
            h.register_hook(hook)
        return self.b(h), None


def _worker(rank, world, port, q, staged=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autoregressive_diffusion_amd.parallel import OnirisDDP, FlatAdamW
    torch.manual_seed(100 + rank)               # different init per rank: the wrapper must broadcast rank 0's
    net = StagedNet() if staged else Net()
    ddp = OnirisDDP(net, bucket_mb=1e-4, auto_wait=False)        # tiny buckets -> several all-reduces; wait() placed below
    assert ddp.flat.check()
    if staged:                                  # b's parameters sit at the end of the flat buffers
        assert ddp.flat.stage_at == "hidden" and 0 < ddp.flat.tail_start < ddp.flat.numel
        assert [id(p) for p in ddp.flat.params[-2:]] == [id(p) for p in net.b.parameters()]
    opt = FlatAdamW(ddp.flat, lr=1e-2, weight_decay=0.0)
    g = torch.Generator().manual_seed(7)
    data = torch.randn(4, 5, 6, generator=g)    # 4 micro-batches: rank r takes 2r, 2r+1
    # step 1: accumulate micro-batch 0 without sync, micro-batch 1 with sync
    opt.zero_grad()
    with ddp.no_sync():
        out, _ = ddp(data[2 * rank]); out.pow(2).mean().backward()
    out, _ = ddp(data[2 * rank + 1]); out.pow(2).mean().backward()
    if staged:                                  # the stage hook ran in both backwards, exchanged only in the synced one
        assert StagedNet.stage_calls == 2 and not any(ddp._sent) and len(ddp._works) > 2
    ddp.wait()
    grad = ddp.flat.grad.clone()
    opt.step()
    names = {id(p): n for n, p in net.named_parameters()}
    q.put((rank, {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}, grad.numpy().copy(),
           [(names[id(p)], o) for p, o in zip(ddp.flat.params, ddp.flat.offsets)]))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("staged", [False, True])
def test_ddp_gloo_world2(staged):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, staged)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sd0, g0, offs), (_, sd1, g1, _) = res
    g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
    assert torch.equal(g0, g1), "ranks disagree on the reduced gradient"
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), f"ranks diverged on {k}"
    # single-process reference: same init as rank 0, gradient = mean over ranks of (sum over the rank's 2 micro-batches)
    torch.manual_seed(100)
    ref = Net()
    g = torch.Generator().manual_seed(7)
    data = torch.randn(4, 5, 6, generator=g)
    for i in range(4):
        out, _ = ref(data[i]); (out.pow(2).mean() / 2).backward()
    params = dict(ref.named_parameters())
    for name, o in offs:
        p = params[name]
        got = g0[o:o + p.numel()].view_as(p)
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(got, want, atol=1e-6), "reduced gradient != mean of per-rank accumulated gradients"


def test_flat_adamw_matches_torch():
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
    torch.manual_seed(0)
    a, b = Net(), Net()
    b.load_state_dict(a.state_dict())
    flat = FlatParams(a)
    opt = FlatAdamW(flat, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    ref = torch.optim.AdamW([p for p in b.parameters()], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    for step in range(3):
        x = torch.randn(5, 6)
        opt.zero_grad(); ref.zero_grad()
        a(x)[0].pow(2).mean().backward(); b(x)[0].pow(2).mean().backward()
        for p in b.unused.parameters():
            p.grad = torch.zeros_like(p)
        opt.step(); ref.step()
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.allclose(v, w, atol=1e-6), k
    assert flat.check()


def test_flat_lazy_small_grads():
    """lazy_small: autograd-owned gradients are gathered into the flat buffer in one call, and a parameter that received
    NO gradient (`unused`: .grad stays None) is skipped by the optimizer exactly like torch.optim.AdamW skips it -- no
    weight decay, no moment update, no step count -- while its EMA copy still follows."""
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW, FlatEMA
    torch.manual_seed(1)
    a, b = Net(), Net()
    b.load_state_dict(a.state_dict())
    fa = FlatParams(a, lazy_small=True)
    assert len(fa._lazy) == len(fa.params)            # no HIP-owned conv weights in this toy net
    oa = FlatAdamW(fa, lr=1e-2, weight_decay=0.1)
    ob = torch.optim.AdamW(list(b.parameters()), lr=1e-2, weight_decay=0.1)
    ema = FlatEMA(fa, stds=(0.05,))
    unused0 = a.unused.weight.detach().clone()
    for step in range(3):
        oa.zero_grad(); ob.zero_grad()
        for _ in range(2):                             # two accumulated micro-batches
            x = torch.randn(5, 6)
            a(x)[0].pow(2).mean().backward(); b(x)[0].pow(2).mean().backward()
        fa.gather()
        for p, q in zip(a.parameters(), b.parameters()):
            if q.grad is not None:
                assert torch.allclose(p.grad, q.grad, atol=1e-7)
        oa.step(ema=ema.weights(8 * (step + 1), 8)); ob.step()
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.allclose(v, w, atol=1e-6), k
    assert torch.equal(a.unused.weight, unused0) and fa.check()
    sd = oa.state_dict()["state"]
    names = [n for n, _ in a.named_parameters()]
    assert sorted(names[i] for i in sd) == ["a.bias", "a.weight", "b.bias", "b.weight"]      # torch: no state for `unused`
    assert all(int(v["step"]) == 3 for v in sd.values())


def test_flat_adamw_state_dict_roundtrip_with_torch():
    """FlatAdamW.state_dict() has torch.optim.AdamW's layout (gym_train.py:76-81,137-138 save / resume through it):
    a torch optimizer's state loads into the flat one and both continue on the same trajectory; param_groups sets lr."""
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW
    torch.manual_seed(3)
    a, b = Net(), Net()
    b.load_state_dict(a.state_dict())
    ref = torch.optim.AdamW(list(b.parameters()), lr=1e-2, eps=1e-8, weight_decay=0.01)
    xs = torch.randn(6, 5, 6)

    def grads(net, x):
        net(x)[0].pow(2).mean().backward()
        for p in net.unused.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
    for i in range(2):                                     # two steps on the torch side only
        ref.zero_grad(); grads(b, xs[i]); ref.step()
    a.load_state_dict(b.state_dict())
    flat = FlatParams(a)
    opt = FlatAdamW(flat, lr=123.0)
    opt.load_state_dict(ref.state_dict())                  # lr, betas, eps, wd, moments, step come from the checkpoint
    assert opt.lr == 1e-2 and opt.steps == 2
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    for i, p in enumerate(b.parameters()):
        assert torch.equal(sd["state"][i]["exp_avg"], ref.state_dict()["state"][i]["exp_avg"])
    for g in opt.param_groups + ref.param_groups:          # the loops' lr schedule (gym_train.py:110-112)
        g["lr"] = 5e-3
    for i in range(2, 5):
        opt.zero_grad(); ref.zero_grad(); grads(a, xs[i]); grads(b, xs[i]); opt.step(); ref.step()
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.allclose(v, w, atol=1e-6), k
    opt2 = FlatAdamW(flat)
    opt2.load_state_dict(opt.state_dict())                 # own round trip
    assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v) and opt2.steps == opt.steps


def test_flat_ema_state_dict_layout():
    """FlatEMA.state_dict() = dict(stds, emas=[module.state_dict()-shaped dicts]) (edm2/phema.py:110-111)."""
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatEMA
    torch.manual_seed(4)
    net = Net()
    net.register_buffer("freqs", torch.randn(3))
    flat = FlatParams(net)
    ema = FlatEMA(flat, stds=(0.05, 0.1))
    with torch.no_grad():
        flat.flat.add_(1.0)
    for e, w in ema.weights(8, 4):
        e.lerp_(flat.flat, w)
    sd = ema.state_dict()
    assert sd["stds"] == [0.05, 0.1] and len(sd["emas"]) == 2
    assert list(sd["emas"][0].keys()) == list(net.state_dict().keys())
    assert torch.equal(sd["emas"][1]["freqs"], net.freqs)
    assert torch.equal(sd["emas"][0]["a.weight"], ema.view(0, net.a.weight))
    other = FlatEMA(flat, stds=(0.05, 0.1))
    other.load_state_dict(sd)
    assert all(torch.equal(other.view(k, p), ema.view(k, p)) for k in range(2) for p in flat.params)


class BufNet(Net):
    """Net with a random buffer (MPFourier's freqs / phases, utils.py:63-64) and autograd-owned small parameters."""
    def __init__(self):
        super().__init__()
        self.register_buffer("freqs", torch.randn(5))
        self.register_buffer("count", torch.zeros((), dtype=torch.int64))

    def forward(self, x):
        return self.b(torch.tanh(self.a(x))) * self.freqs[:3].sum(), None


def _worker_nosync(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autoregressive_diffusion_amd.parallel import OnirisDDP, FlatAdamW, FlatParams
    torch.manual_seed(200 + rank)               # different parameters AND buffers per rank
    net = BufNet()
    flat = FlatParams(net, lazy_small=True)     # (toy net: every gradient is autograd-owned, i.e. lazy)
    ddp = OnirisDDP(net, flat=flat)
    opt = FlatAdamW(flat, lr=1e-2, weight_decay=0.0)
    g = torch.Generator().manual_seed(9)
    data = torch.randn(2, 5, 6, generator=g)
    # the bench's --graph flow: forward+backward under no_sync (a replayed graph), then the exchange issued by hand
    for step in range(2):
        opt.zero_grad()
        with ddp.no_sync():
            out, _ = ddp(data[rank]); out.pow(2).mean().backward()
        ddp.allreduce_grads()
        ddp.wait()
        opt.step()
    q.put((rank, {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_no_sync_then_manual_exchange_keeps_ranks_equal():
    """Buffers follow rank 0 at construction, and gradients that autograd hands over lazily (lazy_small) take part
    in a hand-issued allreduce_grads() after a no_sync backward: parameters stay bit-equal across ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_nosync, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sd0), (_, sd1) = res
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), f"ranks diverged on {k}"
    torch.manual_seed(200)
    ref = BufNet()
    assert (sd0["freqs"] == ref.freqs.numpy()).all(), "buffers must be rank 0's"


def _run2(target, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + args) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _plain_bettermodule_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from torch.nn.parallel import DistributedDataParallel as DDP
    # a BetterModule WITHOUT kernel-written weights -- the reference's VAE takes the class from edm2.utils (vae/vae.py:13) and
    # is wrapped in torch DDP by its own training scripts -- is an ordinary torch model: torch DDP handles it alone
    from edm2.utils import BetterModule

    class PlainModel(BetterModule):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(4, 4)

        def forward(self, x):
            return self.lin(x)
    torch.manual_seed(1)
    plain = DDP(PlainModel())
    plain(torch.randn(2, 4)).sum().backward()
    ok = plain.module.lin.weight.grad is not None and "_oniris_inner_ddp" not in plain.module.__dict__ and not plain.parameters_to_ignore
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_torch_ddp_leaves_plain_bettermodules_alone():
    for rank, ok in _run2(_plain_bettermodule_worker):
        assert ok, rank


# ---- cs_train.py as written: torch's own DistributedDataParallel around the UNet

CS_ACCUM = 2
CS_STEPS = 7


def _cs_batches():
    g = torch.Generator().manual_seed(77)
    # [micro-step][rank] latents (B = 1, T = 2, 8 x 64 x 64)
    return [[torch.randn(1, 2, 8, 64, 64, generator=g) for _ in range(2)] for _ in range(CS_STEPS)]


def _cs_train_worker(rank, world, port, q, wrap_precond, slow_rank):
    """The body of cs_train.py:31-127 with its own names (synthetic latents instead of the streaming dataset + VAE statistics,
    7 micro-steps, accumulation 2 instead of 4): torch DDP, torch AdamW over precond.parameters(), EDM2Loss, no_sync() around
    backward only, the loss all-reduce, EMA copies of the whole Precond, the learning-rate schedule."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    import time
    from contextlib import nullcontext
    from torch.optim import AdamW
    from torch.nn.parallel import DistributedDataParallel as DDP
    torch.set_num_threads(3)
    unet = _build_gym_unet(400 + rank)                      # (ranks start different: construction must make them equal)
    from edm2.networks_edm2 import Precond
    from edm2.loss import EDM2Loss, learning_rate_schedule
    sigma_data = 1.
    if wrap_precond:                                        # (not what cs_train.py does; the other place a user may wrap)
        precond = DDP(Precond(unet, use_fp16=True, sigma_data=sigma_data), find_unused_parameters=True)
        unet_w, net = precond, precond.module
    else:
        unet_w = DDP(unet, find_unused_parameters=True)     # cs_train.py:54 (device_ids / output_device: GPU modules only)
        precond = net = Precond(unet_w, use_fp16=True, sigma_data=sigma_data)
    inner = (net if wrap_precond else unet).__dict__["_oniris_inner_ddp"]
    # torch's reducer holds exactly the parameters the inner engine does not (for the UNet: out_res.*); a root-level parameter
    # such as out_gain needs torch's ".name" spelling in the ignore list as well
    reducer_params = {id(p) for p in unet_w._module_parameters} & {id(p) for _, p in unet_w.module.named_parameters()}
    assert not (reducer_params & {id(p) for p in inner.flat.params})
    built = unet_w._build_params_for_reducer()[0]
    assert {id(p) for p in built} == {id(p) for p in unet_w.module.parameters() if p.requires_grad} - {id(p) for p in inner.flat.params}, \
        "torch's reducer and the inner engine overlap or leave a parameter to nobody"
    fired = []
    hooks = unet.__dict__["_oniris_stage_hooks"]
    for k in list(hooks):
        hooks[k] = (lambda cb, k: (lambda g: (fired.append(k), time.sleep(0.05 if rank == slow_rank else 0), cb(g))[2]))(hooks[k], k)
    loss_fn = EDM2Loss(P_mean=0.9, P_std=1.0, sigma_data=sigma_data, context_noise_reduction=0.1)
    ref_lr = 1e-2
    optimizer = AdamW(precond.parameters(), lr=ref_lr, eps=1e-4)
    optimizer.zero_grad()
    emas = [copy.deepcopy(precond) for _ in range(2)]       # PowerFunctionEMA(precond, stds=[0.050, 0.100]) (phema.py:95)
    assert all("_oniris_bank" not in m.__dict__ for e in emas for m in e.modules()), "copies carry no weight bank"
    data = _cs_batches()
    losses, exchanged, trace, gsum, gtrace = [], [], [], [], []
    for i in range(CS_STEPS):
        latents, actions = data[i][rank], None
        torch.manual_seed(1000 + 10 * i + rank)             # (the loss draws sigma and noise: same draws in the reference run)
        loss, un_weighted_loss = loss_fn(net if wrap_precond else precond, latents, actions, just_2d=i % 4 == 0) if not wrap_precond \
            else _loss_through(precond, loss_fn, latents, actions, i % 4 == 0)
        del fired[:]
        with (nullcontext() if i % CS_ACCUM == 0 else unet_w.no_sync()):
            loss.backward()
        exchanged.append(len(fired))
        # torch's rule: the forward above ran OUTSIDE no_sync(), so this backward is exchanged whether or not it ran inside
        # it -- after EVERY micro-step the kernel-owned gradients are the average over the ranks, like the ones torch's reducer holds
        gsum.append(float(inner.flat.grad.double().abs().sum()))
        gtrace.append(inner.flat.grad.clone().numpy() if i in (1, 2) else None)
        un_weighted_loss = torch.tensor(un_weighted_loss)
        dist.all_reduce(un_weighted_loss, op=dist.ReduceOp.SUM)
        losses.append(un_weighted_loss.item() / dist.get_world_size())
        if i % CS_ACCUM == 0 and i != 0:
            optimizer.step()
            optimizer.zero_grad()
            with torch.no_grad():
                for e, beta in zip(emas, (0.9, 0.99)):
                    for p_net, p_ema in zip(precond.parameters(), e.parameters()):
                        p_ema.lerp_(p_net, 1 - beta)
            for g in optimizer.param_groups:
                g["lr"] = learning_rate_schedule(i, ref_lr, 4, 4)
        trace.append(inner.flat.flat.clone())
    sd = {k: v.detach().clone().numpy() for k, v in unet.state_dict().items()}
    ema_sd = {k: v.detach().clone().numpy() for k, v in emas[0].state_dict().items()}
    q.put((rank, sd, ema_sd, losses, exchanged, len(inner.flat.params), len(inner.flat.stages),
           sorted(unet_w.parameters_to_ignore)[:3], inner.flat.check() or [p.grad is None for p in inner.flat.params].count(False) == 0,
           gsum, gtrace))
    dist.barrier()
    dist.destroy_process_group()


def _loss_through(precond_ddp, loss_fn, latents, actions, just_2d):
    """EDM2Loss calls net(...) and reads net.training / net.noise_weight: a DDP around the Precond forwards the call; the
    attributes are the module's."""
    class _View:
        training = property(lambda self: precond_ddp.module.training)
        noise_weight = property(lambda self: precond_ddp.module.noise_weight)
        sigma_data = property(lambda self: precond_ddp.module.sigma_data)

        def __call__(self, *a, **k):
            return precond_ddp(*a, **k)
    return loss_fn(_View(), latents, actions, just_2d=just_2d)


def _cs_train_reference():
    """ONE process, no wrapper: every micro-step's gradient is the mean of the two ranks' (what an averaging exchange gives)."""
    import copy
    from torch.optim import AdamW
    from edm2.networks_edm2 import Precond
    from edm2.loss import EDM2Loss, learning_rate_schedule
    unet = _build_gym_unet(400)
    precond = Precond(unet, use_fp16=True, sigma_data=1.)
    loss_fn = EDM2Loss(P_mean=0.9, P_std=1.0, sigma_data=1., context_noise_reduction=0.1)
    optimizer = AdamW(precond.parameters(), lr=1e-2, eps=1e-4)
    optimizer.zero_grad()
    ema = copy.deepcopy(precond)
    data = _cs_batches()
    losses = []
    for i in range(CS_STEPS):
        un = 0.0
        for r in range(2):
            torch.manual_seed(1000 + 10 * i + r)
            loss, u = loss_fn(precond, data[i][r], None, just_2d=i % 4 == 0)
            (loss / 2).backward()
            un += u / 2
        losses.append(un)
        if i % CS_ACCUM == 0 and i != 0:
            optimizer.step()
            optimizer.zero_grad()
            with torch.no_grad():
                for p_net, p_ema in zip(precond.parameters(), ema.parameters()):
                    p_ema.lerp_(p_net, 1 - 0.9)
            for g in optimizer.param_groups:
                g["lr"] = learning_rate_schedule(i, 1e-2, 4, 4)
    return unet.state_dict(), ema.state_dict(), losses


@pytest.mark.parametrize("wrap_precond,slow_rank", [(False, -1), (False, 1), (True, -1)])
def test_cs_train_loop_under_torch_ddp_world2(wrap_precond, slow_rank):
    """cs_train.py with ZERO changed lines -- `torch.nn.parallel.DistributedDataParallel(unet,
    find_unused_parameters=True)` (:10,54), `AdamW(precond.parameters())` (:76), `unet.no_sync()` around backward (:108).  The
    kernel-owned weights (every NormalizedWeight: 184 of the 449 parameters) and the parameters whose gradients the fused passes
    deliver (gates, emb_gain, out_gain: 261) are exchanged by the inner OnirisDDP the UNet installs when torch's constructor asks
    for `_ddp_params_and_buffers_to_ignore`; torch's reducer keeps the four parameters of `out_res`, which nothing uses.  Ranks must be
    bit-equal and equal to one process that averages the two ranks' gradients; stage hooks fire on every backward but exchange
    only on the synced ones; slow_rank = 1: that rank's stage hooks run late by 50 ms each (the staged collectives are issued in
    different wall-clock order on the two ranks -- VERDICT next #7c), same result."""
    res = _run2(_cs_train_worker, wrap_precond, slow_rank)
    (_, sd0, ema0, l0, ex0, n0, st0, ign0, ok0, gs0, gt0), (_, sd1, ema1, l1, ex1, n1, st1, ign1, ok1, gs1, gt1) = res
    # after every backward -- the ones inside no_sync() too (i = 1 is one) -- both ranks hold the same, averaged kernel-owned gradients
    assert gs0 == gs1 and all(g > 0 for g in gs0), (gs0, gs1)
    assert all((a is None) == (b is None) and (a is None or (a == b).all()) for a, b in zip(gt0, gt1))
    assert n0 == n1 == 449 - 4 and st0 == st1 >= 4 and ign0 == ign1 and len(ign0) == 3      # (everything but out_res.*)
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), f"ranks diverged on {k}"
    for k in ema0:
        assert (ema0[k] == ema1[k]).all(), f"EMA copies diverged on {k}"
    assert l0 == l1
    assert ex0 == ex1 and all(e == st0 for e in ex0), ex0          # every stage hook, every backward (also under no_sync)
    import cpu_ops_stub
    try:
        ref_sd, ref_ema, ref_losses = _cs_train_reference()
    finally:
        cpu_ops_stub.uninstall()
    for k, v in ref_sd.items():
        a = torch.from_numpy(sd0[k])
        assert torch.allclose(a, v, atol=5e-6, rtol=1e-5), (k, (a - v).abs().max().item())
    ema0 = {k.replace("module.", ""): v for k, v in ema0.items()}          # (the copies keep the DDP wrapper in their tree)
    assert set(ema0) == set(ref_ema)
    for k, v in ref_ema.items():
        a = torch.from_numpy(ema0[k])
        assert torch.allclose(a, v, atol=5e-6, rtol=1e-5), (k, (a - v).abs().max().item())
    assert max(abs(a - b) / abs(b) for a, b in zip(l0, ref_losses)) < 1e-6, (l0, ref_losses)


def _active_mismatch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autoregressive_diffusion_amd.parallel import OnirisDDP, FlatAdamW, FlatParams
    torch.manual_seed(5)
    net = Net()
    ddp = OnirisDDP(net, flat=FlatParams(net, lazy_small=True))
    opt = FlatAdamW(ddp.flat, lr=1e-2)
    x = torch.randn(5, 6)
    outcome = []
    for step in range(2):
        opt.zero_grad()
        out, _ = ddp(x)
        loss = out.pow(2).mean()
        if step == 1 and rank == 1:                      # rank 1 alone also uses `unused`: a different step kind
            loss = loss + net.unused(torch.randn(3, 4)).pow(2).mean()
        loss.backward()
        ddp.wait()
        try:
            opt.step()
            outcome.append("ok")
        except RuntimeError as e:
            outcome.append("raised" if "disagree" in str(e) else "other: " + str(e))
    q.put((rank, outcome))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_skip_different_parameters_are_detected():
    """take_active() is rank-local bookkeeping while the exchange averages the whole buffer -- a rank that
    skipped a parameter another rank updated would silently diverge; OnirisDDP compares the bitmaps and raises."""
    for rank, outcome in _run2(_active_mismatch_worker):
        assert outcome == ["ok", "raised"], (rank, outcome)


class MultiStageNet(nn.Module):
    """Three layers; the module offers TWO early stages (c's parameters are final when the gradient of h2 exists, b's when
    that of h1 does) through the hook dictionary OnirisDDP installs (UNet._oniris_overlap_stages / _oniris_stage_hooks)."""
    fired = []

    def __init__(self):
        super().__init__()
        self.a, self.b, self.c = nn.Linear(6, 16), nn.Linear(16, 16), nn.Linear(16, 3)
        self.unused = nn.Linear(4, 4)

    def _oniris_overlap_stages(self):
        return [("h2", list(self.c.parameters())), ("h1", list(self.b.parameters()))]

    def forward(self, x):
        hooks = self.__dict__.get("_oniris_stage_hooks") or {}
        h1 = torch.tanh(self.a(x))
        h2 = torch.tanh(self.b(h1))
        for key, h in (("h1", h1), ("h2", h2)):
            cb = hooks.get(key)
            if cb is not None and h.requires_grad:
                h.register_hook(lambda g, key=key, cb=cb: (MultiStageNet.fired.append(key), cb(g))[1])
        return self.c(h2), None


def _modes_worker(rank, world, port, q, exchange, bf16, clip):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autoregressive_diffusion_amd.parallel import OnirisDDP, FlatAdamW, FlatParams, FlatEMA
    torch.manual_seed(200 + rank)
    net = MultiStageNet()
    flat = FlatParams(net)          # (autograd-owned gradients must sit in the flat buffer when a stage fires: not lazy)
    ddp = OnirisDDP(net, flat=flat, bucket_mb=1e-4, exchange=exchange, grad_dtype=torch.bfloat16 if bf16 else None)
    assert [k for k, _, _ in flat.stages] == ["h2", "h1"] and flat.head[1] == flat.stages[0][1]
    assert all(lo % FlatParams.SEG_ALIGN == 0 and hi % FlatParams.SEG_ALIGN == 0 for _, lo, hi in flat.stages)
    opt = FlatAdamW(flat, lr=1e-2, weight_decay=0.01)
    ema = FlatEMA(flat, stds=(0.05,))
    g = torch.Generator().manual_seed(9)
    data = torch.randn(3, 2, 5, 6, generator=g)          # [step][rank][batch][features]
    for step in range(3):
        opt.zero_grad()
        MultiStageNet.fired.clear()
        out, _ = ddp(data[step, rank]); out.pow(2).mean().backward()
        assert MultiStageNet.fired == ["h2", "h1"] and not any(ddp._sent)
        ddp.wait()
        opt.step(max_norm=clip, ema=ema.weights(8 * (step + 1), 8))
    q.put((rank, {k: v.detach().numpy().copy() for k, v in net.state_dict().items()},
           flat.flat.numpy().copy(), [tuple(r) for r in getattr(flat, "_owned_ranges", [])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,bf16,clip", [("allreduce", False, None), ("allreduce", True, None), ("mesh", False, None),
                                                ("mesh", True, None), ("mesh", False, 0.05), ("allreduce", False, 0.05)])
def test_ddp_stages_and_exchange_modes(exchange, bf16, clip):
    """Multi-stage overlap with every exchange form: ring/tree all-reduce per stage, bf16 transport, and the mesh form
    (all-to-all reduce-scatter, optimizer on the owned chunks, parameter all-gather) -- ranks stay bit-equal, and equal a
    single process that averages the two ranks' gradients (torch.optim.AdamW + clip_grad_norm_)."""
    res = _run2(_modes_worker, exchange, bf16, clip)
    (_, sd0, p0, own0), (_, sd1, p1, own1) = res
    assert (p0 == p1).all(), "ranks diverged"
    if exchange == "mesh":
        assert own0 and own1 and own0 != own1 and all(a[1] == b[0] for a, b in zip(own0, own1))    # rank 0's chunk, then rank 1's
    torch.manual_seed(200)
    ref = MultiStageNet()
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.01)
    g = torch.Generator().manual_seed(9)
    data = torch.randn(3, 2, 5, 6, generator=g)
    for step in range(3):
        topt.zero_grad()
        for r in range(2):
            out, _ = ref(data[step, r]); (out.pow(2).mean() / 2).backward()
        for p in ref.unused.parameters():              # (a permanent zero .grad view counts as "has a gradient")
            p.grad = torch.zeros_like(p)
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(list(ref.parameters()), clip)
        topt.step()
    tol = 2e-3 if bf16 else 2e-6
    for k, v in ref.state_dict().items():
        assert torch.allclose(torch.from_numpy(sd0[k]), v, atol=tol), (k, (torch.from_numpy(sd0[k]) - v).abs().max())


def _mesh_state_worker(rank, world, port, q, exchange):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autoregressive_diffusion_amd.parallel import OnirisDDP, FlatAdamW, FlatParams, FlatEMA
    torch.manual_seed(300 + rank)
    net = MultiStageNet()
    flat = FlatParams(net)
    ddp = OnirisDDP(net, flat=flat, bucket_mb=1e-4, exchange=exchange)
    opt = FlatAdamW(flat, lr=1e-2, weight_decay=0.01)
    ema = FlatEMA(flat, stds=(0.05, 0.1))
    g = torch.Generator().manual_seed(11)
    data = torch.randn(7, 2, 5, 6, generator=g)           # [micro-step][rank][batch][features]
    K = 3
    # the reference loop's cadence (cs_train.py:105-127): a synced backward at i % K == 0 -- at i = 0 with NO optimizer step
    # and NO zero_grad behind it -- and no_sync() accumulation in between
    import contextlib
    for i in range(7):
        sync = i % K == 0
        with (contextlib.nullcontext() if sync else ddp.no_sync()):
            out, _ = ddp(data[i, rank]); out.pow(2).mean().backward()
        if sync:
            ddp.wait()
            if i != 0:
                opt.step(ema=ema.weights(8 * i, 8 * K))
                opt.zero_grad()
    refused = False
    if exchange == "mesh":
        try:
            opt.state_dict()
        except RuntimeError as e:
            refused = "gather_state" in str(e)
    ddp.gather_state(opt)                                 # collective
    osd = opt.state_dict() if rank == 0 else None         # ... then rank 0 alone writes the checkpoint
    esd = ema.state_dict() if rank == 0 else None
    q.put((rank, refused, flat.flat.numpy().copy(),
           None if osd is None else {i: {k: v.numpy().copy() for k, v in s.items()} for i, s in osd["state"].items()},
           None if esd is None else [{k: v.numpy().copy() for k, v in sd.items()} for sd in esd["emas"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["mesh", "allreduce"])
def test_mesh_mode_checkpoint_state_and_accumulation_cadence(exchange):
    """(1) with exchange="mesh" the optimizer runs on the owned chunks only -- the EMA copies must still be
    complete on every rank and the Adam moments after gather_state(), so that rank 0's checkpoint (cs_train.py:146-159) equals
    a single process's; (2) an exchange that is not followed by zero_grad (micro-step 0 of the reference loop) must not
    leave reduced values in the local gradient buffer: both exchange forms give avg(sum of micro-gradients)."""
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW, FlatEMA
    res = _run2(_mesh_state_worker, exchange)
    (_, ref0, p0, osd, esd), (_, ref1, p1, _, _) = res
    assert (p0 == p1).all(), "ranks diverged"
    if exchange == "mesh":
        assert ref0 and ref1, "state_dict() before gather_state() must be refused in mesh mode"
    # single process: same cadence on the mean of the two ranks' gradients, this repository's optimizer classes unsharded
    torch.manual_seed(300)
    net = MultiStageNet()
    flat = FlatParams(net)
    opt = FlatAdamW(flat, lr=1e-2, weight_decay=0.01)
    ema = FlatEMA(flat, stds=(0.05, 0.1))
    g = torch.Generator().manual_seed(11)
    data = torch.randn(7, 2, 5, 6, generator=g)
    for i in range(7):
        for r in range(2):
            out, _ = net(data[i, r]); (out.pow(2).mean() / 2).backward()
        if i % 3 == 0 and i != 0:
            opt.step(ema=ema.weights(8 * i, 24))
            opt.zero_grad()
    assert torch.allclose(torch.from_numpy(p0), flat.flat, atol=2e-6), (torch.from_numpy(p0) - flat.flat).abs().max()
    want_o, want_e = opt.state_dict(), ema.state_dict()
    assert set(osd) == set(want_o["state"])
    for i, s in want_o["state"].items():
        for k in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.allclose(torch.from_numpy(osd[i][k]), s[k], atol=2e-6), (i, k)
    for got, want in zip(esd, want_e["emas"]):
        for k, v in want.items():
            assert torch.allclose(torch.from_numpy(got[k]), v, atol=2e-6), k


def test_ddp_guard_is_invisible_to_generic_introspection():
    """the torch-DDP refusal must not turn hasattr / inspect.getmembers / attribute copying on a UNet into a
    RuntimeError about DistributedDataParallel when nobody is wrapping anything."""
    import inspect
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from edm2.networks_edm2 import UNet
    unet = UNet(img_resolution=16, img_channels=4, label_dim=4, model_channels=8, channel_mult=[1, 2], num_blocks=1)
    assert not hasattr(unet, "_ddp_params_and_buffers_to_ignore")
    assert getattr(unet, "_ddp_params_and_buffers_to_ignore", None) is None
    names = [n for n, _ in inspect.getmembers(unet)]
    assert "forward" in names and "_ddp_params_and_buffers_to_ignore" not in names


# ---------------------------------------------------------------------------------------------------------------------
# the REAL module tree (gym UNet, 449 parameters, its overlap stages and 2-D / 3-D parameter classes) under OnirisDDP with two
# gloo ranks; the HIP ops are replaced by CPU stand-ins with the same interface and autograd topology (tests/cpu_ops_stub.py)

# gym_train.py:37-47 at HALF the width (model_channels 16 instead of 32: the same 449 parameter tensors, module names, overlap
# stages and parameter classes with a quarter of the elements -- the CPU suite has minutes, and the layout logic under test
# counts tensors, not elements; the attention levels keep whole 64-channel heads: 64 ch at 16x16, 128 ch at 8x8)
GYM_CFG = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=16, channel_mult=[1, 2, 4, 8],
               channel_mult_noise=None, channel_mult_emb=None, num_blocks=2, video_attn_resolutions=[8],
               frame_attn_resolutions=[16])


def _build_gym_unet(seed):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cpu_ops_stub
    cpu_ops_stub.install()
    from edm2.networks_edm2 import UNet
    torch.manual_seed(seed)
    unet = UNet(**GYM_CFG).train()
    for m in unet.modules():
        if hasattr(m, "emb_gain"):
            nn.init.constant_(m.emb_gain, 0.3)
    nn.init.constant_(unet.out_gain, 1.0)
    return unet


def _gym_batches():
    g = torch.Generator().manual_seed(21)
    # [micro-step][rank]: (x, c_noise, labels) -- B = 1 sequence of 2 frames (clean | noised: 4 slots; 2 in a 2-D step)
    return [[(torch.randn(1, 2 if j2d else 4, 8, 64, 64, generator=g), torch.randn(1, 2 if j2d else 4, generator=g),
              torch.randint(0, 4, (1, 2 if j2d else 4), generator=g)) for _ in range(2)] for j2d in (True, False, False, False)]


def _real_tree_worker(rank, world, port, q, exchange, bf16, mismatch):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autoregressive_diffusion_amd.parallel import OnirisDDP, FlatAdamW, FlatParams, FlatEMA
    torch.set_num_threads(3)                             # (two ranks share the container's 8 cores)
    unet = _build_gym_unet(400 + rank)                   # different init per rank: construction broadcasts rank 0's
    flat = FlatParams(unet, lazy_small=True)             # (bench.py's construction)
    ddp = OnirisDDP(unet, flat=flat, exchange=exchange, grad_dtype=torch.bfloat16 if bf16 else None, auto_wait=False)
    opt = FlatAdamW(flat, lr=1e-3, weight_decay=0.01)
    ema = FlatEMA(flat, stds=(0.05,))
    keys = [k for k, _, _ in flat.stages]
    fired = []
    hooks = unet.__dict__["_oniris_stage_hooks"]
    for k in list(hooks):
        hooks[k] = (lambda cb, k: (lambda g: (fired.append(k), cb(g))[1]))(hooks[k], k)
    data = _gym_batches()
    outcome = "ok"
    try:
        for i, j2d in enumerate((True, False, False, False)):
            x, cn, lab = data[i][rank]
            if mismatch and rank == 1 and i == 1:            # rank 1 runs a 2-D step where rank 0 runs a 3-D one
                x, cn, lab, j2d = x[:, :2], cn[:, :2], lab[:, :2], True
            opt.zero_grad()
            del fired[:]
            if i == 2:                                       # one accumulated micro-step under no_sync() first
                with ddp.no_sync():
                    out, _ = ddp(x, cn, lab, just_2d=j2d); out.float().pow(2).mean().backward()
                assert fired == keys and not ddp._works, "stage hooks fire under no_sync() but exchange nothing"
                del fired[:]
            out, _ = ddp(x, cn, lab, just_2d=j2d); out.float().pow(2).mean().backward()
            assert fired == keys, (fired, keys)              # every stage, in backward order, exactly once
            assert not any(ddp._sent) and len(ddp._works) >= len(keys) + 1      # ... each started its exchange; + the head
            ddp.wait()
            opt.step(max_norm=0.5, ema=ema.weights(8 * (i + 1), 8))
    except RuntimeError as e:
        outcome = "raised" if "ranks disagree" in str(e) else f"error: {e}"
    steps = {flat.names[id(p)]: s for p, s in zip(flat.params, opt.param_steps)}
    q.put((rank, outcome, flat.flat.numpy().copy(), steps, len(flat.params), [(str(k), lo, hi) for k, lo, hi in flat.stages]))
    if outcome == "ok":
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,bf16", [("allreduce", False), ("mesh", True)])
def test_real_unet_module_tree_under_ddp_world2(exchange, bf16):
    """VERDICT r03 next #6a: hook ordering, stage firing, no_sync(), the active-bitmap check and the flat layout on the
    gym UNet's own 449 parameters (not a toy net), two gloo ranks, against ONE process that averages both ranks' gradients."""
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW, FlatEMA
    res = _run2(_real_tree_worker, exchange, bf16, False)
    (_, o0, p0, steps0, n0, st0), (_, o1, p1, steps1, n1, st1) = res
    assert o0 == o1 == "ok", (o0, o1)
    assert n0 == n1 == 449 - 5 + 5 and len(st0) >= 4 and st0 == st1        # all 449 are re-homed; >= 4 overlap stages
    assert (p0 == p1).all(), "ranks diverged"
    assert steps0 == steps1
    # parameter classes: own-frame weights step on all 4 steps, context weights / gates only on the 3 3-D steps, out_res /
    # emb_time never (networks_edm2.py:197,205-207)
    assert steps0["enc.64x64_conv.last_frame_conv.weight.weight"] == 4 and steps0["enc.64x64_conv.weight.weight"] == 3
    assert steps0["enc.64x64_conv.gating.mult"] == 3 and steps0["out_res.mult"] == 0 and steps0["emb_time.weight.weight"] == 0
    # single process, same stub ops: mean of the two ranks' gradients per step (accumulated micro-step included)
    import cpu_ops_stub
    try:
        _real_tree_reference(p0, bf16)
    finally:
        cpu_ops_stub.uninstall()                         # (this process runs other test files afterwards)


def _real_tree_reference(p0, bf16):
    from autoregressive_diffusion_amd.parallel import FlatParams, FlatAdamW, FlatEMA
    unet = _build_gym_unet(400)
    flat = FlatParams(unet, lazy_small=True)
    opt = FlatAdamW(flat, lr=1e-3, weight_decay=0.01)
    ema = FlatEMA(flat, stds=(0.05,))
    data = _gym_batches()
    for i, j2d in enumerate((True, False, False, False)):
        opt.zero_grad()
        for r in range(2):
            x, cn, lab = data[i][r]
            for _ in range(2 if i == 2 else 1):
                out, _ = unet(x, cn, lab, just_2d=j2d); (out.float().pow(2).mean() / 2).backward()
        opt.step(max_norm=0.5, ema=ema.weights(8 * (i + 1), 8))
    tol = 3e-3 if bf16 else 1e-5
    d = (torch.from_numpy(p0) - flat.flat).abs().max().item()
    assert d <= tol, d


def test_real_unet_ranks_running_different_step_kinds_are_detected():
    res = _run2(_real_tree_worker, "allreduce", False, True)
    assert sorted(r[1] for r in res) == ["raised", "raised"], [r[1] for r in res]


def _torch_optimizer_worker(rank, world, port, q, staged):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autoregressive_diffusion_amd.parallel import OnirisDDP as DDP
    import contextlib
    torch.manual_seed(400 + rank)
    net = MultiStageNet() if staged else Net()
    # cs_train.py:53-54, with the one changed line `DDP = OnirisDDP`
    ddp = DDP(net, device_ids=[0], output_device=0, find_unused_parameters=True, bucket_mb=1e-4)
    optimizer = torch.optim.AdamW(ddp.parameters(), lr=1e-2, eps=1e-4)       # cs_train.py:76: a plain torch optimizer
    optimizer.zero_grad()                                                      # :77 (set_to_none=True: every .grad is gone)
    g = torch.Generator().manual_seed(13)
    data = torch.randn(9, 2, 5, 6, generator=g)                               # [micro-step][rank][batch][features]
    K = 2
    none_seen = []
    for i in range(9):
        out, _ = ddp(data[i, rank])
        with (contextlib.nullcontext() if i % K == 0 else ddp.no_sync()):      # :108
            out.pow(2).mean().backward()
        if i % K == 0 and i != 0:                                              # :117-121: no wait() anywhere
            torch.nn.utils.clip_grad_norm_(ddp.parameters(), 0.5)
            none_seen.append([n for n, p in net.named_parameters() if p.grad is None])
            optimizer.step()
            optimizer.zero_grad()
    q.put((rank, {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}, none_seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("staged", [False, True])
def test_ddp_with_a_torch_optimizer_in_the_reference_loop(staged):
    """cs_train.py's loop with `DDP = OnirisDDP` and NOTHING else changed: torch.optim.AdamW, its zero_grad() (set_to_none: the
    flat gradient views are released and the next backward creates gradients outside the flat buffer), no_sync() accumulation, a
    synced micro-step 0 that no optimizer step follows, no wait() call, clip_grad_norm_.  Ranks must stay equal and match a single
    process that averages the two ranks' gradients; a parameter that never receives a gradient must keep .grad = None (the
    optimizer skips it, as in the reference with find_unused_parameters=True)."""
    res = _run2(_torch_optimizer_worker, staged)
    (_, sd0, none0), (_, sd1, none1) = res
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), f"ranks diverged on {k}"
    assert none0 == none1 and all(set(n) == {"unused.weight", "unused.bias"} for n in none0), none0
    torch.manual_seed(400)
    ref = MultiStageNet() if staged else Net()
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, eps=1e-4)
    topt.zero_grad()
    g = torch.Generator().manual_seed(13)
    data = torch.randn(9, 2, 5, 6, generator=g)
    K = 2
    # what the loop computes: micro-step 0 is exchanged (averaged) and stays; every later cycle adds its unsynced micro-step locally
    # and is averaged at the synced one -- in total the average over ranks of everything accumulated since the last zero_grad
    for i in range(9):
        for r in range(2):
            out, _ = ref(data[i, r]); (out.pow(2).mean() / 2).backward()
        if i % K == 0 and i != 0:
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
            topt.step()
            topt.zero_grad()
    for k, v in ref.state_dict().items():
        assert torch.allclose(torch.from_numpy(sd0[k]), v, atol=2e-6), (k, (torch.from_numpy(sd0[k]) - v).abs().max())
