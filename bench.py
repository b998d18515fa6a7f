#!/usr/bin/env python3
"""Denoiser-step throughput of the MI355X-native Oniris implementation (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): the Lunar-Lander net of gym_train.py:37-47 (46.2 M parameters), 64-frame sequences of
8x64x64 latents, B sequences per GPU, synthetic N(0,1) latents and random-init weights.  One step = the reference
training micro-step: EDM2Loss forward (Precond -> UNet over clean|noised frame slots), backward, gradient
all-reduce (N > 1), fused AdamW, with the reference's 3:1 mix of 3-D and 2-D steps (gym_train.py:96).
value = latent frames / s over the whole job = N * B * T * K / wall time of the K timed steps (max over ranks).
"""
import argparse
import contextlib
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (RCCL otherwise fails with `hipIpcGetMemHandle: invalid argument`);
# the environment normally exports it already -- set before anything touches the GPU, never overridden
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GYM_CFG = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=32, channel_mult=[1, 2, 4, 8],
               channel_mult_noise=None, channel_mult_emb=None, num_blocks=2, video_attn_resolutions=[8],
               frame_attn_resolutions=[16])
CS_CFG = dict(img_resolution=32, img_channels=8, label_dim=4, model_channels=128, channel_mult=[1, 2, 4, 4],
              channel_mult_noise=None, channel_mult_emb=None, num_blocks=2, video_attn_resolutions=[4],
              frame_attn_resolutions=[8])                       # cs_train.py:35-45 (BASELINE configs 3/4), 310.0 M


def _pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/rNN_pmc_traffic.json of the latest round:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench, summarised by
    scratch/pmc_traffic.py with the gfx950 correction 2*FETCH_SIZE + WRITE_SIZE).  PMC counters cannot be collected
    from inside the timed process, so the number is only as fresh as that file; None when the kernel is not in it."""
    import re
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((p for p in (os.path.join(prof, f"r{r:02d}_pmc_traffic.json") for r in range(9, 0, -1)) if os.path.exists(p)), None)
    if path is None:                                             # (the newest round's passes)
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    name = key.split("<")[0]
    want = [int(v) for v in re.findall(r"=(\d+)", key)]          # template arguments in KernelProfile's key order
    for k, v in table.items():
        if name in k:
            have = [int(x) if x.isdigit() else (1 if x == "true" else 0) for x in re.findall(r"[<,]\s*(\d+|true|false)", k)]
            if have[:len(want)] == want:
                return v["hbm_bytes_per_launch"]
    return None


MFMA_BF16_PEAK = 2.5e15          # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic forward FLOPs per sample of the 3-D step at T=64: 1.82 TFLOP; x3 fwd+bwd


C1_CFG = dict(img_resolution=64, img_channels=8, label_dim=4, model_channels=16, channel_mult=[1, 2, 4, 8], num_blocks=1,
              video_attn_resolutions=[8], frame_attn_resolutions=[16])      # BASELINE configs[0] / SURVEY 8d C1, 8.36 M


def _cpu_step_time(cfg, B, frames, warm):
    """Seconds of ONE 3-D training step (EDM2Loss forward + backward) of the CPU oracle, after `warm` untimed steps."""
    import torch
    import paramgen
    from oracle import oniris_oracle as O
    cfg = {k: v for k, v in cfg.items() if v is not None}
    torch.manual_seed(0)
    p = paramgen.precond_params(cfg, 0)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and "rope" not in k and "fourier" not in k) for k, v in p.items()}
    res = cfg["img_resolution"]
    images = torch.randn(B, frames, 8, res, res)
    labels = torch.randint(0, 4, (B, frames))
    sigma = (torch.randn(B, 2 * frames) * 1.0 + 1.2).exp()
    eps = torch.randn(B, 2 * frames, 8, res, res)
    dt = None
    for it in range(warm + 1):
        for v in p.values():
            v.grad = None
        t0 = time.time()
        loss, _, _ = O.edm2_loss(p, cfg, images, sigma, eps, labels, sigma_data=1.0)
        loss.backward()
        dt = time.time() - t0
    return dt


def cpu_baseline(frames):
    """The CPU oracle (fp32 PyTorch restatement, pinned to the reference by tests/golden) on the host cores, on the
    configurations BASELINE.md section 3 states, each timed on its second step (one warm-up):
      value: the bench workload's own net (gym UNet 46.2 M), B = 1, T = `frames` -- 16 by default (a bounded sample, ~30 s of
             CPU work), 64 = BASELINE configs[1] at B = 1 (`--cpu-frames 64`, about two minutes);
      c1:    BASELINE configs[0], the reference's CPU-runnable plumbing case (tiny UNet 8.36 M, B = 2, T = 8), always."""
    import torch
    cores = torch.get_num_threads()
    dt1 = _cpu_step_time(C1_CFG, 2, 8, 1)
    dt = _cpu_step_time(GYM_CFG, 1, frames, 1)
    return dict(value=frames / dt, unit="latent-frames/s", cores=cores, kind="port",
                sample=f"oracle (fp32 PyTorch CPU restatement), gym UNet 46.2M, B=1, T={frames}, one 3-D forward+backward "
                       f"step after one warm-up step, {dt:.1f} s",
                c1=dict(value=16 / dt1, unit="latent-frames/s", cores=cores,
                        sample=f"configs[0]: tiny UNet 8.36M, B=2, T=8, one 3-D forward+backward step after one warm-up step, "
                               f"{dt1:.2f} s"))


def rollout(args):
    """BASELINE config 5: edm2/sampler.py autoregressive rollout with KV / activation caches (plotting.py:163-166
    settings: num_steps=16, rho=2, sigma in [0.01, 80], S_churn=0, guidance=1 -> 31 UNet evaluations per frame)."""
    import torch
    from edm2.networks_edm2 import UNet, Precond
    from edm2.sampler import edm_sampler_with_mse
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    unet = UNet(**GYM_CFG).to(dev)
    torch.nn.init.constant_(unet.out_gain, 1.0)
    net = Precond(unet, sigma_data=1.0).to(dev).eval()
    B, ctx_frames = args.batch, args.ctx_frames
    with torch.no_grad():
        ctx = torch.randn(B, ctx_frames, 8, 64, 64, device=dev)
        lab = torch.randint(0, 4, (B, ctx_frames), device=dev)
        _, cache = net(ctx, torch.ones(B, ctx_frames, device=dev) * 0.05, lab, update_cache=True)
        for i in range(2):                      # warm-up frames
            _, _, _, cache = edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01,
                                                  sigma_max=80, rho=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.gen_frames):
            x, _, _, cache = edm_sampler_with_mse(net, cache, conditioning=lab[:, :1], num_steps=16, sigma_min=0.01,
                                                  sigma_max=80, rho=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    evals = 31 * args.gen_frames
    print(json.dumps({"metric": "rollout generated frames/s (config 5, KV-cached sampler)", "value": B * args.gen_frames / dt,
                      "unit": "frames/s", "n_gpus": 1, "ms_per_unet_eval": dt / evals * 1e3, "frames_generated": args.gen_frames,
                      "context_frames": ctx_frames + 2, "batch": B, "dtype": "bf16", "data": "synthetic",
                      "finite": bool(torch.isfinite(x).all())}))


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks the way the reference's cs_train.py
    is started (cs_train.py:164-174: LOCAL_RANK / init_process_group("nccl", "env://") under torchrun) -- as a CHILD
    process, `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` on 127.0.0.1 and a free
    port.  The children inherit stdout / stderr, so rank 0's JSON line is this process's output; the exit code is
    theirs.  This parent must not initialise the GPU (and does not even import torch)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """--dry-run: only the multi-rank plumbing of this script (rendezvous, barrier, max-over-ranks timing, rank 0's JSON
    line) on the gloo backend with no GPU -- what the CPU test of the launcher path runs (tests/test_bench_launch.py)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", init_method="env://")
    dist.barrier()
    if os.environ.get("ONIRIS_DRY_RUN_FAIL_RANK") == str(rank):     # (test hook: a rank that dies after the rendezvous)
        os._exit(3)
    t0 = time.perf_counter()
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ranks = [None] * world
    dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "0"))))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher plumbing only, NOT a measurement)", "value": 0.0, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "rccl_world": dist.get_world_size(), "backend": "gloo",
                          "ranks": ranks, "max_over_ranks": float(tt.item()), "dry_run": True,
                          "ms_per_step": (time.perf_counter() - t0) * 1e3}), flush=True)
    dist.destroy_process_group()


def _claim_stdout():
    """Exactly ONE line on stdout: the JSON record.  Native libraries print there too (RCCL writes a six-line version banner to
    fd 1 when its communicator comes up, ROCm tools the odd notice): from here on fd 1 of this process is stderr, and `print`
    to sys.stdout goes to the ORIGINAL stdout through a private descriptor."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(keep, "w", buffering=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=2, help="sequences per GPU (weak scaling)")
    ap.add_argument("--frames", type=int, default=None, help="frames per sequence (default 64 gym / 32 cs)")
    ap.add_argument("--net", choices=["gym", "cs"], default="gym",
                    help="gym = BASELINE configs[1] (the headline metric); cs = the Counter-Strike net of configs[2]/[3] "
                         "(32x32 latents, 310 M parameters, no conditioning) as an extra measurement")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay captured hipGraphs instead of launching eagerly (the step is GPU-bound either way; on ROCm "
                         "7.0 replays of graphs that contain the 8-wave / 98 KB-LDS conv kernel were observed to be "
                         "nondeterministic, so eager launch is the default -- DESIGN.md section 8)")
    ap.add_argument("--mode", choices=["train", "rollout"], default="train",
                    help="train = the BASELINE headline metric (default); rollout = config 5 (KV-cached sampler), extra line")
    ap.add_argument("--gen-frames", type=int, default=8)
    ap.add_argument("--ctx-frames", type=int, default=8, help="rollout: frames of the prefill (+ 2 warm-up frames) before the timed ones")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--accum", type=int, default=1,
                    help="gradient accumulation as in the reference loops (gym_train.py:96-112, cs_train.py:105-127): the optimizer "
                         "(+ clip, EMA, learning-rate schedule) runs every K-th micro-step, the K-1 others run their backward "
                         "under no_sync() (no gradient exchange).  An extra measurement, NOT the headline (K = 1)")
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 64 if args.net == "gym" else 32
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.mode == "train":
        sys.exit(self_launch(args))              # (this process never imports torch, never touches a GPU)
    _claim_stdout()
    if args.mode == "rollout":
        return rollout(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = bool(os.environ.get("ONIRIS_FORCE_DIST"))          # debug: run the RCCL/DDP path with a single rank
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus disagree")
    if args.dry_run:
        return dry_run(args, rank, world)
    # debug aid (NOT a measurement): ONIRIS_SHARE_GPU=1 runs every rank on cuda:0 with the gloo backend, so that the
    # multi-rank control flow of this script (collective matching, staged exchange) can be exercised on a 1-GPU box
    share = bool(os.environ.get("ONIRIS_SHARE_GPU"))
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group("gloo", init_method="env://")
        else:
            dist.init_process_group("nccl", init_method="env://", device_id=dev)
    rccl_world, devices = 1, [torch.cuda.current_device()]
    if world > 1 or force_dist:                    # what the collective library itself saw (goes into the JSON line)
        rccl_world = dist.get_world_size()
        devices = [None] * rccl_world
        dist.all_gather_object(devices, torch.cuda.current_device())

    from edm2.networks_edm2 import UNet, Precond
    from edm2.loss import EDM2Loss
    from autoregressive_diffusion_amd.parallel import FlatParams, OnirisDDP, FlatAdamW, FlatEMA
    from autoregressive_diffusion_amd import ops

    torch.manual_seed(0)
    cs = args.net == "cs"
    unet = UNet(**(CS_CFG if cs else GYM_CFG)).to(dev)
    for m in unet.modules():                      # give the zero-initialised gains a value so every branch carries signal
        if hasattr(m, "emb_gain"):
            torch.nn.init.constant_(m.emb_gain, 0.3)
    torch.nn.init.constant_(unet.out_gain, 1.0)
    flat = FlatParams(unet, lazy_small=True)
    # exchange form / transport: ONIRIS_DDP_EXCHANGE=allreduce|mesh, ONIRIS_DDP_BF16=1 (parallel.OnirisDDP; default: fp32 all-reduce per stage)
    model = OnirisDDP(unet, flat=flat, force_collectives=force_dist) if (world > 1 or force_dist) else unet
    net = Precond(model, use_fp16=True, sigma_data=1.0).to(dev).train()
    opt = FlatAdamW(flat, lr=1e-2, eps=1e-8)
    # optimizer side as in the reference loops: clip_grad_norm_(0.1) (gym_train.py:105 only) + AdamW +
    # PowerFunctionEMA(stds 0.05 / 0.10).update (gym_train.py:108, cs_train.py:121) -- one fused pass
    ema = FlatEMA(flat, stds=(0.050, 0.100))
    max_norm = None if cs else 0.1
    nimg = [0]
    loss_fn = (EDM2Loss(P_mean=0.9, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.1) if cs else   # cs_train.py:75
               EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5))         # gym_train.py:66-67

    B, T = args.batch, args.frames
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    res = unet.img_resolution
    latents = torch.randn(B, T, 8, res, res, device=dev, generator=g)
    actions = None if cs else torch.randint(0, 4, (B, T), device=dev, generator=g)      # cs_train.py:103 conditioning=None

    _host_t = [0.0, 0.0, 0.0] if os.environ.get("ONIRIS_HOST_TIMING") else None
    def fwd_bwd(just_2d):
        opt.zero_grad()
        if _host_t is None:
            loss, _ = loss_fn(net, latents, actions, just_2d=just_2d, sync=False)
            loss.backward()
            return loss
        ta, ca_ = time.perf_counter(), time.thread_time()         # ONIRIS_HOST_TIMING: host enqueue time, forward / backward
        loss, _ = loss_fn(net, latents, actions, just_2d=just_2d, sync=False)
        tb = time.perf_counter()
        _host_t[2] += time.thread_time() - ca_                   # CPU time of the forward (wall - CPU = blocked, not computing)
        loss.backward()
        _host_t[0] += tb - ta; _host_t[1] += time.perf_counter() - tb
        return loss

    use_graph = bool(args.graph)
    graphed = {}
    if use_graph:
        from autoregressive_diffusion_amd.graphs import GraphedStep
        if world > 1:      # the graph holds forward+backward only; the RCCL exchange is issued eagerly after the replay
            def make(j2d):
                def f():
                    with model.no_sync():
                        return fwd_bwd(j2d)
                return f
        else:
            def make(j2d):
                return lambda: fwd_bwd(j2d)
        graphed = {False: GraphedStep(make(False), params=flat.params, flat=flat),
                   True: GraphedStep(make(True), params=flat.params, flat=flat)}

    _only = os.environ.get("ONIRIS_ONLY_MODE")

    accum = max(1, args.accum)
    from edm2.loss import learning_rate_schedule
    ref_lr, sched_steps = 1e-2, 100000 / 50                      # gym_train.py:69,110-112 (total_number_of_steps / 50)
    micro = [0]                                                  # micro-steps taken (the reference's loop index i)

    def step(i, profile=False):
        just_2d = (i % 4 == 0)                                   # gym_train.py:96
        if _only:                                                # profiling aid: ONIRIS_ONLY_MODE=2d|3d (not the metric)
            just_2d = _only == "2d"
        if accum > 1:
            return accum_step(just_2d)
        if use_graph and not profile:
            loss = graphed[just_2d]()
            if world > 1:
                model.allreduce_grads()
        else:
            loss = fwd_bwd(just_2d)
        if world > 1 or force_dist:
            model.wait()
        nimg[0] += world * B
        opt.step(max_norm=max_norm, ema=ema.weights(nimg[0], world * B))      # t_next = images seen so far (gym_train.py:108)
        return loss

    def accum_step(just_2d):
        """One micro-step of the reference loops with accumulation_steps = K (cs_train.py:105-127): backward under
        no_sync() unless i % K == 0; on those i (except i = 0) optimizer.step, zero_grad, EMA update and the learning-rate
        schedule written into param_groups."""
        i = micro[0]
        micro[0] += 1
        sync_now = i % accum == 0
        with (contextlib.nullcontext() if sync_now else model.no_sync()):
            loss, _ = loss_fn(net, latents, actions, just_2d=just_2d, sync=False)
            loss.backward()
        nimg[0] += world * B
        if sync_now:
            if world > 1 or force_dist:
                model.wait()
            if i != 0:
                opt.step(max_norm=max_norm, ema=ema.weights(nimg[0], accum * world * B))
                opt.zero_grad()
                for g_ in opt.param_groups:
                    g_["lr"] = learning_rate_schedule(i, ref_lr, sched_steps, sched_steps)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dbg = os.environ.get("ONIRIS_DEBUG_LOSS")
    for i in range(max(args.warmup, 16 if use_graph else 0)):     # graphs: 3 eager + 1 capture call per step flavour
        l_ = step(i)
        if dbg:
            print("warmup", i, float(l_.item()), file=sys.stderr)
    fence()
    if _host_t:
        _host_t[0] = _host_t[1] = _host_t[2] = 0.0
    t0 = time.perf_counter()
    hist = []
    for i in range(args.steps):
        last = step(i)
        if dbg == "2":
            print("timed", i, float(last.item()), file=sys.stderr)
        if dbg == "3":
            hist.append(last.detach().clone())
    t_enq = time.perf_counter() - t0                               # host time to ENQUEUE the K steps (before the fence)
    fence()
    if dbg == "3":
        print("timed losses", [round(float(h.item()), 4) for h in hist], file=sys.stderr)
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(last.item())
    if os.environ.get("ONIRIS_HOST_TIMING") == "2":
        from autoregressive_diffusion_amd import _lib as _l
        tot = sum(v[1] for v in _l.call_stats.values())
        print(f"C-ABI calls: {sum(v[0] for v in _l.call_stats.values())} calls, {tot * 1e3:.1f} ms in total (whole process)", file=sys.stderr)
        for k, v in sorted(_l.call_stats.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   {k:28s} n={v[0]:6d}  {v[1] / v[0] * 1e6:7.1f} us/call", file=sys.stderr)
    if os.environ.get("ONIRIS_HOST_TIMING"):
        print(f"host enqueue {t_enq / args.steps * 1e3:.2f} ms/step of {dt / args.steps * 1e3:.2f} ms/step "
              f"(forward {_host_t[0] / args.steps * 1e3:.2f} [cpu {_host_t[2] / args.steps * 1e3:.2f}], backward {_host_t[1] / args.steps * 1e3:.2f})", file=sys.stderr)

    # per-mode step times (one 3-D and one 2-D step, timed separately, not part of `value`)
    per_mode = {}
    for name, i in (("ms_3d_step", 1), ("ms_2d_step", 0)):
        fence(); t1 = time.perf_counter(); step(i); fence()
        per_mode[name] = (time.perf_counter() - t1) * 1e3

    roof, kernels, roof_attn, roof_attn_bwd = None, None, None, None
    if rank == 0 and not args.no_profile:
        ops.KernelProfile.start()
        for i in range(4):                                        # one full 3:1 cycle (eager), every MFMA conv launch
            step(i, profile=True)                                 # bracketed by HIP events on its own stream
        agg = ops.KernelProfile.stop()
        kernels = {k: dict(launches=v["launches"], ms_total=round(v["ms"], 3),
                           tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)) for k, v in
                   sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        attn = {k: v for k, v in agg.items() if k.startswith("attn_fwd") and "MODE=2" in k}      # VideoAttention forward
        dom, v = max(((k, v) for k, v in agg.items() if not k.startswith("attn_")), key=lambda kv: kv[1]["ms"])
        achieved = v["flops"] / (v["ms"] * 1e-3)
        roof = dict(bound="mfma", kernel=dom, launches=v["launches"], avg_launch_ms=v["ms"] / v["launches"],
                    flops_per_launch=v["flops"] / v["launches"], achieved=achieved / 1e12, peak=MFMA_BF16_PEAK / 1e12,
                    unit="TFLOP/s", frac=achieved / MFMA_BF16_PEAK, traffic=_pmc_traffic(dom),
                    # operands read once + results written once, mean over the same launches (to set `traffic` against)
                    algorithmic_bytes=v["bytes"] / v["launches"])
        roof_attn = None
        if attn:                                                  # the north star's second roofline: VideoAttention forward
            k, v = max(attn.items(), key=lambda kv: kv[1]["ms"])
            ach = v["flops"] / (v["ms"] * 1e-3)
            roof_attn = dict(bound="mfma", kernel=k, launches=v["launches"], avg_launch_ms=v["ms"] / v["launches"],
                             flops_per_launch=v["flops"] / v["launches"], achieved=ach / 1e12, peak=MFMA_BF16_PEAK / 1e12,
                             unit="TFLOP/s", frac=ach / MFMA_BF16_PEAK,
                             note="algorithmic FLOPs = unmasked token pairs x 4 x 64 x heads x B (SURVEY 8d)")
        # ... and its backward: dQ + dK/dV launches of a layer together, priced on the ALGORITHMIC backward FLOPs = 2.5 x the
        # forward's (five products S, dP, dV, dK, dQ; the two kernels execute seven: each recomputes S and dP)
        bq = {k: v for k, v in agg.items() if k.startswith("attn_bwd_dq") and "MODE=2" in k}
        bkv = {k: v for k, v in agg.items() if k.startswith("attn_bwd_dkv") and "MODE=2" in k}
        roof_attn_bwd = None
        if attn and bq and bkv:
            fwd = max(attn.values(), key=lambda v: v["ms"])
            n = fwd["launches"]
            ms = sum(v["ms"] for v in bq.values()) + sum(v["ms"] for v in bkv.values())
            fl = 2.5 * fwd["flops"]
            ach = fl / (ms * 1e-3)
            roof_attn_bwd = dict(bound="mfma", kernels=sorted(bq) + sorted(bkv), layers=n, avg_layer_ms=ms / n,
                                 flops_per_layer=fl / n, achieved=ach / 1e12, peak=MFMA_BF16_PEAK / 1e12, unit="TFLOP/s",
                                 frac=ach / MFMA_BF16_PEAK,
                                 note="dQ + dK/dV launches of one VideoAttention layer; algorithmic FLOPs = 2.5 x forward")
    elif world > 1:
        for i in range(4):
            step(i, profile=True)                                 # keep collectives matched across ranks
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0 and not cs:
        cpu = cpu_baseline(args.cpu_frames)
    if world > 1:
        dist.barrier()

    if rank == 0:
        frames = world * B * T * args.steps
        out = {"metric": "denoiser-step frames/sec at 1/2/4/8 MI355X; 64-frame Lunar-Lander seq",
               "value": frames / dt, "unit": "latent-frames/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "rccl_world": rccl_world, "devices": sorted(set(devices)),
               "backend": (dist.get_backend() if (world > 1 or force_dist) else None),
               "ddp": ({"exchange": model.exchange, "grad_dtype": str(model.grad_dtype or "fp32"), "stages": len(flat.stages)}
                       if (world > 1 or force_dist) else None),
               "config": {"workload": (f"Counter-Strike latents {T}-frame seq, EDM2 UNet 310.0M (cs_train.py:35-45), " if cs else
                                       f"Lunar-Lander {T}-frame seq, gym EDM2 UNet 46.2M (gym_train.py:37-47), ") +
                                      f"{B} seq/GPU, step = EDM2Loss fwd + bwd + grad all-reduce + [grad-norm clip +] AdamW + 2 EMA profiles, "
                                      f"3:1 mix of 3-D/2-D steps", "global_batch": world * B, "seq_len": T,
                          "parallelism": f"dp{world}", "hip_graph": bool(use_graph),
                          **({"accum_NOT_THE_HEADLINE": accum, "lr": opt.param_groups[0]["lr"]} if accum > 1 else {}),
                          **({"shared_gpu_gloo_NOT_A_MEASUREMENT": True} if share else {}),
                          **({"only_mode_NOT_THE_METRIC": _only} if _only else {}),
                          **{k: round(v, 2) for k, v in per_mode.items()}},
               "loss": loss_val, "roofline": roof, "cpu_baseline": cpu, "roofline_attention": roof_attn,
               "roofline_attention_bwd": roof_attn_bwd, "kernels": kernels}
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
