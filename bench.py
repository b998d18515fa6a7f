#!/usr/bin/env python3
"""Denoiser-step throughput of the MI355X-native Oniris implementation (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): the Lunar-Lander net of gym_train.py:37-47 (46.2 M parameters), 64-frame sequences of
8x64x64 latents, B sequences per GPU -- by default the reference's own micro-batch, 8 (gym_train.py:55; rounds 1-3 quoted
B = 2, which the default run still reports as extra.gym_t64_b2) -- synthetic N(0,1) latents and random-init weights.  One step = the reference
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


def _git_blob_sha1(path):
    """The git blob id of a file's bytes (what `git hash-object` prints), so that the JSON line names the exact committed
    PMC summary its `traffic` figure was read from."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def _pmc_traffic(key):
    """(HBM bytes per launch of kernel `key`, source) from the committed PMC passes (profiles/rNN_pmc_traffic.json of the
    latest round: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench, summarised by
    scratch/pmc_traffic.py with the gfx950 correction 2*FETCH_SIZE + WRITE_SIZE).  PMC counters cannot be collected from
    inside the timed process, so the number is only as fresh as that file: `source` = {file, git_blob} says which one it
    was.  (None, source) when the kernel is not in it."""
    import re
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((p for p in (os.path.join(prof, f"r{r:02d}_pmc_traffic.json") for r in range(9, 0, -1)) if os.path.exists(p)), None)
    if path is None:                                             # (the newest round's passes)
        return None, None
    try:
        table = json.load(open(path))
        source = {"file": os.path.relpath(path, ROOT), "git_blob": _git_blob_sha1(path),
                  "note": "committed rocprofv3 --pmc passes of this bench (not collected in this run)"}
    except Exception:
        return None, None
    name = key.split("<")[0]
    want = [int(v) for v in re.findall(r"=(\d+)", key)]          # template arguments in KernelProfile's key order
    for k, v in table.items():
        if name in k:
            have = [int(x) if x.isdigit() else (1 if x == "true" else 0) for x in re.findall(r"[<,]\s*(\d+|true|false)", k)]
            if have[:len(want)] == want:
                return v["hbm_bytes_per_launch"], source
    return None, source


MFMA_BF16_PEAK = 2.5e15          # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE = 6.3e12          # B/s, the figure SURVEY 8d / BASELINE.md section 3 price memory-bound layers with
# algorithmic forward FLOPs per sample of the 3-D step; x3 forward + backward
ALGO_FWD_FLOPS = {("gym", 64): 1.82e12, ("cs", 32): 2.58e12, ("cs", 64): 5.16e12}


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


def rollout(args, quiet=False):
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
    out = {"metric": "rollout generated frames/s (config 5, KV-cached sampler)", "value": B * args.gen_frames / dt,
           "unit": "frames/s", "n_gpus": 1, "ms_per_unet_eval": dt / evals * 1e3, "frames_generated": args.gen_frames,
           "context_frames": ctx_frames + 2, "batch": B, "dtype": "bf16", "data": "synthetic",
           "finite": bool(torch.isfinite(x).all()),
           "runtime_env": {k: os.environ.get(k) for k in ROLLOUT_ENV}}
    if not quiet:
        print(json.dumps(out))
    return out


# Runtime environment of the KV-cached rollout (BASELINE configs[4]).  A generated frame is 31 replays of a ~130-node hipGraph on ONE
# stream; with the runtime's default of four hardware queues every replayed node and every replay boundary pays for cross-queue ordering
# (profiles/r06_rollout_gaps.txt: 19 % of a frame idle; profiles/r06_rollout_stream_ab*.txt: 27.2 -> 32.3 frames/s with one queue,
# nothing else changed).  One hardware queue is right for a single-stream inference process and WRONG for data-parallel training
# (RCCL's stream must overlap the backward kernels), so it is set for rollout-only processes -- `--mode rollout`, and the child process
# in which the default training run measures its `extra.rollout_*` records -- never for a training process.  setdefault: an explicit
# setting in the caller's environment wins.  The training step does not notice it (profiles/r06_hwq_train_ab.txt: 8551 vs 8554 frames/s).
ROLLOUT_ENV = {"GPU_MAX_HW_QUEUES": "1"}


def rollout_child(batch, gen_frames, ctx_frames=8):
    """bench.py --mode rollout in a CHILD process with ROLLOUT_ENV (this process has initialised the HIP runtime long ago: the
    variables are read when it comes up).  Returns the child's JSON record."""
    import subprocess
    env = dict(os.environ)
    for k, v in ROLLOUT_ENV.items():
        env.setdefault(k, v)
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "rollout", "--batch", str(batch), "--gen-frames", str(gen_frames),
           "--ctx-frames", str(ctx_frames)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError(f"rollout child exited {r.returncode}: {r.stderr[-400:]}")
    return json.loads(r.stdout.strip().splitlines()[-1])


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


class Watchdog:
    """Multi-rank runs only: a daemon thread that ends THIS rank with a non-zero exit code and the name of the stage it
    was in when no progress has been reported for `limit` seconds -- a hung collective (a rank that died, a mismatched
    exchange) otherwise blocks the whole job until the driver's timeout with nothing to read.  os._exit from the thread:
    no re-exec, no signal to other processes; torch.distributed.run then takes the other ranks down."""

    def __init__(self, rank, limit):
        import threading
        self.rank, self.limit, self.stage, self.t = rank, limit, "start", time.monotonic()
        self._stop = False
        self.thread = threading.Thread(target=self._run, daemon=True)
        if limit > 0:
            self.thread.start()

    def beat(self, stage):
        self.stage, self.t = stage, time.monotonic()

    def stop(self):
        self._stop = True

    def _run(self):
        while not self._stop:
            time.sleep(1.0)
            idle = time.monotonic() - self.t
            if idle > self.limit and not self._stop:
                print(f"bench.py watchdog: rank {self.rank} made no progress for {idle:.0f} s in stage '{self.stage}' "
                      f"(hung collective / dead peer?) -- exiting 124", file=sys.stderr, flush=True)
                os._exit(124)


def dry_run(args, rank, world):
    """--dry-run: only the multi-rank plumbing of this script (rendezvous, barrier, max-over-ranks timing, rank 0's JSON
    line, the watchdog) on the gloo backend with no GPU -- what the CPU test of the launcher path runs
    (tests/test_bench_launch.py)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    wd = Watchdog(rank, args.watchdog)
    wd.beat("init_process_group(gloo)")
    dist.init_process_group("gloo", init_method="env://")
    wd.beat("first barrier")
    dist.barrier()
    if os.environ.get("ONIRIS_DRY_RUN_FAIL_RANK") == str(rank):     # (test hook: a rank that dies after the rendezvous)
        os._exit(3)
    if os.environ.get("ONIRIS_DRY_RUN_HANG_RANK") == str(rank):     # (test hook: a rank that stops answering)
        wd.stop()
        time.sleep(3600)
    t0 = time.perf_counter()
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    wd.beat("all_reduce(MAX) of the step time")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ranks = [None] * world
    dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "0"))))
    wd.beat("last barrier")
    dist.barrier()
    wd.stop()
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


def train(args, netname, steps, warmup, rank, world, dev, wd, light=False, light_batch=2, light_frames=None):
    """The timed training job on this rank: build the net, W warm-up + K timed steps between fences, max over ranks.
    light: an extra measurement inside the headline run (no per-kernel profile, no CPU baseline): {frames_s, ms_per_step, ...}.
    Returns the JSON record (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from edm2.networks_edm2 import UNet, Precond
    from edm2.loss import EDM2Loss
    from autoregressive_diffusion_amd.parallel import FlatParams, OnirisDDP, FlatAdamW, FlatEMA
    from autoregressive_diffusion_amd import ops

    force_dist = bool(os.environ.get("ONIRIS_FORCE_DIST"))          # debug: run the RCCL/DDP path with a single rank
    share = bool(os.environ.get("ONIRIS_SHARE_GPU"))
    multi = world > 1 or force_dist
    rccl_world, devices = 1, [torch.cuda.current_device()]
    if multi:                                      # what the collective library itself saw (goes into the JSON line)
        wd.beat("all_gather_object(devices)")
        rccl_world = dist.get_world_size()
        devices = [None] * rccl_world
        dist.all_gather_object(devices, torch.cuda.current_device())

    torch.manual_seed(0)
    cs = netname == "cs"
    unet = UNet(**(CS_CFG if cs else GYM_CFG)).to(dev)
    for m in unet.modules():                      # give the zero-initialised gains a value so every branch carries signal
        if hasattr(m, "emb_gain"):
            torch.nn.init.constant_(m.emb_gain, 0.3)
    torch.nn.init.constant_(unet.out_gain, 1.0)
    torch_loop = getattr(args, "wrapper", "oniris") == "torch" and not light
    if torch_loop:
        # --wrapper torch: the reference's loop AS WRITTEN (cs_train.py:53-54,76-78,105-121; gym_train.py:105-108): torch's own
        # DistributedDataParallel around the UNet (multi-rank / ONIRIS_FORCE_DIST runs), torch.optim.AdamW over precond.parameters(),
        # clip_grad_norm_, two deep-copied EMA trackers updated parameter by parameter (phema.py:104-108), the loss with its host
        # round trip.  An extra measurement of the drop-in path, NOT the headline (which uses OnirisDDP + the fused optimizer).
        import copy
        from torch.nn.parallel import DistributedDataParallel as TorchDDP
        from autoregressive_diffusion_amd.parallel import power_function_beta
        flat = None
        wd.beat("torch DistributedDataParallel construction")
        model = (TorchDDP(unet, device_ids=[dev.index], output_device=dev.index, find_unused_parameters=True) if multi else unet)
        if multi and force_dist and world == 1:
            unet.__dict__["_oniris_inner_ddp"].force_collectives = True
        net = Precond(model, use_fp16=True, sigma_data=1.0).to(dev).train()
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2, eps=1e-8)
        opt.zero_grad()
        ema_stds = (0.050, 0.100)
        ema_copies = [copy.deepcopy(net) for _ in ema_stds]
        ema = None
    else:
        flat = FlatParams(unet, lazy_small=True)
        # exchange form / transport: ONIRIS_DDP_EXCHANGE=allreduce|mesh, ONIRIS_DDP_BF16=1 (parallel.OnirisDDP; default: fp32 all-reduce per stage)
        wd.beat("OnirisDDP construction (parameter / buffer broadcast)")
        model = OnirisDDP(unet, flat=flat, force_collectives=force_dist, auto_wait=False) if multi else unet      # (wait() is placed and timed below)
        net = Precond(model, use_fp16=True, sigma_data=1.0).to(dev).train()
        opt = FlatAdamW(flat, lr=1e-2, eps=1e-8)
        # optimizer side as in the reference loops: clip_grad_norm_(0.1) (gym_train.py:105 only) + AdamW +
        # PowerFunctionEMA(stds 0.05 / 0.10).update (gym_train.py:108, cs_train.py:121) -- one fused pass
        ema = FlatEMA(flat, stds=(0.050, 0.100))
    max_norm = None if cs else 0.1
    nimg = [0]
    loss_fn = (EDM2Loss(P_mean=0.9, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.1) if cs else   # cs_train.py:75
               EDM2Loss(P_mean=1.2, P_std=1.0, sigma_data=1.0, context_noise_reduction=0.5))         # gym_train.py:66-67

    B = args.batch if not light else light_batch
    T = (args.frames if not light else light_frames) or (32 if cs else 64)
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

    _only = os.environ.get("ONIRIS_ONLY_MODE")

    accum = max(1, args.accum) if not light else 1
    from edm2.loss import learning_rate_schedule
    ref_lr, sched_steps = 1e-2, 100000 / 50                      # gym_train.py:69,110-112 (total_number_of_steps / 50)
    micro = [0]                                                  # micro-steps taken (the reference's loop index i)
    comm_events = []                                             # (before, after) model.wait() on the compute stream
    comm_stage = {}                                              # exchange label -> [(before, after)] around ITS wait

    def wait_exchange():
        """model.wait() makes the compute stream wait for RCCL's: the events around it measure how long the compute stream
        stood still for communication that the backward pass did not hide (+ the bf16 / mesh finishing passes)."""
        if comm_events is not None and len(comm_events) < 4096:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); model.wait(timing=comm_stage); e1.record()
            comm_events.append((e0, e1))
        else:
            model.wait()

    def torch_step(just_2d):
        """One micro-step of the reference loops with accumulation_steps = 1, line by line."""
        loss, _unweighted = loss_fn(net, latents, actions, just_2d=just_2d)      # (sync: the reference's .cpu().item(), loss.py:41)
        loss.backward()
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm)           # gym_train.py:105
        opt.step()
        opt.zero_grad()
        nimg[0] += world * B
        with torch.no_grad():                                                    # phema.py:104-108
            for std, cp in zip(ema_stds, ema_copies):
                beta = power_function_beta(std, nimg[0], world * B)
                for p_net, p_ema in zip(net.parameters(), cp.parameters()):
                    p_ema.lerp_(p_net, 1 - beta)
        return loss

    def step(i):
        just_2d = (i % 4 == 0)                                   # gym_train.py:96
        if _only:                                                # profiling aid: ONIRIS_ONLY_MODE=2d|3d (not the metric)
            just_2d = _only == "2d"
        if torch_loop:
            return torch_step(just_2d)
        if accum > 1:
            return accum_step(just_2d)
        loss = fwd_bwd(just_2d)
        if multi:
            wait_exchange()
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
            if multi:
                wait_exchange()
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

    tag = f"[{netname}] "
    dbg = os.environ.get("ONIRIS_DEBUG_LOSS")
    for i in range(warmup):
        wd.beat(f"{tag}warm-up step {i}")
        l_ = step(i)
        if dbg:
            print("warmup", i, float(l_.item()), file=sys.stderr)
    wd.beat(f"{tag}fence before the timed steps")
    fence()
    if _host_t:
        _host_t[0] = _host_t[1] = _host_t[2] = 0.0
    del comm_events[:]
    comm_stage.clear()
    t0 = time.perf_counter()
    hist = []
    for i in range(steps):
        wd.beat(f"{tag}timed step {i}")
        last = step(i)
        if dbg == "2":
            print("timed", i, float(last.item()), file=sys.stderr)
        if dbg == "3":
            hist.append(last.detach().clone())
    t_enq = time.perf_counter() - t0                               # host time to ENQUEUE the K steps (before the fence)
    wd.beat(f"{tag}fence after the timed steps")
    fence()
    if dbg == "3":
        print("timed losses", [round(float(h.item()), 4) for h in hist], file=sys.stderr)
    dt = time.perf_counter() - t0
    if world > 1:
        wd.beat(f"{tag}all_reduce(MAX) of the step time")
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(last.item())
    exposed = (sum(a.elapsed_time(b) for a, b in comm_events) / max(1, steps)) if (multi and comm_events) else None
    exposed_stage = ({k: round(sum(a.elapsed_time(b) for a, b in v) / max(1, steps), 4) for k, v in sorted(comm_stage.items())}
                     if (multi and comm_stage) else None)
    comm_events = None                                            # (no events in the extra steps below)
    if light:
        return dict(frames_s=world * B * T * steps / dt, ms_per_step=dt / steps * 1e3, steps=steps, warmup=warmup,
                    seq_len=T, seq_per_gpu=B, loss=loss_val,
                    workload=("Counter-Strike latents, EDM2 UNet 310.0M (cs_train.py:35-45)" if cs else "gym UNet 46.2M")) if rank == 0 else None
    if os.environ.get("ONIRIS_HOST_TIMING") == "2":
        from autoregressive_diffusion_amd import _lib as _l
        tot = sum(v[1] for v in _l.call_stats.values())
        print(f"C-ABI calls: {sum(v[0] for v in _l.call_stats.values())} calls, {tot * 1e3:.1f} ms in total (whole process)", file=sys.stderr)
        for k, v in sorted(_l.call_stats.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   {k:28s} n={v[0]:6d}  {v[1] / v[0] * 1e6:7.1f} us/call", file=sys.stderr)
    if os.environ.get("ONIRIS_HOST_TIMING"):
        print(f"host enqueue {t_enq / steps * 1e3:.2f} ms/step of {dt / steps * 1e3:.2f} ms/step "
              f"(forward {_host_t[0] / steps * 1e3:.2f} [cpu {_host_t[2] / steps * 1e3:.2f}], backward {_host_t[1] / steps * 1e3:.2f})", file=sys.stderr)

    # per-mode step times (one 3-D and one 2-D step, timed separately, not part of `value`; median of 3)
    per_mode = {}
    for name, i in (("ms_3d_step", 1), ("ms_2d_step", 0)):
        ts = []
        for _ in range(3):
            wd.beat(f"{tag}per-mode step ({name})")
            fence(); t1 = time.perf_counter(); step(i); fence()
            ts.append((time.perf_counter() - t1) * 1e3)
        per_mode[name] = sorted(ts)[1]

    roof, kernels, roof_attn, roof_attn_bwd, roof_step, roof_step_2d = None, None, None, None, None, None
    if rank == 0 and not args.no_profile:
        # one full 3:1 cycle (eager), every MFMA launch bracketed by HIP events on its own stream; the 2-D step and the three
        # 3-D steps are aggregated separately (roofline_step is the 3-D step's)
        agg, agg3, agg2 = {}, {}, {}
        for i in range(4):
            wd.beat(f"{tag}profiled step {i}")
            ops.KernelProfile.start()
            step(i)
            part = ops.KernelProfile.stop()
            for k, v in part.items():
                for tgt in ((agg, agg3) if i % 4 else (agg, agg2)):
                    a = tgt.setdefault(k, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0, t_min=0.0))
                    for f_ in ("launches", "flops", "ms", "bytes", "t_min"):
                        a[f_] += v[f_]
        # per kernel: frac = algorithmic FLOPs / time / bf16 MFMA peak; roof = sum of the launches' roofline times max(FLOPs / 2.5 PF,
        # algorithmic bytes / 6.3 TB/s) / measured time, bound = which term makes up most of that sum (the 32-channel level and the
        # 1x1 convs are HBM-bound: their `frac` of the MFMA peak says little, `roof` is the figure to read)
        kernels = {k: dict(launches=v["launches"], ms_total=round(v["ms"], 3),
                           tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1), frac=round(v["flops"] / (v["ms"] * 1e-3) / MFMA_BF16_PEAK, 3),
                           roof=round(v["t_min"] / (v["ms"] * 1e-3), 3),
                           bound=("hbm" if v["bytes"] / HBM_ACHIEVABLE > v["flops"] / MFMA_BF16_PEAK else "mfma"),
                           gbytes_s=(round(v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["bytes"] else None))
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:16]}
        if os.environ.get("ONIRIS_PROFILE_SHAPES"):      # per launch shape (keys carry it: ops.PROFILE_SHAPES), the three 3-D steps
            print("3-D steps of the profiled cycle, per conv launch shape: launches, ms, roof (= sum of t_min / time), algorithmic TB/s, TFLOP/s", file=sys.stderr)
            for k, v in sorted(agg3.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"  {k:100s} {v['launches']:4d} {v['ms']:8.3f} ms  roof {v['t_min'] / (v['ms'] * 1e-3):5.3f}  "
                      f"{v['bytes'] / (v['ms'] * 1e-3) / 1e12:5.2f} TB/s {v['flops'] / (v['ms'] * 1e-3) / 1e12:7.1f} TF", file=sys.stderr)
        attn = {k: v for k, v in agg.items() if k.startswith("attn_fwd") and "MODE=2" in k}      # VideoAttention forward
        dom, v = max(((k, v) for k, v in agg.items() if not k.startswith("attn_")), key=lambda kv: kv[1]["ms"])
        achieved = v["flops"] / (v["ms"] * 1e-3)
        traffic, traffic_src = _pmc_traffic(dom)
        roof = dict(bound="mfma", kernel=dom, launches=v["launches"], avg_launch_ms=v["ms"] / v["launches"],
                    flops_per_launch=v["flops"] / v["launches"], achieved=achieved / 1e12, peak=MFMA_BF16_PEAK / 1e12,
                    unit="TFLOP/s", frac=achieved / MFMA_BF16_PEAK, traffic=traffic, traffic_source=traffic_src,
                    # operands read once + results written once, mean over the same launches (to set `traffic` against)
                    algorithmic_bytes=v["bytes"] / v["launches"])
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
        # the WHOLE 3-D step against the roofline (SURVEY 8d "achieved = sum t_min / t_measured"):
        # frac = algorithmic FLOPs of a forward + backward step / its wall time / MFMA peak; sum_t_min = the sum over the step's
        # MFMA launches (conv forward / dgrad / wgrad, attention) of max(FLOPs / MFMA peak, algorithmic bytes / 6.3 TB/s)
        if agg3:
            algo = ALGO_FWD_FLOPS.get((netname, T))
            prof_fl = sum((2.5 / 3.5 if (k.startswith("attn_bwd") and "MODE=2" in k) or k.startswith("frame_attn_qkv_bwd") else 1.0) * v["flops"]
                          for k, v in agg3.items()) / 3
            fl = 3.0 * algo * B if algo is not None else prof_fl
            ms3 = per_mode["ms_3d_step"]
            tmin = sum(v["t_min"] for v in agg3.values()) / 3 * 1e3
            mfma_ms = sum(v["ms"] for v in agg3.values()) / 3
            roof_step = dict(bound="mfma", flops_per_step=fl, flops_source=("BASELINE.md section 4 x 3 x B" if algo is not None else
                                                                            "sum of the launches' algorithmic FLOPs"),
                             flops_profiled=prof_fl, ms_3d_step=ms3, achieved=fl / (ms3 * 1e-3) / 1e12, peak=MFMA_BF16_PEAK / 1e12,
                             unit="TFLOP/s", frac=fl / (ms3 * 1e-3) / MFMA_BF16_PEAK, sum_t_min_ms=tmin, frac_t_min=tmin / ms3,
                             mfma_kernels_ms=mfma_ms,
                             note="3-D step (3 of 4 steps; the 2-D step does a third of the work): forward + backward + optimizer; "
                                  "sum_t_min over its MFMA launches = sum of max(FLOPs / 2.5 PF, algorithmic bytes / 6.3 TB/s); "
                                  "mfma_kernels_ms = their measured time, the rest of ms_3d_step is elementwise / weight / optimizer "
                                  "passes and launch boundaries")
        # the 2-D step (one step in four, gym_train.py:96: own-frame convolutions on the T real frames, 1x1 convs, per-frame
        # attention at both attention levels -- conv.py:60 `just_2d`): algorithmic FLOPs = the sum over its MFMA launches
        # (attention backward priced at 2.5 x its forward, as above) x 1, against its own wall time
        if agg2:
            fl2 = sum((2.5 / 3.5 if k.startswith("attn_bwd") or k.startswith("frame_attn_qkv_bwd") or k == "frame_attn_bwd_kernel" else 1.0) * v["flops"]
                      for k, v in agg2.items())
            ms2 = per_mode["ms_2d_step"]
            tmin2 = sum(v["t_min"] for v in agg2.values()) * 1e3
            roof_step_2d = dict(bound="mfma", flops_per_step=fl2, flops_source="sum of the launches' algorithmic FLOPs",
                                ms_2d_step=ms2, achieved=fl2 / (ms2 * 1e-3) / 1e12, peak=MFMA_BF16_PEAK / 1e12, unit="TFLOP/s",
                                frac=fl2 / (ms2 * 1e-3) / MFMA_BF16_PEAK, sum_t_min_ms=tmin2, frac_t_min=tmin2 / ms2,
                                mfma_kernels_ms=sum(v["ms"] for v in agg2.values()),
                                kernels={k: dict(launches=v["launches"], ms_total=round(v["ms"], 3),
                                                 frac=round(v["flops"] / (v["ms"] * 1e-3) / MFMA_BF16_PEAK, 3),
                                                 roof=round(v["t_min"] / (v["ms"] * 1e-3), 3))
                                         for k, v in sorted(agg2.items(), key=lambda kv: -kv[1]["ms"])[:8]})
    elif world > 1:
        for i in range(4):
            wd.beat(f"{tag}unprofiled cycle step {i} (collectives matched across ranks)")
            step(i)                                               # keep collectives matched across ranks
    if rank != 0:
        return None
    frames = world * B * T * steps
    return {"metric": "denoiser-step frames/sec at 1/2/4/8 MI355X; 64-frame Lunar-Lander seq",
            "value": frames / dt, "unit": "latent-frames/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "rccl_world": rccl_world, "devices": sorted(set(devices)), "device_of_rank": devices,
            "backend": (dist.get_backend() if multi else None),
            "ddp": ({"wrapper": "torch.nn.parallel.DistributedDataParallel + inner OnirisDDP for the kernel-owned weights",
                     "kernel_owned_parameters": len(unet.__dict__["_oniris_inner_ddp"].flat.params),
                     "stages": len(unet.__dict__["_oniris_inner_ddp"].flat.stages)} if (multi and torch_loop) else
                    {"exchange": model.exchange, "grad_dtype": str(model.grad_dtype or "fp32"), "stages": len(flat.stages),
                     "stage_mb": [round((hi - lo) * 4 / 2 ** 20, 1) for _, lo, hi in flat.stages],
                     "head_mb": round((flat.head[1] - flat.head[0]) * 4 / 2 ** 20, 1),
                     "exposed_comm_ms_per_step": exposed,
                     "exposed_comm_ms_per_step_by_stage": exposed_stage,
                     "comm_cus": getattr(model, "comm_cus", 0),
                     "exposed_comm_note": "compute-stream time spent inside OnirisDDP.wait() per timed step on rank 0 (HIP events): "
                                          "what of the gradient exchange the backward pass did not hide"}
                    if multi else None),
            "config": {"workload": (f"Counter-Strike latents {T}-frame seq, EDM2 UNet 310.0M (cs_train.py:35-45), " if cs else
                                    f"Lunar-Lander {T}-frame seq, gym EDM2 UNet 46.2M (gym_train.py:37-47), ") +
                                   f"{B} seq/GPU" + (" (cs_train.py:59 micro_batch_size)" if (cs and B == 2) else
                                                     " (gym_train.py:55 micro_batch_size)" if (not cs and B == 8) else "") +
                                   f", step = EDM2Loss fwd + bwd + grad all-reduce + [grad-norm clip +] AdamW + 2 EMA profiles, "
                                   f"3:1 mix of 3-D/2-D steps", "global_batch": world * B, "seq_len": T,
                       "parallelism": f"dp{world}", "hip_graph": False,
                       **({"accum_NOT_THE_HEADLINE": accum, "lr": opt.param_groups[0]["lr"]} if accum > 1 else {}),
                       **({"wrapper_NOT_THE_HEADLINE": "torch: the reference loop as written (torch DDP when multi-rank, torch.optim.AdamW, "
                                                       "clip_grad_norm_, deep-copied EMA trackers, synced loss)"} if torch_loop else {}),
                       **({"shared_gpu_gloo_NOT_A_MEASUREMENT": True} if share else {}),
                       **({"only_mode_NOT_THE_METRIC": _only} if _only else {}),
                       **{k: round(v, 2) for k, v in per_mode.items()}},
            "loss": loss_val, "roofline": roof, "roofline_step": roof_step, "roofline_step_2d": roof_step_2d,
            "roofline_attention": roof_attn,
            "roofline_attention_bwd": roof_attn_bwd, "kernels": kernels}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=None,
                    help="sequences per GPU (weak scaling).  Default: the reference's own micro-batch -- 8 for the Lunar-Lander "
                         "net (gym_train.py:55 micro_batch_size), 2 for the Counter-Strike net (cs_train.py:59); rollout: 1")
    ap.add_argument("--frames", type=int, default=None, help="frames per sequence (default 64 gym / 32 cs)")
    ap.add_argument("--net", choices=["gym", "cs"], default="gym",
                    help="gym = BASELINE configs[1] (the headline metric); cs = the Counter-Strike net of configs[2]/[3] "
                         "(32x32 latents, 310 M parameters, no conditioning) as an extra measurement")
    ap.add_argument("--extra-rollout-frames", type=int, default=256,
                    help="generated frames of the `extra.rollout_<n>` record (BASELINE configs[4]: 256)")
    ap.add_argument("--cpu-frames", type=int, default=64,
                    help="frames of the CPU-baseline sample: 64 = BASELINE configs[1] at B = 1 (BASELINE.md section 3: about two "
                         "minutes of host time); smaller = a shorter sample; 0 = skip")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the `extra` records (Counter-Strike T = 32 training steps, a KV-cached rollout) the default "
                         "1-GPU gym run appends to its JSON line")
    ap.add_argument("--mode", choices=["train", "rollout"], default="train",
                    help="train = the BASELINE headline metric (default); rollout = config 5 (KV-cached sampler), extra line")
    ap.add_argument("--gen-frames", type=int, default=8)
    ap.add_argument("--ctx-frames", type=int, default=8, help="rollout: frames of the prefill (+ 2 warm-up frames) before the timed ones")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--watchdog", type=float, default=300.0,
                    help="multi-rank runs: seconds without progress after which a rank exits 124 naming its stage (0 = off)")
    ap.add_argument("--wrapper", choices=["oniris", "torch"], default="oniris",
                    help="oniris = OnirisDDP + the fused optimizer / EMA pass (the headline); torch = the reference's loop as written: "
                         "torch DistributedDataParallel around the UNet (when there is a process group), torch.optim.AdamW, "
                         "clip_grad_norm_, two deep-copied EMA trackers, the loss with its host round trip -- an extra measurement of "
                         "the drop-in path (cs_train.py / gym_train.py with no changed line)")
    ap.add_argument("--accum", type=int, default=1,
                    help="gradient accumulation as in the reference loops (gym_train.py:96-112, cs_train.py:105-127): the optimizer "
                         "(+ clip, EMA, learning-rate schedule) runs every K-th micro-step, the K-1 others run their backward "
                         "under no_sync() (no gradient exchange).  An extra measurement, NOT the headline (K = 1)")
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 64 if args.net == "gym" else 32
    if args.batch is None:
        args.batch = 1 if args.mode == "rollout" else (8 if args.net == "gym" else 2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.mode == "train":
        sys.exit(self_launch(args))              # (this process never imports torch, never touches a GPU)
    if args.mode == "rollout":
        for k, v in ROLLOUT_ENV.items():         # before anything touches the HIP runtime (torch is imported inside rollout())
            os.environ.setdefault(k, v)
    _claim_stdout()
    if args.mode == "rollout":
        return rollout(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = bool(os.environ.get("ONIRIS_FORCE_DIST"))          # debug: run the RCCL/DDP path with a single rank
    # the rollout records of the default run are measured FIRST, each in a child process of its own (ROLLOUT_ENV), while this process
    # has not touched the GPU yet: a child is started (fork + exec) only from a process without HIP state
    want_extra = (rank == 0 and world == 1 and not force_dist and args.net == "gym" and not args.no_extra and args.accum == 1
                  and not os.environ.get("ONIRIS_ONLY_MODE") and not args.dry_run and args.gpus == 1)
    pre_extra = {}
    if want_extra:
        nroll = int(args.extra_rollout_frames)
        try:
            pre_extra[f"rollout_{nroll}"] = rollout_child(1, nroll)
            pre_extra[f"rollout_{nroll}"]["note"] = (f"configs[4]: {nroll} generated frames, plotting.py:165 settings (16 Heun steps = 31 UNet "
                                                     f"evaluations per frame), one sequence, KV / activation caches growing from 10 to {10 + nroll} "
                                                     "frames; measured in a process of its own (bench.ROLLOUT_ENV)")
            r8 = rollout_child(8, 8)
            pre_extra["rollout_b8"] = {k: r8[k] for k in ("value", "unit", "ms_per_unet_eval", "frames_generated", "batch", "finite", "runtime_env")}
            pre_extra["rollout_b8"]["note"] = f"the same sampler on 8 sequences at once (throughput; rollout_{nroll} is the one-sequence latency case)"
        except Exception as e:                                   # (never lose the headline to an extra)
            pre_extra["rollout_error"] = f"{type(e).__name__}: {e}"

    import torch
    import torch.distributed as dist
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
    wd = Watchdog(rank, args.watchdog if (world > 1 or force_dist) else 0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        wd.beat(f"init_process_group({'gloo' if share else 'nccl'}) on device {local}")
        if share:
            dist.init_process_group("gloo", init_method="env://")
        else:
            dist.init_process_group("nccl", init_method="env://", device_id=dev)

    out = train(args, args.net, args.steps, args.warmup, rank, world, dev, wd)
    single = rank == 0 and world == 1 and not force_dist
    if single and args.net == "gym" and not args.no_extra and args.accum == 1 and not os.environ.get("ONIRIS_ONLY_MODE"):
        # BASELINE configs[2] and [4] next to the headline, so that their figures are witnessed by the same run:
        # the Counter-Strike net at its own 32 frames (8 timed steps, ~1 s) and a 32-frame KV-cached rollout (~2 s)
        import types
        extra = {}
        try:
            extra["cs_t32"] = train(args, "cs", 8, 4, rank, world, dev, wd, light=True, light_batch=2)   # cs_train.py:59
            torch.cuda.empty_cache()
            if args.batch != 2:      # rounds 1-3 quoted the headline at 2 sequences per GPU: kept for round-over-round comparison
                extra["gym_t64_b2"] = train(args, "gym", 8, 4, rank, world, dev, wd, light=True, light_batch=2)
                torch.cuda.empty_cache()
            # BASELINE configs[3]'s per-GPU share: the 310 M net on 64-frame sequences, cs_train.py:59's 2 sequences per GPU
            extra["cs_t64"] = train(args, "cs", 8, 4, rank, world, dev, wd, light=True, light_batch=2, light_frames=64)
            torch.cuda.empty_cache()
            extra.update(pre_extra)                                 # BASELINE configs[4]: measured before this process touched the GPU (above)
        except Exception as e:                                   # (never lose the headline to an extra)
            extra["error"] = f"{type(e).__name__}: {e}"
        out["extra"] = extra
    if single and args.cpu_frames > 0 and args.net == "gym":
        out["cpu_baseline"] = cpu_baseline(args.cpu_frames)
    elif rank == 0:
        out["cpu_baseline"] = None
    if world > 1:
        wd.beat("final barrier")
        dist.barrier()
    wd.stop()
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
