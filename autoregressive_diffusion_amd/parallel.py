"""Batch-sharded data parallelism for the denoiser step: one process per GPU, gradients summed with RCCL
(torch.distributed backend "nccl" == RCCL on ROCm) over the xGMI mesh.

Replaces `DDP(unet, device_ids=[local_rank], find_unused_parameters=True)` + `unet.no_sync()` of the reference
(cs_train.py:53-54,108).  Design for MI355X: parameters and gradients live in ONE flat fp32 buffer each
(FlatParams), so the gradient exchange is a handful of large all-reduces on contiguous memory (xGMI ring
collectives are per-link bound: few large messages beat many 25 MB buckets), parameters that received no gradient
(2-D steps, `out_res.*`, `emb_time`) are simply zeros in the flat buffer (no unused-parameter bookkeeping, no
deadlock), and the optimizer is a single fused kernel over the same buffers.
"""
import contextlib
import torch
import torch.distributed as dist
from torch import nn


class ParamPack:
    """See FlatParams.direct_pack."""

    def __init__(self, flat, idx, ids):
        self.flat, self.idx, self.ids = flat, idx, ids

    def values(self):
        return self.flat.flat.index_select(0, self.idx)

    def deliver(self, grad):
        """grad: flat fp32 tensor in pack order; accumulates like autograd would."""
        self.flat.grad.index_add_(0, self.idx, grad.reshape(-1))
        self.flat._got.update(self.ids)


class FlatParams:
    """Re-homes every trainable parameter of `module` (and its .grad) as a view into one flat fp32 buffer."""

    _registry = {}                 # id(parameter) -> weakref to the FlatParams that re-homed it (ops.direct_pack)
    SEG_ALIGN = 3360               # = 2^5 * 3 * 5 * 7 elements: a segment splits evenly, in 16-byte units, over 1..8 ranks

    _instances = None              # WeakSet of the live FlatParams

    @classmethod
    def owner_of(cls, param):
        ref = cls._registry.get(id(param))
        return ref() if ref is not None else None

    @classmethod
    def live(cls):
        return list(cls._instances) if cls._instances is not None else []

    def __init__(self, module, lazy_small=False, only=None):
        """only: an iterable of parameters -- re-home just these (torch's own DistributedDataParallel around the net:
        BetterModule._ddp_params_and_buffers_to_ignore hands the kernel-owned weights to an inner OnirisDDP and leaves every
        other parameter an ordinary torch parameter for torch's reducer).
        lazy_small: gradients of the parameters autograd itself accumulates (everything except the conv weights,
        whose .grad the HIP weight_bwd kernel writes through a raw pointer) are NOT accumulated into their flat slice
        one tiny `add_` kernel per parameter (~190 launches per step for the gym net); their .grad is None during
        backward, so autograd just hands the tensor over, and `gather()` adds all of them into the flat buffer with one
        multi-tensor call (called by OnirisDDP at the end of backward and by FlatAdamW.step)."""
        only = None if only is None else {id(p) for p in only}
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and (only is None or id(p) in only)]
        self.params = [p for _, p in named]
        assert self.params, "no trainable parameters"
        self.module = module
        self.names = {id(p): n for n, p in named}          # state_dict keys of the re-homed parameters
        self.orig_params = list(self.params)                # module.parameters() order (the order a torch optimizer sees)
        # parameters whose gradient is final early in backward go, stage by stage, contiguous, BEHIND the rest of the
        # buffers: OnirisDDP exchanges a stage's [lo, hi) while the rest of backward is still running.  The module names
        # the stages in the order they become final: _oniris_overlap_stages() -> [(hook key, [parameters])], or the
        # one-stage form _oniris_overlap_plan() -> (hook key, [parameters]).
        # (the module itself, or the first submodule that names stages: a Precond around the UNet)
        self.stage_module = stager = next((m for m in module.modules() if hasattr(m, "_oniris_overlap_stages")
                                           or hasattr(m, "_oniris_overlap_plan")), None)
        if stager is not None and hasattr(stager, "_oniris_overlap_stages"):
            plan = [(k, list(ps)) for k, ps in (stager._oniris_overlap_stages() or []) if ps]
        elif stager is not None:
            one = stager._oniris_overlap_plan()
            plan = [(one[0], list(one[1]))] if one is not None and one[1] else []
        else:
            plan = []
        # parameters that receive a gradient on the same steps sit next to each other (module hint: 0 = every step,
        # 1 = 3-D steps only, 2 = never), so that FlatAdamW -- which, like torch.optim, skips parameters without a
        # gradient -- covers the buffer with a handful of contiguous launches.  Correctness never depends on the hint.
        classer = next((m for m in module.modules() if hasattr(m, "_oniris_param_classes")), None)
        cls = classer._oniris_param_classes() if classer is not None else {}
        by_class = lambda ps: sorted(ps, key=lambda p: cls.get(id(p), 0))          # (stable)
        mine = {id(p) for p in self.params}
        staged, segs = set(), []
        for key, ps in plan:
            ps = [p for p in ps if id(p) in mine and id(p) not in staged]
            staged.update(id(p) for p in ps)
            if ps:
                segs.append((key, by_class(ps)))
        head = by_class([p for p in self.params if id(p) not in staged])
        self.params = head + [p for _, ps in segs for p in ps]
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.dtype == dt and p.device == dev for p in self.params)
        # every segment (head, stages) starts on a multiple of SEG_ALIGN elements: divisible by every world size up to 8
        # with 16-byte aligned shares -- the mesh exchange (OnirisDDP exchange="mesh") deals a segment out in equal chunks
        self.offsets, off, seg_lo = [], 0, [0]
        first_of_seg = {id(ps[0]) for _, ps in segs}
        for p in self.params:
            if id(p) in first_of_seg:
                off = (off + self.SEG_ALIGN - 1) // self.SEG_ALIGN * self.SEG_ALIGN
                seg_lo.append(off)
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4                      # 16-byte aligned slices
        off = (off + self.SEG_ALIGN - 1) // self.SEG_ALIGN * self.SEG_ALIGN
        self.numel = off
        seg_hi = seg_lo[1:] + [off]
        self.head = (0, seg_lo[1] if segs else off)               # [lo, hi) of everything that is exchanged at the end
        self.stages = [(key, lo, hi) for (key, _), lo, hi in zip(segs, seg_lo[1:], seg_hi[1:])]      # firing order
        self.stage_at = self.stages[0][0] if self.stages else None
        self.tail_start = self.stages[0][1] if self.stages else off
        self.flat = torch.zeros(off, dtype=dt, device=dev)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.flat[o:o + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                p.grad = self.grad[o:o + p.numel()].view_as(p)
        import weakref
        me = weakref.ref(self)
        ids = [id(p) for p in self.params]
        for i in ids:
            FlatParams._registry[i] = me
        if FlatParams._instances is None:
            FlatParams._instances = weakref.WeakSet()
        FlatParams._instances.add(self)

        def _forget(reg=FlatParams._registry, ids=ids, me=me):      # the registry is keyed by id(): drop the entries with
            for i in ids:                                           # their owner, or a recycled id would find a dead /
                if reg.get(i) is me:                                # foreign FlatParams
                    del reg[i]
        weakref.finalize(self, _forget)
        self._lazy = []
        self._owner = {}                   # kernel-owned weight -> its NormalizedWeight module (.pw.touched: a wgrad ran)
        self._got = set()                  # autograd-owned (lazy) parameters that received a gradient since take_active()
        try:
            from .edm2.conv import NormalizedWeight
            self._owner = {id(m.weight): m for m in module.modules() if isinstance(m, NormalizedWeight)}
        except ImportError:                # (toy modules in the CPU tests)
            pass
        self._direct = {}                  # id(parameter) -> True: a fused backward kernel adds its gradient into self.grad
        if lazy_small:
            self._lazy = [(p, p.grad) for p in self.params if id(p) not in self._owner]
            for p, _ in self._lazy:
                p.grad = None

    def direct_pack(self, params):
        """A ParamPack over `params` (all re-homed here): their values as one gathered vector, their gradients added
        straight into the flat gradient buffer by one index_add_ -- no autograd node, no AccumulateGrad, no per-parameter
        Python work per step (the ~270 scalar parameters of the gates and emb_gain's cost ~2 ms of host time per step
        that way).  Their .grad stays the permanent view into the flat buffer (like the kernel-owned conv weights)."""
        offs = []
        for p in params:
            o = self.offset_of(p)
            offs.extend(range(o, o + p.numel()))
        idx = torch.tensor(offs, dtype=torch.int64, device=self.flat.device)
        ids = [id(p) for p in params]
        views = {id(p): v for p, v in self._lazy}
        for p in params:
            self._direct[id(p)] = True
            if id(p) in views:
                p.grad = views[id(p)]
        self._lazy = [(p, v) for p, v in self._lazy if id(p) not in self._direct]
        return ParamPack(self, idx, ids)

    def zero_grad(self):
        self.grad.zero_()
        for p, _ in self._lazy:
            p.grad = None
        if getattr(self, "_foreign", False):               # a foreign optimizer's zero_grad released views at some point: re-alias
            self._foreign = False
            lazy = {id(p) for p, _ in self._lazy}
            for p, o in zip(self.params, self.offsets):
                if id(p) not in lazy and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
                    p.grad = self.grad[o:o + p.numel()].view_as(p)

    # ---- gradients released by somebody else: `torch.optim.AdamW(...).zero_grad()` (set_to_none=True is torch's default) is what
    # the reference's loops call (gym_train.py:72,108, cs_train.py:77,121).  It sets every .grad to None -- also the permanent views
    # into the flat gradient buffer -- and does NOT clear the buffer.  The next backward then creates fresh gradient tensors
    # outside the buffer (autograd for its own parameters, WeightBank._ensure for the kernel-owned weights).  OnirisDDP notices
    # (foreign_zero_grad), clears the buffer once per cycle (begin_foreign_cycle), and moves the fresh gradients into their
    # slices before a slice is exchanged (adopt); parameters that received nothing keep .grad = None, so that the foreign
    # optimizer skips them exactly as it would without this class.
    def foreign_zero_grad(self):
        lazy = getattr(self, "_lazy_ids", None)
        if lazy is None or len(lazy) != len(self._lazy):
            lazy = self._lazy_ids = {id(p) for p, _ in self._lazy}
        seen = False
        for p in self.params:
            if id(p) in lazy or id(p) in self._direct:
                continue
            if p.grad is not None:
                return False
            seen = True
        return seen

    def begin_foreign_cycle(self):
        self.grad.zero_()
        self._got.clear()                  # (direct packs that deliver a gradient in this cycle: ParamPack.deliver)
        self._foreign = True
        # The kernel-owned weights get their flat slices back as .grad RIGHT AWAY: weight_bwd then writes into the buffer that is
        # exchanged (no 185 MB of fresh zero tensors per cycle, no copy into the slices afterwards) and the WeightBank's descriptor
        # table keeps its pointers (no rebuild + upload per step).  Which of them really received a gradient is known at the end of
        # the cycle (PackedWeight.touched, host side): adopt() hands the others back as None, so that the foreign optimizer skips
        # them as it does in the reference.
        for p, o in zip(self.params, self.offsets):
            m = self._owner.get(id(p))
            if m is not None and p.grad is None:
                p.grad = self.grad[o:o + p.numel()].view_as(p)
                if m.pw is not None:
                    m.pw.touched = False

    def adopt(self, lo=None, hi=None):
        """Foreign cycle only: the gradients that live outside the flat buffer move into their slices of [lo, hi) (everything by
        default) and .grad is re-aliased; a parameter without a gradient keeps None (its slice is zero)."""
        if not getattr(self, "_foreign", False):
            return
        lazy = self._lazy_ids if getattr(self, "_lazy_ids", None) is not None else {id(p) for p, _ in self._lazy}
        base = self.grad.data_ptr()
        src, dst = [], []
        for p, o in zip(self.params, self.offsets):
            if (lo is not None and o < lo) or (hi is not None and o >= hi) or id(p) in lazy:
                continue                                   # (lazy parameters: gather() below, same rule)
            view = self.grad[o:o + p.numel()].view_as(p)
            g = p.grad
            if g is None:
                # a fused backward kernel added into the slice itself -- IF its pack ran a backward in this cycle (gates on a
                # 2-D-only cycle did not: .grad stays None and the foreign optimizer skips them, as in the reference)
                if id(p) in self._direct and id(p) in self._got:
                    p.grad = view
                continue
            if g.data_ptr() != base + 4 * o:
                src.append(g); dst.append(view)
                p.grad = view
            elif lo is None and hi is None:
                m = self._owner.get(id(p))         # (final adopt of a backward: a kernel-owned weight no wgrad launch targeted in
                if m is not None and m.pw is not None and not m.pw.touched:       # this cycle has no gradient)
                    p.grad = None
        if src:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)

    def offset_of(self, param):
        if not hasattr(self, "_off_by_id"):
            self._off_by_id = {id(p): o for p, o in zip(self.params, self.offsets)}
        return self._off_by_id[id(param)]

    def slice_of(self, buf, param):
        """The part of `buf` (any tensor laid out like self.flat: optimizer moments, EMA copies) that belongs to `param`."""
        o = self.offset_of(param)
        return buf[o:o + param.numel()].view_as(param)

    def gather(self):
        """Add the autograd-owned gradients into their flat slices (one multi-tensor add) and re-alias .grad."""
        src, dst = [], []
        foreign = getattr(self, "_foreign", False)
        for p, view in self._lazy:
            g = p.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                src.append(g.reshape(view.shape)); dst.append(view)
                self._got.add(id(p))
            if g is not None or not foreign:               # (foreign optimizer: None stays None -- "no gradient, skip me")
                p.grad = view
        if src:
            with torch.no_grad():
                torch._foreach_add_(dst, src)

    def take_active(self):
        """[bool per parameter, flat order]: did it receive a gradient since the last call?  (What torch.optim asks with
        `p.grad is None`.)  Kernel-owned weights: a weight-gradient launch targeted them (PackedWeight.touched, host
        side, no device sync); autograd-owned lazy parameters: autograd handed a gradient over (gather()); parameters
        whose .grad is a permanent view of the flat buffer cannot tell "none" from "zero" and count as active."""
        self.gather()
        kinds = getattr(self, "_kinds", None)
        if kinds is None or kinds[0] != (len(self._lazy), len(self._direct)):
            lazy = {id(p) for p, _ in self._lazy}
            kinds = self._kinds = ((len(self._lazy), len(self._direct)),
                                   [(self._owner.get(id(p)), id(p), id(p) in lazy or id(p) in self._direct) for p in self.params])
        out, got = [], self._got
        for m, pid, tracked in kinds[1]:                       # (runs every step between backward and the optimizer launch)
            pw = m.pw if m is not None else None
            if pw is not None:
                out.append(bool(pw.touched)); pw.touched = False
            elif tracked:
                out.append(pid in got)
            else:
                out.append(True)
        got.clear()
        return out

    def check(self):
        """True while every parameter still aliases the flat buffers (a .to()/deepcopy breaks the aliasing)."""
        base = self.flat.data_ptr()
        lazy = {id(p) for p, _ in self._lazy}
        return all(p.data_ptr() == base + 4 * o and (id(p) in lazy and p.grad is None or p.grad is not None and
                   p.grad.data_ptr() == self.grad.data_ptr() + 4 * o) for p, o in zip(self.params, self.offsets))


class OnirisDDP(nn.Module):
    """Data-parallel wrapper: forward delegates to `module`; during / at the end of every backward pass (unless inside
    `no_sync()`) the flat gradient buffer is averaged over the process group.

    Overlap.  The module names STAGES (FlatParams.stages): groups of parameters whose gradients are final when the
    backward pass reaches a given activation (UNet: the decoder and deep-encoder levels, which hold the parameters, are
    done long before the activation-heavy outer levels).  A tensor hook per stage turns the pending weight-gradient slabs
    into parameter gradients and starts that stage's exchange on RCCL's stream, beside the remaining backward kernels;
    the rest (`FlatParams.head`) follows at the end of backward.

    exchange = "allreduce": one (bucketed) all-reduce per stage -- RCCL's ring / tree over xGMI.
    exchange = "mesh":      reduce-scatter as ONE all-to-all per stage (every rank sends chunk r of the segment straight to
                            rank r: on the xGMI full mesh each of the 7 links carries 1/8 of the bytes, instead of a ring
                            pushing 7/8 of them through every link), a local sum, the optimizer on the OWNED chunks only
                            (1/world of the AdamW / EMA work and traffic per GPU), then one all-gather of the updated
                            parameters (FlatAdamW.step calls back: `after_step`).  SURVEY section 5: 1.24 GB of Counter-Strike
                            gradients = 14.2 ms as a ring all-reduce vs ~2 x 1 ms.
    grad_dtype = torch.bfloat16: the gradients travel as bf16 (half the bytes; fp32 master gradients are kept, the
                            average is rounded once per element)."""

    def __init__(self, module, process_group=None, bucket_mb=256, flat=None, exchange=None, grad_dtype=None,
                 force_collectives=False, auto_wait=True, device_ids=None, output_device=None, find_unused_parameters=None,
                 broadcast_buffers=None, gradient_as_bucket_view=None, static_graph=None, torch_ddp=None):
        """torch_ddp: the torch.nn.parallel.DistributedDataParallel instance this object works UNDER (inner mode, built by
        BetterModule._ddp_params_and_buffers_to_ignore while torch's constructor runs): torch's reducer owns every autograd-
        accumulated parameter, this object the kernel-owned weights in `flat`; nobody calls its forward() -- the wrapped
        module's own forward calls inner_pre_forward() / inner_post_forward(out) -- and "are gradients exchanged in this
        pass" is torch DDP's `require_backward_grad_sync` (its no_sync()) as it stood when the FORWARD ran -- torch's own rule (_pass_synced).
        force_collectives: issue the collectives in a one-rank group too (tests / profiling of the exchange path).
        auto_wait: the end of every synced backward also orders the current stream behind the exchange (a stream-side wait,
        the host does not block), so `optimizer.step()` may follow `loss.backward()` directly, as in the reference's loops;
        False: the caller places `wait()` itself (bench.py brackets it with events).
        device_ids / output_device / find_unused_parameters / broadcast_buffers / gradient_as_bucket_view / static_graph:
        torch.nn.parallel.DistributedDataParallel's keyword arguments, accepted so that `DDP(unet, device_ids=[local_rank],
        output_device=local_rank, find_unused_parameters=True)` (cs_train.py:53-54) works with `DDP = OnirisDDP`; none of them
        changes anything here (one device per process; parameters without a gradient are always allowed; buffers are
        broadcast once at construction)."""
        super().__init__()
        import os
        self.module = module
        import weakref
        self._torch_ddp = weakref.ref(torch_ddp) if torch_ddp is not None else None
        self.auto_wait = bool(auto_wait)
        self.force_collectives = bool(force_collectives)
        self.process_group = process_group
        self.flat = flat if flat is not None else FlatParams(module)
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) // 4))
        self.exchange = exchange or os.environ.get("ONIRIS_DDP_EXCHANGE", "allreduce")
        assert self.exchange in ("allreduce", "mesh"), self.exchange
        if grad_dtype is None and os.environ.get("ONIRIS_DDP_BF16"):
            grad_dtype = torch.bfloat16
        self.grad_dtype = grad_dtype
        self._sync_flag = True
        self._queued = False
        self._fwd_synced = True            # inner mode: the flag torch DDP's forward saw (torch prepares its reducer there)
        self.comm_cus = int(os.environ.get("ONIRIS_COMM_CUS", "0"))    # CUs left to RCCL while an exchange is in flight
        self._reserved = False
        self._bank_holder = None
        self._works = []                   # [(work, finish callable | None)]
        self._sent = [False] * len(self.flat.stages)
        self._g16 = None                   # bf16 transport buffer (grad_dtype)
        self._recv = {}                    # mesh: receive buffers per segment
        # FlatAdamW skips parameters that received no gradient, decided from rank-LOCAL bookkeeping (take_active), while the
        # exchange averages the whole flat gradient: ranks that ran different step kinds (just_2d on one rank only,
        # conditioning on some) would silently diverge.  The reference loops use i % 4 on every rank; this guard turns
        # a violation into an error: the bitmaps are compared on the first steps and every `active_check_every`-th.
        self.active_check_every, self._opt_steps = 100, 0
        self.flat._active_check = self._check_active
        if self.flat.stages:                             # early exchanges (see FlatParams / _stage)
            hooks = {key: self._make_stage(i) for i, (key, _, _) in enumerate(self.flat.stages)}
            stager = getattr(self.flat, "stage_module", None) or module       # (a Precond around the UNet: the UNet)
            stager.__dict__["_oniris_stage_hooks"] = hooks
            stager.__dict__["_oniris_stage_at"] = self.flat.stages[0][0]       # (one-stage form, toy modules)
            stager.__dict__["_oniris_stage_cb"] = hooks[self.flat.stages[0][0]]
            self._stager = stager
        # every rank starts from rank 0's parameters AND buffers (what torch DDP does at construction: MPFourier's
        # random freqs / phases are buffers, utils.py:63-64 -- ranks built from different RNG states would otherwise
        # keep different noise / time embeddings under shared weights)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            dist.broadcast(self.flat.flat, src=0, group=self.process_group)
            self._broadcast_buffers()                    # (inner mode too: the buffers are named in torch DDP's ignore list)
        if self.exchange == "mesh" and self._active():
            world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
            segs = [self.flat.head] + [(lo, hi) for _, lo, hi in self.flat.stages]
            self.flat._owned_ranges = [(lo + rank * ((hi - lo) // world), lo + (rank + 1) * ((hi - lo) // world))
                                       for lo, hi in segs if hi > lo]
            self.flat._after_step = self._allgather_params
            self.flat._norm_reduce = self._sum_over_ranks

    def _broadcast_buffers(self):
        """Rank 0's module buffers to every rank: one coalesced broadcast per dtype, copied back in place."""
        by_dtype = {}
        for b in self.module.buffers():
            if b is not None and b.numel():
                by_dtype.setdefault((b.dtype, b.device), []).append(b)
        for (dt, dev), bufs in by_dtype.items():
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            if dt == torch.bool:
                flat = flat.to(torch.uint8)
            dist.broadcast(flat, src=0, group=self.process_group)
            off = 0
            with torch.no_grad():
                for b in bufs:
                    b.copy_(flat[off:off + b.numel()].view_as(b).to(dt))
                    off += b.numel()

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)

    @property
    def _sync_enabled(self):
        if self._torch_ddp is not None:
            outer = self._torch_ddp()
            if outer is None:
                raise RuntimeError("the torch DistributedDataParallel wrapper of this network is gone, but the network is still "
                                   "being trained: its kernel-owned weight gradients would no longer be exchanged -- wrap it again")
            return bool(outer.require_backward_grad_sync) and self._sync_flag
        return self._sync_flag

    @_sync_enabled.setter
    def _sync_enabled(self, v):
        self._sync_flag = bool(v)

    def _pass_synced(self):
        """Is the backward pass that is running now one whose gradients are exchanged?
        Inner mode (under torch's DistributedDataParallel): torch's own rule -- the value `require_backward_grad_sync` had when
        the FORWARD ran (torch prepares its reducer in forward and nowhere else; distributed.py `_pre_forward`).  cs_train.py:108-110
        wraps only `loss.backward()` in `unet.no_sync()`, so under torch DDP every micro-step of that loop is exchanged, and so it is
        here: both parameter groups (torch's reducer's and the kernel-owned weights) are in the same state after every backward,
        and an `optimizer.step()` behind a backward that ran inside `no_sync()` sees averaged gradients on all of them, as with
        plain torch DDP.  A forward inside `no_sync()` defers the exchange to the next synced pass -- again as torch.
        Stand-alone OnirisDDP: its own `no_sync()` is honoured at forward AND at backward time (documented in INTEGRATION.md)."""
        if self._torch_ddp is not None:
            return self._fwd_synced
        return self._sync_enabled and self._fwd_synced

    def _bank(self):
        """The WeightBank of the wrapped tree (it lives on the module whose forward entered `weights_ready` first: the UNet,
        also when a Precond around it is what got wrapped)."""
        m = self._bank_holder
        if m is not None:
            b = m.__dict__.get("_oniris_bank")
            if b is not None:
                return b
        for m in self.module.modules():
            b = m.__dict__.get("_oniris_bank")
            if b is not None:
                self._bank_holder = m
                return b
        return None

    # ---- inner mode (under torch's DistributedDataParallel): called by the wrapped module's forward
    def inner_pre_forward(self):
        if torch.is_grad_enabled() and self.flat.foreign_zero_grad():
            self.flat.begin_foreign_cycle()

    def inner_post_forward(self, out):
        self._fwd_synced = self._sync_enabled
        if torch.is_grad_enabled() and self._fwd_synced:
            first = out[0] if isinstance(out, (tuple, list)) else out
            if torch.is_tensor(first) and first.requires_grad:
                first.register_hook(self._on_backward_start)

    def forward(self, *args, **kwargs):
        if torch.is_grad_enabled() and self.flat.foreign_zero_grad():
            if self.exchange == "mesh":
                raise RuntimeError("OnirisDDP(exchange='mesh') shards the optimizer: it needs FlatAdamW (and its zero_grad()), "
                                   "not a torch.optim optimizer -- use the default exchange='allreduce' with torch.optim")
            self.flat.begin_foreign_cycle()
        out = self.module(*args, **kwargs)
        if torch.is_grad_enabled() and self._sync_enabled:
            first = out[0] if isinstance(out, (tuple, list)) else out
            if torch.is_tensor(first) and first.requires_grad:
                first.register_hook(self._on_backward_start)
        return out

    def _on_backward_start(self, grad):
        if not self._queued:
            self._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        return grad

    def _active(self):
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.process_group) > 1 or getattr(self, "force_collectives", False)

    def _make_stage(self, i):
        def hook(grad):
            """Tensor hook on the activation that ends stage i (fires in the middle of backward): the weights of every
            block behind it have all their wgrad slabs, so turn them into parameter gradients now (weight_bwd skips
            weights with no pending slab) and start their exchange; RCCL runs it on its own stream, beside the backward
            kernels that are still to come.  Stages that did not fire by themselves (their activation needed no
            gradient) go out with the next one."""
            if self._pass_synced() and self._active() and not self._sent[i]:
                bank = self._bank()
                if bank is not None:
                    bank.backward()
                for j in range(i + 1):
                    if not self._sent[j]:
                        _, lo, hi = self.flat.stages[j]
                        self.flat.adopt(lo, hi)
                        self._exchange(lo, hi, f"stage{j}")
                        self._sent[j] = True
            return None
        return hook

    def _stage(self, grad):                              # (one-stage form kept for callers of the round-2 API)
        return getattr(self, "_stager", self.module).__dict__["_oniris_stage_cb"](grad)

    def _end_of_backward(self):
        self._queued = False
        if not self._pass_synced():
            return
        self.allreduce_grads()                           # (finalises weight gradients + gathers the small ones first)
        if self.auto_wait:
            self.wait()

    def _reserve_cus(self, on):
        """ONIRIS_COMM_CUS=k: while an exchange is in flight the persistent kernels (one workgroup per CU) launch on k fewer
        CUs, so RCCL's workgroups do not have to share a CU's LDS with a convolution workgroup that fills it.  Host-side
        toggle = exactly the launches that can overlap the exchange: RCCL's stream waits for the compute stream at the
        point the collective is issued, and wait() orders everything issued after it behind the exchange."""
        k = self.comm_cus
        if k <= 0 or not self.flat.grad.is_cuda or on == self._reserved:
            return
        from . import ops
        ops.set_cu_reserve(k if on else ops_always_reserve())
        self._reserved = on

    def _exchange(self, lo, hi, label=None):
        if hi <= lo:
            return
        self._reserve_cus(True)
        g = self.flat.grad
        world = dist.get_world_size(self.process_group)
        nccl = g.is_cuda and dist.get_backend(self.process_group) == "nccl"    # RCCL averages in the collective; gloo has no AVG
        if self.exchange == "mesh":
            return self._exchange_mesh(lo, hi, world, label)
        buf = g
        if self.grad_dtype is not None:                  # bf16 transport: one cast pass each way, half the bytes on the links
            if self._g16 is None:
                self._g16 = torch.empty(self.flat.numel, dtype=self.grad_dtype, device=g.device)
            buf = self._g16
            buf[lo:hi].copy_(g[lo:hi])
        if not nccl:
            buf[lo:hi].mul_(1.0 / world)
        op = dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM
        for s in range(lo, hi, self.bucket_elems):
            e = min(hi, s + self.bucket_elems)
            w = dist.all_reduce(buf[s:e], op=op, group=self.process_group, async_op=True)
            fin = (lambda s=s, e=e: g[s:e].copy_(self._g16[s:e])) if buf is not g else None
            self._works.append((w, fin, label))

    def _exchange_mesh(self, lo, hi, world, label=None):
        """Reduce-scatter of segment [lo, hi) as ONE all-to-all (chunk r goes straight to rank r) + a local sum; the
        averaged chunk this rank owns lands in `flat.grad_reduced` (laid out like the flat buffers; only the owned ranges
        are ever written or read: the optimizer runs on them).  `flat.grad` itself stays purely LOCAL: an exchange that
        is not followed by zero_grad -- the reference loop's micro-step 0, cs_train.py:108,119: a synced backward with no
        optimizer step -- is followed by more local accumulation and another exchange, which must again see every
        rank's own sum (a reduced chunk left in place would be sent around a second time)."""
        g = self.flat.grad
        if getattr(self.flat, "grad_reduced", None) is None:
            self.flat.grad_reduced = torch.zeros_like(g)
        red = self.flat.grad_reduced
        rank = dist.get_rank(self.process_group)
        n = hi - lo
        assert n % world == 0, "segments are SEG_ALIGN-aligned: divisible by every world size up to 8"
        chunk = n // world
        dt = self.grad_dtype or g.dtype
        key = (lo, hi, dt)
        if key not in self._recv:
            self._recv[key] = (torch.empty(n, dtype=dt, device=g.device), torch.empty(n, dtype=dt, device=g.device) if dt != g.dtype else None)
        recv, send16 = self._recv[key]
        send = g[lo:hi]
        if send16 is not None:
            send16.copy_(send)
            send = send16
        w = dist.all_to_all_single(recv, send, group=self.process_group, async_op=True)

        def fin(recv=recv, lo=lo, chunk=chunk, world=world, rank=rank):
            own = red[lo + rank * chunk: lo + (rank + 1) * chunk]
            torch.sum(recv.view(world, chunk).float() if recv.dtype != g.dtype else recv.view(world, chunk), dim=0, out=own)
            own.mul_(1.0 / world)
        self._works.append((w, fin, label))

    def allreduce_grads(self):
        """Exchange whatever the stage hooks have not sent yet (everything, when none fired in this backward).
        Also the entry point after a backward that ran under `no_sync()` (bench.py --graph replays forward+backward
        from a hipGraph and exchanges eagerly): the weight gradients are finalised and the autograd-owned small
        gradients (gates, emb_gain, out_gain, grouped emb weights) are gathered into the flat buffer FIRST, so that
        they take part in the average instead of being added, un-averaged, by the optimizer's own gather()."""
        bank = self._bank()
        if bank is not None:
            bank._finish()                               # no-op unless a backward left it pending
        foreign = getattr(self.flat, "_foreign", False)
        self.flat.adopt()                                # (gradients a foreign zero_grad pushed outside the buffer)
        self.flat.gather()
        if foreign:
            self._check_foreign_active()
        sent, self._sent = self._sent, [False] * len(self.flat.stages)
        if not self._active():
            return
        for j, ((_, lo, hi), done) in enumerate(zip(self.flat.stages, sent)):
            if not done:
                self._exchange(lo, hi, f"stage{j}")
        self._exchange(*self.flat.head, "head")

    def _check_active(self, active):
        self._opt_steps += 1
        if not self._active() or (self._opt_steps > 2 and self._opt_steps % self.active_check_every):
            return
        self._compare_bitmaps(active)

    def _check_foreign_active(self):
        """A torch.optim optimizer skips a parameter whose .grad is None, decided per rank, while the exchange averages the
        whole buffer: ranks that ran different step kinds would update different sets of weights.  (torch's reducer makes
        "received a gradient" global for ITS parameters; this is the same guarantee for the kernel-owned ones, as a check:
        the None-bitmaps are compared on the first exchanges of foreign cycles and on every `active_check_every`-th.)"""
        self._foreign_syncs = getattr(self, "_foreign_syncs", 0) + 1
        if not self._active() or (self._foreign_syncs > 2 and self._foreign_syncs % self.active_check_every):
            return
        self._compare_bitmaps([p.grad is not None for p in self.flat.params])

    def _compare_bitmaps(self, active):
        import zlib
        h = zlib.crc32(bytes(bytearray(int(bool(a)) for a in active)))
        t = torch.tensor([h, -h], dtype=torch.int64, device=self.flat.grad.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.process_group)
        hi, lo = int(t[0]), -int(t[1])
        if hi != lo:
            raise RuntimeError("OnirisDDP: the ranks disagree about which parameters received a gradient in this step "
                               "(different step kinds per rank?  e.g. just_2d or conditioning on some ranks only): the "
                               "optimizer would update them on some ranks and skip them on others")

    def _sum_over_ranks(self, t):
        """mesh mode: FlatAdamW's sum of squares covers the owned chunks only -- add the other ranks' (in place)."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
        return t

    def _allgather_params(self):
        """mesh mode, called by FlatAdamW.step after the update of the owned chunks: every rank receives the other ranks'
        updated chunks of every segment (one all-gather per segment, straight over the mesh links)."""
        world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
        p = self.flat.flat
        for lo, hi in [self.flat.head] + [(a, b) for _, a, b in self.flat.stages]:
            if hi <= lo:
                continue
            chunk = (hi - lo) // world
            mine = p[lo + rank * chunk: lo + (rank + 1) * chunk].clone()       # (input must not alias the output)
            dist.all_gather_into_tensor(p[lo:hi], mine, group=self.process_group)

    def _allgather_like_params(self, buf):
        world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
        for lo, hi in [self.flat.head] + [(a, b) for _, a, b in self.flat.stages]:
            if hi <= lo:
                continue
            chunk = (hi - lo) // world
            mine = buf[lo + rank * chunk: lo + (rank + 1) * chunk].clone()
            dist.all_gather_into_tensor(buf[lo:hi], mine, group=self.process_group)

    def gather_state(self, optimizer):
        """mesh mode, COLLECTIVE (every rank calls it): complete the optimizer moments on every rank so that
        `optimizer.state_dict()` -- which the reference loop calls on rank 0 only, cs_train.py:146-159 -- describes all
        parameters.  Valid until the next optimizer step.  A no-op for exchange="allreduce" (the state is replicated)."""
        if self.exchange == "mesh" and self._active():
            with torch.no_grad():
                self._allgather_like_params(optimizer.m)
                self._allgather_like_params(optimizer.v)
        optimizer._state_complete_at = optimizer.steps

    def wait(self, timing=None):
        """Block the current stream until the gradient exchange is done (call before the optimizer step).
        timing: a dict -- per exchange label ("stage0" .. / "head") a list of (before, after) event pairs recorded on the
        current stream around that exchange's wait: how long the compute stream stood still for EACH stage (bench.py)."""
        for w, fin, label in self._works:
            if timing is not None and self.flat.grad.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            w.wait()
            if fin is not None:
                with torch.no_grad():
                    fin()
            if timing is not None and self.flat.grad.is_cuda:
                e1.record()
                timing.setdefault(label or "exchange", []).append((e0, e1))
        self._works = []
        self._reserve_cus(False)

    @contextlib.contextmanager
    def no_sync(self):
        old, self._sync_flag = self._sync_flag, False
        try:
            yield
        finally:
            self._sync_flag = old


def ops_always_reserve():
    import os
    return int(os.environ.get("ONIRIS_COMM_CUS_ALWAYS", "0"))


def power_function_exponent(std):
    """Exponent gamma of the power-function EMA profile with relative standard deviation `std` (EDM2, the relation
    edm2/phema.py:18-32 inverts): std^2 = (gamma + 1) / ((gamma + 2)^2 (gamma + 3)), solved by bisection on the
    decreasing branch gamma >= 0 (std <= 0.2886)."""
    f = lambda g: (g + 1.0) / ((g + 2.0) ** 2 * (g + 3.0))
    target = float(std) ** 2
    if not (0.0 < target <= f(0.0)):
        raise ValueError(f"relative std {std} outside (0, {f(0.0) ** 0.5:.4f}]")
    lo, hi = 0.0, 1.0
    while f(hi) > target:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if f(mid) > target else (lo, mid)
    return 0.5 * (lo + hi)


def power_function_beta(std, t_next, t_delta):
    """EMA decay of one update that advances training time from t_next - t_delta to t_next (edm2/phema.py:66-68)."""
    return (1.0 - t_delta / t_next) ** (power_function_exponent(std) + 1.0)


class FlatEMA:
    """PowerFunctionEMA (edm2/phema.py:90-106) on the flat parameter buffer: one fp32 copy per profile, updated
    inside the fused optimizer kernel (FlatAdamW.step(ema=...)) instead of one lerp_ launch per parameter and profile."""

    def __init__(self, flat, stds=(0.050, 0.100)):
        assert 1 <= len(stds) <= 2
        self.flat, self.stds = flat, list(stds)
        self.emas = [flat.flat.clone() for _ in stds]

    def reset(self):
        for e in self.emas:
            e.copy_(self.flat.flat)

    def weights(self, cur_nimg, batch_size):
        """[(ema buffer, 1 - beta)] for the update that ends at `cur_nimg` images."""
        return [(e, 1.0 - power_function_beta(s, cur_nimg, batch_size)) for e, s in zip(self.emas, self.stds)]

    def view(self, k, param):
        """The EMA value of `param` (a parameter re-homed in self.flat) under profile k."""
        return self.flat.slice_of(self.emas[k], param)

    def state_dict(self):
        """The reference's layout (edm2/phema.py:110-111): dict(stds, emas=[module-style state_dict per profile]) --
        every profile keyed like `module.state_dict()` of the module the flat buffer was built on (parameters from
        the EMA copy, buffers and frozen parameters from the live module, as `PowerFunctionEMA.get()` copies them)."""
        f = self.flat
        live = f.module.state_dict()
        by_name = {n: p for n, p in f.module.named_parameters()}
        out = []
        for e in self.emas:
            sd = {}
            for k, v in live.items():
                p = by_name.get(k)
                sd[k] = (f.slice_of(e, p) if p is not None and id(p) in f.names else v).detach().clone()
            out.append(sd)
        return dict(stds=list(self.stds), emas=out)

    def load_state_dict(self, state):
        assert list(state["stds"]) == list(self.stds), (state["stds"], self.stds)
        f = self.flat
        by_name = {n: p for n, p in f.module.named_parameters()}
        with torch.no_grad():
            for e, sd in zip(self.emas, state["emas"]):
                for k, v in sd.items():
                    p = by_name.get(k)
                    if p is not None and id(p) in f.names:
                        f.slice_of(e, p).copy_(v)


class FlatAdamW:
    """AdamW on the flat buffers, optionally with gradient-norm clipping and the EMA update in the same pass: fused HIP
    kernels (oniris_sqnorm + oniris_adamw_clip_ema).  CPU tensors are refused unless a test installed `cpu_update`."""

    cpu_update = None          # tests only: callable(opt, runs, grad_scale, max_norm, ema, owned, norm_reduce)

    def __init__(self, flat, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.flat = flat
        # one group, torch.optim style: the reference loops set the learning rate through it (gym_train.py:110-112)
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                                  maximize=False, params=list(range(len(flat.orig_params))))]
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.steps = 0                       # = max over the per-parameter step counters
        self.param_steps = [0] * len(flat.params)      # flat order (torch.optim keeps `step` per parameter too)
        self._norm_buf = None

    lr = property(lambda self: self.param_groups[0]["lr"], lambda self, v: self.param_groups[0].__setitem__("lr", v))
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def state_dict(self):
        """torch.optim.AdamW's layout (what gym_train.py:137-138 saves): per-parameter `step` / `exp_avg` /
        `exp_avg_sq`, indexed in module.parameters() order, + param_groups."""
        f = self.flat
        if getattr(f, "_owned_ranges", None) is not None and getattr(self, "_state_complete_at", None) != self.steps:
            raise RuntimeError("FlatAdamW.state_dict(): under OnirisDDP(exchange='mesh') every rank holds exp_avg / exp_avg_sq "
                               "of its OWN chunks only; call OnirisDDP.gather_state(optimizer) on EVERY rank first (a "
                               "collective), then state_dict() on whichever rank writes the checkpoint")
        pos = {id(p): k for k, p in enumerate(f.params)}
        st = {}
        for i, p in enumerate(f.orig_params):
            n = self.param_steps[pos[id(p)]] if len(self.param_steps) == len(f.params) else self.steps
            if n > 0:                                  # (torch has no state for a parameter that never had a gradient)
                st[i] = dict(step=torch.tensor(float(n)), exp_avg=f.slice_of(self.m, p).detach().clone(),
                             exp_avg_sq=f.slice_of(self.v, p).detach().clone())
        return dict(state=st, param_groups=[dict(g) for g in self.param_groups])

    def load_state_dict(self, state):
        """Accepts its own state_dict() and a torch.optim.AdamW state_dict over the same parameters in the same
        order (one group; every parameter must be at the same step -- the fused kernel keeps ONE step counter)."""
        f = self.flat
        groups = state["param_groups"]
        assert len(groups) == 1 and len(groups[0]["params"]) == len(f.orig_params), "one group over the same parameters"
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        st = state["state"]
        self.m.zero_(); self.v.zero_()
        pos = {id(p): k for k, p in enumerate(f.params)}
        self.param_steps = [0] * len(f.params)
        with torch.no_grad():
            for i, s in st.items():
                p = f.orig_params[int(i)]
                self.param_steps[pos[id(p)]] = int(float(s["step"]))
                f.slice_of(self.m, p).copy_(s["exp_avg"])
                f.slice_of(self.v, p).copy_(s["exp_avg_sq"])
        self.steps = max(self.param_steps, default=0)

    @torch.no_grad()
    def step(self, grad_scale=1.0, max_norm=None, ema=None):
        """max_norm: torch.nn.utils.clip_grad_norm_(params, max_norm) before the update (gym_train.py:105);
        ema: FlatEMA.weights(cur_nimg, batch_size) -> the tracked copies follow the updated parameters (:108).
        Like torch.optim.AdamW, a parameter WITHOUT a gradient in this step (2-D steps: the context weights and gates;
        never: out_res.*, emb_time; emb_label without conditioning) is skipped -- no moment decay, no weight decay, its
        own step counter stands still; only its EMA copies follow.  The flat buffer is covered by one launch per run of
        consecutive parameters with equal (has gradient, step count)."""
        f = self.flat
        active = f.take_active()
        chk = getattr(f, "_active_check", None)        # (set by OnirisDDP: all ranks must skip the same parameters)
        if chk is not None:
            chk(active)
        ema = list(ema or ())
        if len(self.param_steps) != len(f.params):
            self.param_steps = [self.steps] * len(f.params)
        runs = []                                      # (lo, hi, step): step = new Adam step count, 0 = no gradient
        for i, (p, o) in enumerate(zip(f.params, f.offsets)):
            if active[i]:
                self.param_steps[i] += 1
            st = self.param_steps[i] if active[i] else 0
            hi = f.offsets[i + 1] if i + 1 < len(f.params) else f.numel
            if runs and runs[-1][2] == st:
                runs[-1][1] = hi
            else:
                runs.append([o, hi, st])
        self.steps = max(self.param_steps)
        owned = getattr(f, "_owned_ranges", None)      # OnirisDDP(exchange="mesh"): this rank updates its chunks only
        if owned is not None:
            cut = []
            for lo, hi, st in runs:
                for a, b in owned:
                    x, y = max(lo, a), min(hi, b)
                    if x < y:
                        cut.append([x, y, st])
            runs = cut
        norm_reduce = getattr(f, "_norm_reduce", None)
        # mesh exchange: the reduced gradients of the owned chunks live in f.grad_reduced (f.grad stays rank-local)
        gsrc = self.grad_src = f.grad if owned is None else f.grad_reduced
        if not f.flat.is_cuda and FlatAdamW.cpu_update is None:
            # No CPU arithmetic in the product path: the host-side LOGIC of this class (runs, per-parameter step counters,
            # owned ranges, state_dict layout) is exercised by the CPU tests with a reference update they install themselves
            # (tests/cpu_reference_optimizer.py); without it CPU tensors are refused.
            raise RuntimeError("FlatAdamW needs HIP tensors (no CPU fallback in the product path; the CPU tests "
                               "install tests/cpu_reference_optimizer.py)")

        def update(runs_, max_norm_):
            if not f.flat.is_cuda:
                return FlatAdamW.cpu_update(self, runs_, grad_scale, max_norm_, ema, owned, norm_reduce)
            from . import ops
            for lo, hi, st in runs_:
                if st == 0 and not ema:
                    continue
                ops.adamw_(f.flat[lo:hi], gsrc[lo:hi], self.m[lo:hi], self.v[lo:hi], self.lr, self.betas[0],
                           self.betas[1], self.eps, self.weight_decay, st, grad_scale, max_norm_, self._norm_buf,
                           [(e[lo:hi], w) for e, w in ema], norm_ready=True)

        if f.flat.is_cuda and max_norm is not None:
            from . import ops
            if self._norm_buf is None:
                self._norm_buf = torch.zeros(1 + ops.SQNORM_WS, dtype=torch.float32, device=f.flat.device)
            if owned is None:
                ops.sqnorm_(gsrc, self._norm_buf)          # over the whole buffer (parameters without gradient hold zeros)
            else:                                          # sum of squares of the owned (reduced) chunks, then over the ranks
                tot = torch.zeros(1, dtype=torch.float32, device=f.flat.device)
                for a, b in owned:
                    ops.sqnorm_(gsrc[a:b], self._norm_buf)
                    tot += self._norm_buf[:1]
                self._norm_buf[:1].copy_(norm_reduce(tot))
        update(runs, max_norm)
        after = getattr(f, "_after_step", None)
        if after is not None:
            after()
        if owned is not None:
            # mesh exchange: m / v exist for the owned chunks only (state_dict() refuses until OnirisDDP.gather_state() has
            # collected them); the EMA copies are a function of the parameter trajectory alone, so every rank brings its
            # copies of the OTHER ranks' chunks up to date from the parameters the all-gather just delivered -- an EMA-only
            # pass (step 0) over the complement of the owned ranges: rank 0 can write the reference's checkpoint
            # (phema.py:110, cs_train.py:146-159) without any collective
            self._state_complete_at = None
            if ema:
                rest, pos = [], 0
                for a, b in sorted(owned):
                    if a > pos:
                        rest.append([pos, a, 0])
                    pos = max(pos, b)
                if pos < f.numel:
                    rest.append([pos, f.numel, 0])
                update(rest, None)

    def zero_grad(self):
        self.flat.zero_grad()
