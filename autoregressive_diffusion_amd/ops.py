"""Tensor-level wrappers over the C-ABI (include/oniris.h): torch only provides device memory, the current HIP
stream and autograd bookkeeping.  Activations are channels-last bf16: (N, H, W, C) with N = B*S*T frame-slots.

No CPU / PyTorch fallback exists for these ops: they raise if the tensors are not on a HIP device.
"""
import ctypes
import math
import numpy as np
import torch

from . import _lib
from ._lib import lib, check

BF16 = torch.bfloat16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as a raw handle (every kernel launch calls this: the raw getter
    avoids building a torch.cuda.Stream object per launch, ~1.5 ms of host time per training step)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """Device pointer as a plain int (ctypes converts it for c_void_p arguments and struct fields; None = NULL)."""
    return None if t is None else t.data_ptr()


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("oniris ops need HIP device tensors (there is no CPU fallback in the product path)")


def roundup(a, b):
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------------------------------------------------
# mask tables (host, int32) -- the reference's make_train_mask / make_infer_mask (attention_masking.py:27-90)

def train_mask_table(n_frames, image_size):
    """(kv_num_blocks (2nb,), kv_indices (2nb,2nb), block_size) as int32 numpy arrays, or None (reference: None)."""
    blk = ctypes.c_int(0)
    n = lib.oniris_train_mask(n_frames, image_size, None, None, ctypes.byref(blk))
    if n < 0:
        check(n, "train_mask")
    if n == 0:
        return None
    num = np.zeros(n, dtype=np.int32)
    idx = np.zeros((n, n), dtype=np.int32)
    lib.oniris_train_mask(n_frames, image_size, num.ctypes.data_as(ctypes.c_void_p),
                          idx.ctypes.data_as(ctypes.c_void_p), ctypes.byref(blk))
    return num, idx, blk.value


def infer_mask_table(n_frames, image_size):
    blk = ctypes.c_int(0)
    n = lib.oniris_infer_mask(n_frames, image_size, None, None, ctypes.byref(blk))
    if n < 0:
        check(n, "infer_mask")
    if n == 0:
        return None
    num = np.zeros(n, dtype=np.int32)
    idx = np.zeros((n, n), dtype=np.int32)
    lib.oniris_infer_mask(n_frames, image_size, num.ctypes.data_as(ctypes.c_void_p),
                          idx.ctypes.data_as(ctypes.c_void_p), ctypes.byref(blk))
    return num, idx, blk.value


def mask_transpose(num, idx):
    n_rows, n_cols = idx.shape
    qn = np.zeros(n_cols, dtype=np.int32)
    qi = np.zeros((n_cols, n_rows), dtype=np.int32)
    check(lib.oniris_mask_transpose(n_rows, n_cols, num.ctypes.data_as(ctypes.c_void_p),
                                    idx.ctypes.data_as(ctypes.c_void_p), qn.ctypes.data_as(ctypes.c_void_p),
                                    qi.ctypes.data_as(ctypes.c_void_p)), "mask_transpose")
    return qn, qi


_table_cache = {}


def device_tables(kind, n_frames, image_size, device):
    """Device copies of the (kv, q) tables for the kernels; cached per (kind, T, P, device)."""
    key = (kind, n_frames, image_size, str(device))
    if key not in _table_cache:
        tab = train_mask_table(n_frames, image_size) if kind == "train" else infer_mask_table(n_frames, image_size)
        if tab is None:
            _table_cache[key] = None
        else:
            num, idx, _ = tab
            qn, qi = mask_transpose(num, idx)
            _table_cache[key] = tuple(torch.from_numpy(a).to(device) for a in (num, idx, qn, qi)) + (tab[2],)
    return _table_cache[key]


def frame_tables(n_frames, image_size, device):
    """Block-diagonal (kv, q) tables: table block r (one frame of `image_size` = 128 * 2^k tokens) lists itself and nothing else.
    With mask_mode 1 (key frame <= query frame: everything inside a frame) the table-driven persistent kernels compute dense
    attention INSIDE every frame of a long pseudo-sequence -- FrameAttention (attention_modules.py:105-119) on the work lists of
    the VideoAttention kernels."""
    key = ("frame", n_frames, image_size, str(device))
    if key not in _table_cache:
        num = np.ones(n_frames, dtype=np.int32)
        idx = np.arange(n_frames, dtype=np.int32).reshape(n_frames, 1)
        _table_cache[key] = tuple(torch.from_numpy(a).to(device) for a in (num, idx, num.copy(), idx.copy())) + (image_size,)
    return _table_cache[key]


_sched_cache = {}
_cu_count = {}
ATTN_PERSISTENT = int(__import__("os").environ.get("ONIRIS_ATTN_PERSISTENT", "1"))   # 0: grid kernels (A/B knob)


_cu_reserve = 0


def set_cu_reserve(k):
    """CUs the persistent kernels leave free from the next launch on (oniris_set_cu_reserve: the LDS-DMA convolutions; here:
    the attention work lists, which are built per workgroup count).  parallel.OnirisDDP sets ONIRIS_COMM_CUS while a gradient
    exchange is in flight; ONIRIS_COMM_CUS_ALWAYS=k keeps k reserved for the whole process (one-GPU measurement of what the
    reservation costs the step).  Returns the previous value."""
    global _cu_reserve
    old = lib.oniris_set_cu_reserve(int(k))
    if old < 0:
        check(old, "set_cu_reserve")
    _cu_reserve = (int(k) + 7) // 8 * 8
    return old


if int(__import__("os").environ.get("ONIRIS_COMM_CUS_ALWAYS", "0")) > 0:
    set_cu_reserve(int(__import__("os").environ["ONIRIS_COMM_CUS_ALWAYS"]))


def set_ew_nt_bytes(nbytes):
    """Size from which tensors are streamed with non-temporal accesses (oniris_set_ew_nt_bytes; default 96 MiB / ONIRIS_EW_NT_MB;
    negative: never).  Returns the previous threshold.  The arithmetic does not depend on it; tests/ sets 0 so that the NT
    instantiations the B = 8 bench launches run on oracle-sized tensors."""
    return int(lib.oniris_set_ew_nt_bytes(int(nbytes)))


def census_start():
    """Start noting every kernel launch of the library (include/oniris.h: dispatch census)."""
    check(lib.oniris_census(1), "census")


def census_stop():
    """Stop the census; returns {'kernel instantiation[ [tag]]': launches}."""
    check(lib.oniris_census(0), "census")
    need = int(lib.oniris_census_read(None, 0))
    buf = ctypes.create_string_buffer(need)
    lib.oniris_census_read(buf, need)
    out = {}
    for line in buf.value.decode().splitlines():
        n, name = line.split("\t", 1)
        out[name] = out.get(name, 0) + int(n)
    return out


def attn_schedule(weights, n_pairs, device, n_wg=None):
    """Device copy of the static balanced schedule (oniris_attn_schedule) of n_pairs x len(weights) work items over
    the persistent workgroups (one per CU); cached per (weights, pairs, device).  Returns (tensor [n_wg][slots], n_wg,
    slots)."""
    if n_wg is None:
        n_wg = _cu_count.get(str(device))
        if n_wg is None:
            n_wg = _cu_count[str(device)] = torch.cuda.get_device_properties(device).multi_processor_count
        n_wg = max(8, n_wg - _cu_reserve)
    w = np.ascontiguousarray(weights, dtype=np.int32)
    key = (w.tobytes(), n_pairs, str(device), n_wg)
    hit = _sched_cache.get(key)
    if hit is None:
        need = lib.oniris_attn_schedule(n_pairs, len(w), w.ctypes.data_as(ctypes.c_void_p), n_wg, None, 0)
        if need < 0:
            check(need, "attn_schedule")
        tab = np.zeros((n_wg, need), dtype=np.int32)
        rc = lib.oniris_attn_schedule(n_pairs, len(w), w.ctypes.data_as(ctypes.c_void_p), n_wg,
                                      tab.ctypes.data_as(ctypes.c_void_p), need)
        if rc < 0:
            check(rc, "attn_schedule")
        hit = _sched_cache[key] = (torch.from_numpy(tab).to(device), n_wg, need)
    return hit


# ------------------------------------------------------------------------------------------------------------------
# weight bank: descriptor table + packed buffers for every NormalizedWeight of a model

class PackedWeight:
    """Handle of one weight inside a WeightBank (what the conv wrappers consume)."""
    __slots__ = ("param", "cout", "cin", "taps", "kt", "CoutP", "CinP", "CoutPb", "CinPb", "perm3", "gain",
                 "wf", "wb", "dwp", "dws", "bank", "nsplit_cap", "nsplit", "group", "goff", "members", "touched", "embcache", "gaincache",
                 "gview", "fresh", "hit")


def _nsplit_cap(cin, cout, taps, gated_pair=False):
    """Upper bound of the split-K columns (= slabs) the wgrad launchers use for a weight: launch_wgrad() in
    csrc/conv_wgrad.hip (64x64 or 32x32 output tiles) and launch_wgrad1x1_glds() (128x128 tiles, 1x1 weights).
    gated_pair: the weight is the own-frame or the context weight of an MPCausal3DGatedConv -- the only weights whose
    gradient launch_wgrad_stream() can take."""
    tile = 2 if (cin > 32 and cout > 32) else 1
    gy = -(-roundup(cin, 16) // (32 * tile)) * -(-roundup(cout, 8) // (32 * tile))
    cap = max(1, 512 // gy)       # (two 4-wave workgroups per CU for both tile forms since round 5: WGRAD_NG = 1 in csrc/conv_wgrad.hip)
    if tile == 1 and taps >= 9 and gated_pair:
        cap = 512       # launch_wgrad_stream() (csrc/conv_wgrad_stream.h): one slab per (sequence, 8x16-pixel tile, segment)
    if taps == 1 and cin >= 64 and cout >= 64:
        cap = max(cap, 256 // (-(-cin // 128) * -(-cout // 128)))
    return cap


class _ZeroArena:
    """Pre-zeroed fp32 scratch for the accumulators the backward kernels add into with atomics (gate / emb-scale
    gradient sums): ONE fill per step (WeightBank.prepare) instead of one `torch.zeros` launch per conv.

    The fill covers the WHOLE buffer on every reset, whatever the previous step took: the launch sequence of
    `prepare()` must not depend on host-side state, because it is captured into hipGraphs (a graph captured right
    after a 2-D step, which takes nothing, would otherwise contain no fill at all and every 3-D replay behind another
    3-D replay would add its gate / emb-scale gradient sums on top of the previous step's)."""

    MIN_ELEMS = 1 << 22                      # 16 MB: one ~4 us fill per step

    def __init__(self):
        self.buf, self.off, self.want = None, 0, self.MIN_ELEMS

    def _capturing(self):
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def take(self, n, device):
        n = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != device:
            self.buf, self.off = torch.zeros(self.want, dtype=torch.float32, device=device), 0
        if self.off + n > self.buf.numel():
            self.want = max(self.want, 2 * (self.off + n))      # grown at the next reset outside a capture
            self.off += n
            return torch.zeros(n, dtype=torch.float32, device=device)
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v

    def reset(self, device=None):
        if self.buf is not None and self.want > self.buf.numel() and not self._capturing():
            self.buf = None                                      # a step overflowed: take a larger arena
        if self.buf is None and device is not None and not self._capturing():
            self.buf = torch.empty(self.want, dtype=torch.float32, device=device)
        if self.buf is not None:
            self.buf.zero_()
        self.off = 0


_weights_epoch = 0          # bumped whenever a kernel writes parameters behind torch's back (weight_prep training, adamw_)


class WeightBank:
    def __init__(self):
        self.zero_arena = _ZeroArena()
        self.items = []
        self._dev_table = None
        self._sig = None
        self.total_rows = 0
        self._finish_queued = False
        self.post_backward_hooks = []      # callables run after the weight gradients are final (DDP all-reduce)
        self._flags, self._flag_off = None, 0
        self.groups = []                   # PackedWeight views over row-concatenated 1x1 weights (add_group)

    def add(self, param, perm3=False, gain=1.0, need_dgrad=True, gated_pair=False):
        w = PackedWeight()
        w.param = param
        shp = tuple(param.shape)
        w.cout, w.cin = shp[0], shp[1]
        w.taps = int(np.prod(shp[2:])) if len(shp) > 2 else 1
        w.kt = shp[2] if len(shp) == 5 else 1
        w.CoutP, w.CinP = roundup(w.cout, 32), roundup(w.cin, 64)
        w.CoutPb, w.CinPb = roundup(w.cin, 32), roundup(w.cout, 64)
        w.perm3, w.gain = bool(perm3), float(gain)
        w.wf = w.wb = w.dwp = w.dws = w.nsplit = None
        w.group, w.goff, w.members = None, 0, None
        w.touched = False              # a weight-gradient launch targeted this weight since the last optimizer step
        w.hit = False                  # ... in the running backward pass
        w.gview, w.fresh = None, False # released gradients: see WeightBank._reassign
        # split-K slabs of the weight-gradient kernel: same bound as launch_wgrad() in csrc/conv_wgrad.hip
        w.nsplit_cap = _nsplit_cap(w.cin, w.cout, w.taps, gated_pair)
        w.bank = self
        self.items.append((w, need_dgrad))
        return w

    def add_group(self, members):
        """Row-concatenate 1x1 weights that share their input (the emb_linear of every Block reads the same
        embedding, networks_edm2.py:78): ONE GEMM / dgrad / wgrad launch serves all of them.  `members` are
        PackedWeights of this bank with taps == 1 and equal cin; each keeps its own descriptor (normalisation and
        gradient are per weight) but its packed buffers become slices of the group's matrices: rows
        [off, off+cout) of wf [Ctot][CinP], columns of wb [CoutPb][Ctot], rows of every dwp slab -- which the
        descriptor expresses as CoutP = CinPb = Ctot plus a pointer offset.  Member k is padded to roundup(cout,64)
        rows (zero weights).  Returns the group PackedWeight; member.goff / member.gpad locate its columns."""
        assert members and all(m.taps == 1 and m.cin == members[0].cin and not m.perm3 for m in members)
        g = PackedWeight()
        g.members = list(members)
        off = 0
        for m in g.members:
            m.group, m.goff = g, off
            off += roundup(m.cout, 64)
        g.cout = g.CoutP = g.CinPb = off
        g.cin, g.taps, g.kt = members[0].cin, 1, 1
        g.CinP, g.CoutPb = roundup(g.cin, 64), roundup(g.cin, 32)
        g.perm3, g.gain, g.param, g.bank = False, 1.0, members[0].param, self
        g.wf = g.wb = g.dwp = g.dws = g.nsplit = None
        g.touched = g.hit = False
        g.nsplit_cap = _nsplit_cap(g.cin, g.cout, 1)
        self.groups.append(g)
        self._dev_table = None
        return g

    def _signature(self):
        return tuple((w.param.data_ptr(), w.param.grad.data_ptr() if w.param.grad is not None else 0)
                     for w, _ in self.items)

    def _build(self, device):
        self._packed_sig = None
        descs = (_lib.WeightDesc * len(self.items))()
        row = tile = 0
        nslots = len(self.items) + len(self.groups)
        if getattr(self, "nsplit_all", None) is None or self.nsplit_all.device != device or self.nsplit_all.numel() != nslots:
            self.nsplit_all = torch.zeros(nslots, dtype=torch.int32, device=device)
        for i, (w, _) in enumerate(self.items):
            w.nsplit = self.nsplit_all[i:i + 1]
        for j, g in enumerate(self.groups):                     # group buffers; members become slices of them
            k = len(self.items) + j
            g.nsplit = self.nsplit_all[k:k + 1]
            if g.wf is None or g.wf.device != device:
                g.wf = torch.zeros(g.cout * g.CinP, dtype=BF16, device=device)
                g.wb = torch.zeros(g.CoutPb * g.cout, dtype=BF16, device=device)
                g.dwp = torch.empty(g.nsplit_cap * g.cout * g.CinP, dtype=BF16, device=device)
                g.dws = torch.empty(g.cout * g.CinP, dtype=torch.float32, device=device)
            for m in g.members:
                m.nsplit, m.nsplit_cap = g.nsplit, g.nsplit_cap
                m.wf = g.wf[m.goff * g.CinP:]
                m.wb = g.wb[m.goff:]
                m.dwp = g.dwp[m.goff * g.CinP:]
                m.dws = g.dws[m.goff * g.CinP:]
        for i, (w, need_dgrad) in enumerate(self.items):
            p = w.param
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("weights must be contiguous fp32")
            grouped = getattr(w, "group", None) is not None
            if not grouped and (w.wf is None or w.wf.device != p.device):
                w.wf = torch.zeros(w.taps * w.CoutP * w.CinP, dtype=BF16, device=device)
                w.wb = torch.zeros(w.taps * w.CoutPb * w.CinPb, dtype=BF16, device=device) if need_dgrad else None
                # split-K slabs of the weight-gradient kernels: bf16 (the fp32 partial sum of a workgroup column is rounded
                # once; ~2 GB per step in fp32 for the gym net, written by the wgrad kernels and read back by weight_bwd)
                w.dwp = torch.empty(w.nsplit_cap * w.taps * w.CoutP * w.CinP, dtype=BF16, device=device)
                w.dws = torch.empty(w.taps * w.CoutP * w.CinP, dtype=torch.float32, device=device)
            d = descs[i]
            d.w = p.data_ptr()
            d.grad = p.grad.data_ptr() if p.grad is not None else None
            d.wf = w.wf.data_ptr()
            d.wb = w.wb.data_ptr() if w.wb is not None else None
            d.dwp = w.dwp.data_ptr()
            d.dws = w.dws.data_ptr()
            d.cout, d.cin, d.taps, d.kt = w.cout, w.cin, w.taps, w.kt
            d.CoutP, d.CinP, d.CoutPb, d.CinPb = w.CoutP, w.CinP, w.CoutPb, w.CinPb
            if grouped:                                         # slice of the group's matrices (see add_group)
                d.CoutP = d.CinPb = w.group.cout
            d.row_start, d.perm3, d.gain = row, int(w.perm3), w.gain
            d.nsplit_cap, d.nsplit = w.nsplit_cap, w.nsplit.data_ptr()
            d.tile_start = tile
            row += w.cout
            tile += -(-w.cout // 32)
        self.total_rows, self.total_tiles = row, tile
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        self._dev_table = torch.from_numpy(raw).to(device)
        self._sig = self._signature()

    def _reassign(self, released):
        """Gradients somebody released (`torch.optim.AdamW(...).zero_grad()` -- set_to_none=True is torch's default and what the
        reference's loops call, gym_train.py:72,108) get a zeroed tensor again, as autograd would create one for a parameter of
        its own.  They come out of ONE pooled buffer with a fixed slice per weight: a step of the reference loop then costs one
        fill instead of 184 allocations + fills, and the slices keep their addresses, so the descriptor table on the device
        stays valid (no rebuild + upload per step).  Handing a slice out again is only sound while nobody else still looks at
        it -- a loop that kept `g = p.grad` across zero_grad() expects `g` to stay what it was: the pool is used while its
        storage has no holder but the bank (checked: storage use count + reference counts of the slices), fresh tensors otherwise
        (a one-time RuntimeWarning says so; ONIRIS_GRAD_POOL=0 opts out of the pool altogether).
        A weight whose gradient was created here and that no weight-gradient launch targets in the backward pass that follows
        gets None back at the end of that pass (`_finish`): the reference leaves such parameters without a gradient (emb_time,
        networks_edm2.py:205-207) and torch.optim skips them."""
        trainable = [w for w, _ in self.items if w.param.requires_grad]
        dev = self.items[0][0].param.device
        pool = getattr(self, "_gpool", None)
        if pool is None or pool.device != dev or self._gpool_n != len(trainable):
            total = sum(w.param.numel() for w in trainable)
            pool = self._gpool = torch.zeros(total, dtype=torch.float32, device=dev)
            o = 0
            for w in trainable:
                n = w.param.numel()
                w.gview = pool[o:o + n].view_as(w.param)
                o += n
            self._gpool_n = len(trainable)
            self._gpool_uses = None
        use_count = getattr(torch._C, "_storage_Use_Count", None)       # (a private torch call: without it, no pooling)
        uses = use_count(pool.untyped_storage()._cdata) if use_count is not None else -1
        if self._gpool_uses is None:
            self._gpool_uses = uses
        pooled = (GRAD_POOL and uses >= 0 and uses == self._gpool_uses
                  and all(_sys.getrefcount(w.gview) == 2 for w in released))   # (2: the slot + the argument)
        if GRAD_POOL and not pooled and not getattr(self, "_gpool_warned", False):
            # both checks lean on interpreter / torch internals (CPython reference counts, a private storage use count): say so
            # ONCE when they switch the pool off -- fresh tensors are always correct, they cost a descriptor rebuild + upload per step
            self._gpool_warned = True
            import warnings
            warnings.warn("oniris: released weight gradients are handed out as fresh tensors instead of slices of the pooled buffer "
                          "(somebody else still holds a released gradient or the pool's storage, or torch's storage use count is "
                          "unavailable); correct, but every step rebuilds and uploads the weight descriptor table.  "
                          "ONIRIS_GRAD_POOL=0 switches the pool off for good and silences this.", RuntimeWarning, stacklevel=3)
        if pooled:
            if len(released) == len(trainable):
                pool.zero_()
            else:
                torch._foreach_zero_([w.gview for w in released])
        for w in released:
            w.param.grad = w.gview if pooled else torch.zeros_like(w.param)
            w.fresh = True

    def _ensure(self, assign=True):
        """assign=False (an evaluation under no_grad): released gradients stay released."""
        p0 = self.items[0][0].param
        _need_gpu(p0)
        sig, ok, released = [], self._dev_table is not None, None
        for w, _ in self.items:                        # ONE pass (this runs at the start of every step, on the critical
            p = w.param                                # path of the host: ~150 us for the 184 weights of the gym net)
            g = p.grad
            if g is None:
                if p.requires_grad and assign:
                    if released is None:
                        released = []
                    released.append(w)
                else:
                    sig.append((p.data_ptr(), 0))
                    continue
            sig.append((p.data_ptr(), g.data_ptr() if g is not None else -1))
        if released is not None:
            self._reassign(released)
            sig = self._signature()
        if not ok or self._sig != tuple(sig):
            self._build(p0.device)

    def prepare(self, training):
        """Forced weight normalisation (training: written back to the parameters) + bf16 packing; one launch."""
        self._ensure(assign=torch.is_grad_enabled())
        if training or torch.is_grad_enabled():        # (a no_grad evaluation -- the rollout -- never takes from the arena)
            self.zero_arena.reset(self.items[0][0].param.device)   # the previous step's backward is done with its accumulators
            self._flags = None                         # (clip flags: a fresh tensor per forward, see take_flag)
            GradSlot.live = []
        global _weights_epoch
        if training:
            _weights_epoch += 1                        # parameters are rewritten in place (forced normalisation)
        else:
            # eval (the sampler calls the net 31 times per generated frame): the packed copies stay valid until a
            # parameter changes -- through torch (._version) or through a raw-pointer kernel (_weights_epoch)
            sig = (_weights_epoch, tuple(w.param._version for w, _ in self.items))
            if getattr(self, "_packed_sig", None) == sig:
                return
        check(lib.oniris_weight_prep(_p(self._dev_table), len(self.items), self.total_rows, self.total_tiles, int(training),
                                     _stream()),
              "weight_prep")
        self._packed_sig = None if training else (_weights_epoch, tuple(w.param._version for w, _ in self.items))

    def take_flag(self, device):
        """One zeroed int32 for a forward launch to report into and the matching backward to read (OnirisConvArgs.clip_flag).
        Its storage lives as long as the autograd graph that holds it (the ctx keeps the slice): every grad-enabled forward
        starts a fresh tensor (`prepare`), so a second forward before the first one's backward -- two micro-batches summed into
        one loss, a train-mode evaluation in between -- neither clears nor aliases the first one's flags (the
        step's zero arena, which they came from before, is rewound and refilled by every forward)."""
        if self._flags is None or self._flags.device != device or self._flag_off >= self._flags.numel():
            self._flags, self._flag_off = torch.zeros(256, dtype=torch.int32, device=device), 0
        f = self._flags[self._flag_off:self._flag_off + 1]
        self._flag_off += 1
        return f

    def packed_valid(self):
        """True while the packed EVAL weights of the last `prepare(False)` still match the parameters."""
        sig = getattr(self, "_packed_sig", None)
        return sig is not None and sig == (_weights_epoch, tuple(w.param._version for w, _ in self.items))

    def backward(self):
        """Packed fp32 weight gradients (from the wgrad kernels) -> parameter .grad (accumulated); one launch."""
        # re-validated on every call (~150 us of host time): between forward and backward the gradients may have been
        # released -- `loss = model(x); opt.zero_grad(); loss.backward()` with torch's set_to_none=True -- and the table
        # would still hold the freed .grad pointers (the kernel would write into recycled allocator memory)
        self._ensure()
        check(lib.oniris_weight_bwd(_p(self._dev_table), len(self.items), self.total_rows, _stream()), "weight_bwd")
        self.nsplit_all.zero_()        # the slabs are consumed: a second backward() must not add them again

    def request_finish(self):
        """Called from inside a conv backward: run `backward()` once, after the autograd engine has finished the
        whole pass (all wgrad kernels enqueued), then the post-backward hooks."""
        if not self._finish_queued:
            self._finish_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)

    def _finish(self):
        if not self._finish_queued:        # already finished early (e.g. by the DDP end-of-backward callback)
            return
        self._finish_queued = False
        self.backward()
        for w, _ in self.items:            # (gradients the bank created for this pass and nothing was added to: _reassign)
            if w.fresh:
                if not w.hit:
                    w.param.grad = None
                w.fresh = False
            w.hit = False
        for h in self.post_backward_hooks:
            h()
        GradSlot.check_all_taken()


# ------------------------------------------------------------------------------------------------------------------
# convolution

import os as _os
import sys as _sys
FUSED_ROPE = int(_os.environ.get("ONIRIS_FUSED_ROPE", "1"))        # 0: qkv normalisation and the two rotations as three launches (A/B, tests)
ATTN_DKV_CHUNKS = int(_os.environ.get("ONIRIS_DKV_CHUNKS", "4"))   # dK/dV query-list chunks (OnirisAttnArgs.dkv_chunks)
ATTN_DKV_MIN_L = 2048                                              # ... one chunk per this many tokens at most
ATTN_DKV_PERSISTENT = int(_os.environ.get("ONIRIS_DKV_PERSISTENT", "1"))   # 0: grid dK/dV kernel + chunk reduction (A/B, tests)
ATTN_DQ_PERSISTENT = int(_os.environ.get("ONIRIS_DQ_PERSISTENT", "1"))      # 0: VideoAttention dQ through the grid kernel
WGRAD_VARIANT = int(_os.environ.get("ONIRIS_WGRAD", "0"))   # < 0: register-staged wgrad kernel everywhere (A/B knob)
BIG_TILE = int(_os.environ.get("ONIRIS_BIG_TILE", "4"))     # conv tuning knob (see OnirisConvArgs.big_tile)
GRAD_POOL = int(_os.environ.get("ONIRIS_GRAD_POOL", "1"))    # 0: gradients released by a foreign zero_grad() always come back as fresh tensors (WeightBank._reassign)
GRAD_SLOTS = int(_os.environ.get("ONIRIS_GRAD_SLOTS", "1"))  # 0: skip / residual gradients are joined by autograd (A/B, partial backward)


class KernelProfile:
    """Optional per-launch HIP-event timing of the MFMA kernels (bench.py's roofline leg), on the stream the kernel is
    launched on (torch's current stream): see _timed_launch."""
    enabled = False
    records = []          # (key, algorithmic flops, start_event, end_event[, algorithmic HBM bytes])

    @classmethod
    def start(cls):
        cls.records, cls.enabled = [], True

    @classmethod
    def stop(cls):
        cls.enabled = False
        torch.cuda.synchronize()
        agg = {}
        for key, flops, e0, e1, *rest in cls.records:
            a = agg.setdefault(key, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0, t_min=0.0))
            a["launches"] += 1
            a["flops"] += flops
            a["ms"] += e0.elapsed_time(e1)
            a["bytes"] += rest[0] if rest else 0.0
            # roofline time of the launch (seconds): SURVEY 8d's max(FLOPs / bf16 MFMA peak, algorithmic bytes / 6.3 TB/s)
            a["t_min"] += max(flops / 2.5e15, (rest[0] if rest else 0.0) / 6.3e12)
        cls.records = []
        return agg


def _timed_launch(fn):
    """fn() = ONE kernel launch on the current stream; returns (start, end) HIP events.  The pair is armed in the library
    (oniris_profile_arm): launch sites with the hook record the dispatch's own begin / end into it (what rocprofv3 reports
    as the kernel's duration); entry points without the hook leave it armed and get events recorded around the call."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    e1.record()                                    # (creates the handles; both are recorded again below)
    check(lib.oniris_profile_arm(e0.cuda_event, e1.cuda_event), "profile_arm")
    try:
        fn()
    finally:
        if lib.oniris_profile_disarm():
            e1.record()
    return e0, e1


def _profiled(key, flops, fn, nbytes=None):
    """Run fn() (a kernel launch on the current stream); when KernelProfile is on, time it with HIP events.  nbytes: the launch's
    algorithmic HBM bytes where its roofline is the memory system's (t_min = max(FLOPs / MFMA peak, bytes / 6.3 TB/s))."""
    if not KernelProfile.enabled:
        return fn()
    e0, e1 = _timed_launch(fn)
    KernelProfile.records.append((key, flops, e0, e1) if nbytes is None else (key, flops, e0, e1, float(nbytes)))


def train_frame_pairs(T, P):
    """Unmasked (query frame, key frame) pairs of the DART training mask = table AND mask_mod (reference
    attention_masking.py:32-53): clean rows T(T+1)/2, noised row f sees clean frames < fpb*floor(f/fpb) and itself,
    fpb = max(1, 128 // P) frames per 128-token mask block (4128 at T=64, P=64; 944 at T=32, P=16)."""
    fpb = max(1, 128 // P)
    return T * (T + 1) // 2 + sum(1 + fpb * (f // fpb) for f in range(T))


def _attn_flops(kind, B, T, heads, L, P):
    """Algorithmic FLOPs of ONE product pair (QK^T + PV) over the unmasked token pairs: video training mask
    train_frame_pairs(T, P) x P^2, dense per-frame attention L^2 per frame."""
    pairs = (train_frame_pairs(T, P) * P * P * B) if kind == "video" else (B * L * L)
    return 4.0 * 64 * heads * pairs


def _patch_w(W):
    return 16 if W >= 16 else W


def _s2ctx_family(T, H, W, Cin, CinP, CoutP, ctx_T, coff, ctx_fill):
    """Which kernel family conv_dispatch_s2ctx() (csrc/conv_fwd_s2ctx.hip) picks for a gated conv in the DART training layout:
    'stream' (conv_stream.h), 'glds16' / 'glds8' (conv_glds.h) or 'staged' (conv_kernels.h).  Mirrors the C side."""
    ok = ctx_fill in (0.0, 1.0)
    if (BIG_TILE >= 4 and ok and Cin == 32 and CinP == 64 and CoutP == 32 and W % 16 == 0 and H % 4 == 0 and ctx_T == T
            and tuple(coff) in ((-2, -1), (2, 1))):
        return "stream"
    if BIG_TILE >= 3 and ok and Cin % 32 == 0 and H % 16 == 0 and W % 16 == 0:
        return "glds16"
    if BIG_TILE >= 3 and ok and Cin % 32 == 0 and H == 8 and W == 8 and CoutP % 64 == 0:
        return "glds8"
    return "staged"


PROFILE_SHAPES = int(_os.environ.get("ONIRIS_PROFILE_SHAPES", "0"))   # KernelProfile keys of the conv launches carry their shape (scratch/r05_conv_shapes.py)
ALIAS2 = int(_os.environ.get("ONIRIS_ALIAS2", "1"))          # 0: round-4 extent of the aliasing protocol only (A/B: no residual alias, no plain-conv alias)
CLIP_FLAG = int(_os.environ.get("ONIRIS_CLIP_FLAG", "1"))    # 0: the mp_sum backward always reads the clipped output and writes a masked gradient copy (A/B, tests)


def _conv_launch(x, ctx, w_own, w_ctx, out, coef_own, coef_ctx, B, S, T, H, W, Cin, CinP, Cout, CoutP, taps,
                 ctx_bstride=0, ctx_T=0, coff=(0, 0), ctx_fill=0.0, epi=0, res=None, escale=None, emb_gain=None,
                 out2=None, ta=0.0, tb=0.0, clip=0.0, ctx_out=None, clip_flag=None, ctx_prod=None, ctx_prod_mode=0,
                 x2=None, act_out=None, x_split=0, cat_w=(1.0, 1.0)):
    if KernelProfile.enabled:
        flops = 2.0 * B * S * T * H * W * Cout * Cin * taps
        if ctx is not None:
            flops += 2.0 * B * T * H * W * Cout * Cin * 2 * taps
        nt = 2 if CoutP % 64 == 0 else 1
        if (BIG_TILE >= 4 and S == 2 and ctx is not None and taps == 9 and Cin == 32 and CinP == 64 and CoutP == 32 and W % 16 == 0
                and H % 4 == 0 and ctx_fill in (0.0, 1.0) and ctx_T == T and tuple(coff) in ((-2, -1), (2, 1))):
            key = f"conv_stream_kernel<ALIAS={int(ctx.data_ptr() == x.data_ptr() and ctx_bstride == 2 * T and coff[0] < 0)}>"   # conv_stream_ok()
        elif (BIG_TILE >= 3 and S == 2 and ctx is not None and taps == 9 and Cin % 32 == 0 and H % 16 == 0 and W % 16 == 0
                and ctx_fill in (0.0, 1.0)):          # mirrors conv_glds_ok() in csrc/conv_glds.h
            key = f"conv_glds_kernel<NT={nt},PW=16,NW=8,MT=1,WC=1,CTX=1>"
        elif (BIG_TILE >= 3 and S == 2 and ctx is not None and taps == 9 and Cin % 32 == 0 and H == 8 and W == 8
              and CoutP % 64 == 0 and ctx_fill in (0.0, 1.0)):
            key = "conv_glds_kernel<NT=1,PW=8,NW=8,MT=1,WC=2,CTX=1>"
        elif (BIG_TILE >= 4 and not (BIG_TILE & 128) and S == 1 and ctx is None and taps == 9 and Cin == 32 and CinP == 64 and CoutP == 32
              and W % 16 == 0 and H % 8 == 0 and x2 is None and B * T * (H // 8) * (W // 16) >= 512):   # conv_plain_stream_ok()
            key = "conv_plain_stream_kernel"
        elif (BIG_TILE >= 3 and S == 1 and ctx is None and taps == 9 and Cin % 32 == 0 and T % 2 == 0
              and ((H % 16 == 0 and W % 16 == 0) or (H == 8 and W == 8 and CoutP % 64 == 0))):   # conv_dispatch_s1()
            key = (f"conv_glds_kernel<NT={nt},PW=16,NW=8,MT=1,WC=1,CTX=0>" if H % 16 == 0 else
                   "conv_glds_kernel<NT=1,PW=8,NW=8,MT=1,WC=2,CTX=0>")
        elif (BIG_TILE >= 3 and taps == 1 and ctx is None and Cin % 64 == 0 and CinP == Cin and epi in (0, _lib.EPI_MPSUM)
              and B * S * T * H * W >= 8192):                 # conv1x1_glds_ok() in csrc/conv1x1_glds.h
            key = "conv1x1_glds_kernel"
        else:
            key = f"conv_fwd_kernel<S={S},TAPS={taps},CK={32 if taps == 9 else 64},NT={nt},CTX={int(ctx is not None)},PW={_patch_w(W)}>"
        if PROFILE_SHAPES:
            key += f" [{H}x{W} {Cin}->{Cout} epi={epi}{' res' if res is not None else ''}{' out2' if out2 is not None else ''}{' y3' if ctx_out is not None else ''}{' dgrad' if coff[0] > 0 else ''}]"
        KernelProfile.enabled = False
        try:
            e0, e1 = _timed_launch(lambda: _conv_launch(x, ctx, w_own, w_ctx, out, coef_own, coef_ctx, B, S, T, H, W, Cin, CinP,
                                                        Cout, CoutP, taps, ctx_bstride, ctx_T, coff, ctx_fill, epi, res, escale,
                                                        emb_gain, out2, ta, tb, clip, ctx_out, clip_flag, ctx_prod, ctx_prod_mode))
        finally:
            KernelProfile.enabled = True
        # algorithmic HBM bytes of the launch (every operand read once, every result written once; the context
        # frames of a training launch are frames of x itself and are not counted twice)
        px = B * T * H * W
        nbytes = 2.0 * (S * px * Cin + S * px * Cout * (1 + (out2 is not None) + (res is not None))
                        + (px * Cout if ctx_out is not None else 0)
                        + taps * CoutP * CinP * (3 if ctx is not None else 1))
        if ctx is not None and ctx.data_ptr() != x.data_ptr():
            nbytes += 2.0 * B * ctx_T * H * W * Cin
        KernelProfile.records.append((key, flops, e0, e1, nbytes))
        return
    a = _lib.ConvArgs()
    a.x, a.ctx, a.w_own, a.w_ctx, a.out = _p(x), _p(ctx), _p(w_own), _p(w_ctx), _p(out)
    a.coef_own, a.coef_ctx = _p(coef_own), _p(coef_ctx)
    a.B, a.S, a.T, a.H, a.W = B, S, T, H, W
    a.Cin, a.CinP, a.Cout, a.CoutP, a.taps = Cin, CinP, Cout, CoutP, taps
    a.ctx_bstride, a.ctx_T, a.coff0, a.coff1, a.ctx_fill = ctx_bstride, ctx_T, coff[0], coff[1], ctx_fill
    a.epi, a.res, a.escale, a.emb_gain, a.out2 = epi, _p(res), _p(escale), _p(emb_gain), _p(out2)
    if escale is not None:
        a.escale_pitch = escale.stride(0)
    a.ta, a.tb, a.clip, a.ctx_out = ta, tb, clip, _p(ctx_out)
    a.clip_flag = _p(clip_flag)
    a.ctx_prod, a.ctx_prod_mode = _p(ctx_prod), ctx_prod_mode
    if x2 is not None:
        a.x2, a.act_out, a.x_split, a.cat_w1, a.cat_w2 = _p(x2), _p(act_out), x_split, cat_w[0], cat_w[1]
    a.big_tile = BIG_TILE
    if SPLITK and B * S * T * H * W <= 64 * 256 and ctx_prod_mode == 0:      # few tiles (one rollout frame): lend the split-K workspace
        ws = _splitk_workspace(x.device)
        a.splitk_ws, a.splitk_ws_bytes = _p(ws), ws.numel() * 4
    check(lib.oniris_conv_fwd(ctypes.byref(a), _stream()), "conv_fwd")


SPLITK = int(_os.environ.get("ONIRIS_SPLITK", "1"))
_splitk_cache = {}


def _splitk_workspace(device):
    """fp32 partial-sum workspace of the split-K conv path (OnirisConvArgs.splitk_ws); one per device: the two
    launches of a conv and consecutive convs are ordered on the stream."""
    key = str(device)
    if key not in _splitk_cache:
        _splitk_cache[key] = torch.empty(8 << 20, dtype=torch.float32, device=device)        # 32 MB
    return _splitk_cache[key]


def _wgrad_args(x, dy, pw, scale, B, T, H, W, Cin, CinP, Cout, CoutP, taps, xb_stride, x_T, coff, fill, tap0=0):
    """pw: PackedWeight whose slabs receive the partial sums (taps [tap0, tap0+taps) of its pw.taps-deep slabs)."""
    a = _lib.WgradArgs()
    pw.touched = pw.hit = True
    if pw.members is not None:
        for m_ in pw.members:
            m_.touched = m_.hit = True
    a.x, a.dy, a.dwp, a.scale = _p(x), _p(dy), _p(pw.dwp), _p(scale)
    a.nsplit_cap, a.taps_total, a.tap0, a.nsplit_out = pw.nsplit_cap, pw.taps, tap0, _p(pw.nsplit)
    a.B, a.T, a.H, a.W, a.Cin, a.CinP, a.Cout, a.CoutP, a.taps = B, T, H, W, Cin, CinP, Cout, CoutP, taps
    a.xb_stride, a.x_T, a.coff, a.fill = xb_stride, x_T, coff, fill
    a.pad_ = WGRAD_VARIANT
    return a


def _wgrad_launch_group(arglist):
    """One launch for 1..3 weight-gradient problems of the same geometry (oniris_conv_wgrad_group)."""
    if KernelProfile.enabled:
        a0 = arglist[0]
        tile = 2 if (a0.Cin > 32 and a0.Cout > 32) else 1
        if (WGRAD_VARIANT >= 0 and not (WGRAD_VARIANT & 2) and len(arglist) == 3 and tile == 1 and a0.taps == 9 and a0.W % 16 == 0
                and a0.H % 8 == 0 and arglist[1].B * (a0.W // 16) * (a0.H // 8) <= min(a0.nsplit_cap, arglist[1].nsplit_cap)):
            key = "conv_wgrad_stream_kernel"                                      # mirrors wgrad_stream_ok() / launch_wgrad_stream()
        elif WGRAD_VARIANT >= 0 and all(a.taps == 9 and ((a.W % 16 == 0 and a.H % 8 == 0) or (a.W == 8 and a.H == 8))
                                      and a.fill in (0.0, 1.0) for a in arglist):     # mirrors wgrad_glds_ok() in csrc
            key = f"conv_wgrad_glds_kernel<CT={tile},IT={tile},NG=1,PW={_patch_w(a0.W)}>"
        elif (WGRAD_VARIANT >= 0 and len(arglist) == 1 and a0.taps == 1 and a0.Cin >= 64 and a0.Cout >= 64 and not a0.scale
              and a0.coff == 0):                                               # mirrors wgrad1x1_glds_ok()
            key = "wgrad1x1_glds_kernel<NG=2>"
        else:
            key = f"conv_wgrad_kernel<TAPS={a0.taps},PW={_patch_w(a0.W)},CT={tile},IT={tile}>"
        flops = sum(2.0 * a.B * a.T * a.H * a.W * a.Cout * a.Cin * a.taps for a in arglist)
        # algorithmic HBM bytes: every distinct operand read once (the context groups of a gated conv read frames of the own
        # group's x again: counted once), + one slab set written
        seen, nbytes = set(), 0.0
        for a in arglist:
            for ptr, n in ((a.x, a.B * a.xb_stride * a.H * a.W * a.Cin), (a.dy, a.B * a.T * a.H * a.W * a.Cout)):
                if ptr not in seen:
                    seen.add(ptr)
                    nbytes += 2.0 * n
        KernelProfile.enabled = False
        try:
            e0, e1 = _timed_launch(lambda: _wgrad_launch_group(arglist))
        finally:
            KernelProfile.enabled = True
        KernelProfile.records.append((key, flops, e0, e1, nbytes))
        return
    arr = (_lib.WgradArgs * len(arglist))(*arglist)
    check(lib.oniris_conv_wgrad_group(arr, len(arglist), _stream()), "conv_wgrad")


def _wgrad_launch(x, dy, pw, scale, B, T, H, W, Cin, CinP, Cout, CoutP, taps, xb_stride, x_T, coff, fill, tap0=0):
    _wgrad_launch_group([_wgrad_args(x, dy, pw, scale, B, T, H, W, Cin, CinP, Cout, CoutP, taps, xb_stride, x_T, coff,
                                     fill, tap0)])


class ConvCfg:
    """Static configuration of one conv op (not a tensor: passed through autograd untouched)."""
    __slots__ = ("pw2", "pw3", "B", "T", "epi", "ta", "tb", "clip", "need_grad", "in_slot", "res_slot", "grad_private", "res_alias")

    def __init__(self, pw2, pw3=None, B=1, T=1, epi="none", ta=0.0, tb=0.0, clip=0.0, need_grad=True, in_slot=None,
                 res_slot=None, grad_private=False, res_alias=False):
        self.pw2, self.pw3, self.B, self.T = pw2, pw3, B, T
        # the caller vouches that the gradient of this op's output is a tensor nobody else reads (UNet.forward with GradSlots:
        # it comes out of the ONE backward kernel that joined the gradients of all consumers): the backward may then mask it
        # in place instead of writing a masked copy (the clip_flag aliasing protocol below).  Anywhere else -- y.backward(g)
        # with the caller's g, torch.autograd.grad(grad_outputs=...), a consumer whose backward hands one tensor to two
        # inputs -- the gradient is read-only.
        self.grad_private = bool(grad_private)
        # res_slot's taker applies a scale itself (GradSlot.take_scaled: the activation whose output `res` is): under the
        # aliasing protocol the residual gradient ta * g is then not written at all
        self.res_alias = bool(res_alias) and res_slot is not None
        self.epi, self.ta, self.tb, self.clip, self.need_grad = epi, ta, tb, clip, need_grad
        self.in_slot = in_slot             # GradSlot of the input (plain convs): a second gradient of x joins in the dgrad epilogue
        self.res_slot = res_slot           # GradSlot that receives the gradient of `res` (mp_sum epilogue) instead of autograd


def _rows_f32(t):
    """(N, C) fp32 rows the kernels can read in place: unit stride inside a row, 16-byte aligned rows (a column block of
    the UNet's one emb-scale matrix qualifies); anything else is copied."""
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4 != 0 or t.stride(0) < t.shape[1] or t.data_ptr() % 16 != 0:
        t = t.contiguous()
    return t


def _wants_wgrad(pw):
    """Does a weight-gradient launch have anybody to serve?  A row-concatenated group (WeightBank.add_group) does as soon as ONE
    member is trainable: weight_bwd_kernel finishes every member on its own and leaves frozen ones (no .grad) alone."""
    members = getattr(pw, "members", None)
    if members:
        return any(m.param.requires_grad for m in members)
    return pw.param.requires_grad


class _ConvOp(torch.autograd.Function):
    """One fused conv op on channels-last bf16 activations.

    plain  (cfg.pw3 None):  v = conv(x, W2)                                      x (N,H,W,Cin)
    gated  (training DART layout, slot order (b s t), N = B*2*T; edm2/conv.py:59-95):
                            v = ca[n]*conv2d(x[n]) + cb[n]*(conv(clean[t-2]) + conv(clean[t-1])), ones-padded in time
    epilogue 'none': returns v; 'emb_silu': returns silu(v*cscale[n,co])/0.596 (v kept for the backward);
             'mpsum': returns clip(ta*res + tb*v).
    The weight Parameters are inputs only so that autograd always schedules the node; their gradients are
    accumulated into the bank's packed fp32 buffers (WeightBank.backward() finishes them)."""

    @staticmethod
    def forward(ctx, x, w2param, w3param, ca, cb, cscale, res, cfg):
        _need_gpu(x)
        pw2, pw3 = cfg.pw2, cfg.pw3
        gated = pw3 is not None
        N, H, W, Cin = x.shape
        Co = roundup(pw2.cout, 8)                       # channel counts in HBM are multiples of 8 (16-byte vectors)
        dev = x.device
        kw = {}
        raw = torch.empty((N, H, W, Co), dtype=BF16, device=dev)
        ret = raw
        clip_flag = None
        if cfg.epi == "emb_silu":
            cs = _rows_f32(cscale)
            assert tuple(cs.shape) == (N, Co), (cs.shape, N, Co)
            ret = torch.empty_like(raw)
            kw = dict(epi=_lib.EPI_EMB_SILU, escale=cs, out2=ret)
        elif cfg.epi == "mpsum":
            keep_raw = gated and cfg.need_grad
            ret = torch.empty_like(raw)
            kw = dict(epi=_lib.EPI_MPSUM, res=res, ta=cfg.ta, tb=cfg.tb, clip=cfg.clip, out2=raw if keep_raw else None)
            if not keep_raw:
                raw = None
            # "did the clip change anything?" -- one int of the step's pre-zeroed arena, set by the forward launch (the kernel
            # families that support it), read by the backward pre-pass: see OnirisConvArgs.clip_flag
            # (the family test only saves arming a flag the register-staged kernels would answer with "assume clipped":
            # conv_dispatch_s2ctx sets the flag itself when it falls back to them)
            if (CLIP_FLAG and keep_raw and cfg.grad_private and Co <= 512 and cfg.clip > 0 and
                    _s2ctx_family(cfg.T, H, W, Cin, pw2.CinP, pw2.CoutP, cfg.T, (-2, -1), 1.0) != "staged"):
                clip_flag = pw2.bank.take_flag(dev)
                kw["clip_flag"] = clip_flag
            # plain 3x3 conv (conv_res1 of the 2-D steps): the same protocol, with oniris_mpsum_mask as the backward's (almost
            # always empty) masking pass
            if (CLIP_FLAG and ALIAS2 and not gated and cfg.need_grad and cfg.grad_private and pw2.taps == 9 and cfg.clip > 0 and
                    Cin % 32 == 0 and (H % 16 == 0 and W % 16 == 0 or H == 8 and W == 8 and pw2.CoutP % 64 == 0) and N % 2 == 0):
                clip_flag = pw2.bank.take_flag(dev)
                kw["clip_flag"] = clip_flag
        y3 = ca32 = cb32 = None
        ctx.clip_flag = kw.get("clip_flag")
        first_out = ret if cfg.epi == "mpsum" else (raw if raw is not None else ret)
        if gated:
            B, T = cfg.B, cfg.T
            assert N == B * 2 * T
            y3 = torch.empty((B * T, H, W, Co), dtype=BF16, device=dev) if cfg.need_grad else None
            ca32, cb32 = ca.detach().float().contiguous(), cb.detach().float().contiguous()
            _conv_launch(x, x, pw2.wf, pw3.wf, first_out, ca32, cb32, B, 2, T, H, W, Cin, pw2.CinP, Co, pw2.CoutP, 9,
                         ctx_bstride=2 * T, ctx_T=T, coff=(-2, -1), ctx_fill=1.0, ctx_out=y3, **kw)
        else:
            _conv_launch(x, None, pw2.wf, None, first_out, None, None, 1, 1, N, H, W, Cin, pw2.CinP, Co, pw2.CoutP,
                         pw2.taps, **kw)
        ctx.cfg = cfg
        ctx.save_for_backward(x, raw, y3, ca32, cb32, kw["escale"] if cfg.epi == "emb_silu" else None,
                              ret if (cfg.epi == "mpsum" and cfg.clip > 0) else None)
        return ret

    @staticmethod
    def backward(ctx, g):
        x, raw, y3, ca, cb, cs, xo = ctx.saved_tensors
        cfg = ctx.cfg
        pw2, pw3 = cfg.pw2, cfg.pw3
        gated = pw3 is not None
        N, H, W, Cin = x.shape
        g = g.contiguous()
        Co = g.shape[-1]
        dev = x.device
        dcs = dres = None
        dx = dca = dcb = None
        plain_coef = None                           # per-frame coefficient of `dout` for a plain conv's dgrad / wgrad (aliasing)
        fused = gated and cfg.epi in ("emb_silu", "mpsum") and Co <= 512
        ca_own = ca                                 # the own-frame coefficient dgrad / wgrad apply to `dout`
        if fused:                                   # epilogue adjoint + gate/context pre-pass in ONE pass over g
            B, T = cfg.B, cfg.T
            flag = getattr(ctx, "clip_flag", None) if cfg.epi == "mpsum" else None
            # (clip <= 0 -- conv_res1 in front of an attention layer, which clips later: nothing to mask, no flag needed)
            alias = cfg.epi == "mpsum" and cfg.grad_private and CLIP_FLAG and (flag is not None or (cfg.clip <= 0 and ALIAS2))
            if alias:
                # aliasing protocol (include/oniris.h): dout = tb * g * mask is not written -- dgrad and wgrad read g with
                # the coefficient vector tb * ca; the mask exists (and is applied to g in place) only if the forward clipped
                # (g is the gradient of this conv's output alone -- produced by the one kernel that joined all of its
                # consumers' gradients -- and nothing reads it after this backward)
                dout = g
                ca_own = torch.empty_like(ca)
            else:
                dout = torch.empty_like(g)
            dy3 = torch.empty_like(y3)
            acc = pw2.bank.zero_arena.take(N * (2 + (Co if cfg.epi == "emb_silu" else 0)), dev)
            dca, dcb = acc[:N], acc[N:2 * N]
            res_alias = alias and cfg.res_alias            # the residual's consumer reads g itself, times ta
            if cfg.epi == "emb_silu":
                dcs = acc[2 * N:2 * N + N * Co].view(N, Co)
            elif not res_alias:
                dres = torch.empty_like(g)
            check(lib.oniris_gconv_bwd_fused(1 if cfg.epi == "emb_silu" else 2, _p(g), _p(raw), _p(y3), _p(ca), _p(cb),
                                             _p(cs), _p(xo), _p(dout), _p(dres), _p(dy3), _p(dca), _p(dcb), _p(dcs), B, T,
                                             H * W, Co, cfg.ta, cfg.tb, cfg.clip, cs.stride(0) if cs is not None else 0,
                                             _p(flag) if alias else None, _p(ca_own) if alias else None, _stream()), "gconv_bwd_fused")
            if res_alias:
                cfg.res_slot.put(g, cfg.ta)
                cfg = _no_res_slot(cfg)
        elif cfg.epi == "emb_silu":
            dout = torch.empty_like(g)
            dcs = pw2.bank.zero_arena.take(N * Co, dev)[:N * Co].view(N, Co)     # (zero-filled once per step with all the others)
            check(lib.oniris_emb_silu_bwd(_p(g), _p(raw), _p(cs), _p(dout), _p(dcs), N, H * W, Co, cs.stride(0), 1, _stream()),
                  "emb_silu_bwd")
        elif cfg.epi == "mpsum" and not gated and cfg.grad_private and CLIP_FLAG and ALIAS2 and pw2.taps == 9 and (
                getattr(ctx, "clip_flag", None) is not None or cfg.clip <= 0):
            # aliasing protocol, plain 3x3 conv: dv = tb * g * mask and dres = ta * g * mask are scaled copies of g -- dgrad and
            # wgrad read g with the per-frame coefficient tb, the residual's consumer with the scale ta; the mask exists (and is
            # applied to g in place) only if the forward launch reported a clip
            flag = getattr(ctx, "clip_flag", None)
            if flag is not None:
                check(lib.oniris_mpsum_mask(_p(g), _p(xo), g.numel(), cfg.clip, _p(flag), _stream()), "mpsum_mask")
            dout, plain_coef = g, _const_vec(N, cfg.tb, dev)
            if cfg.res_alias:
                cfg.res_slot.put(g, cfg.ta)
                cfg = _no_res_slot(cfg)
            else:
                dres = torch.empty_like(g)
                check(lib.oniris_mpsum_bwd(_p(g), None, _p(dres), None, g.numel(), cfg.ta, cfg.tb, 0.0, _stream()), "mpsum_bwd")
        elif cfg.epi == "mpsum":
            dres, dout = torch.empty_like(g), torch.empty_like(g)
            check(lib.oniris_mpsum_bwd(_p(g), _p(xo), _p(dres), _p(dout), g.numel(), cfg.ta, cfg.tb, cfg.clip, _stream()),
                  "mpsum_bwd")
        else:
            dout = g
        if gated:
            B, T = cfg.B, cfg.T
            if not fused:
                acc = pw2.bank.zero_arena.take(2 * N, dev)           # (accumulated by the launch's pixel slices)
                dca, dcb = acc[:N], acc[N:2 * N]
                dy3 = torch.empty_like(y3)
                check(lib.oniris_gconv_bwd_prep(_p(dout), _p(raw), _p(y3), _p(ca), _p(cb), _p(dca), _p(dcb), _p(dy3), B, 2,
                                                T, H * W * Co, _stream()), "gconv_bwd_prep")
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                sel = _clean_selector(B, T, dev)               # the context gradient only reaches the clean slot
                _conv_launch(dout, dy3, pw2.wb, pw3.wb, dx, ca_own, sel, B, 2, T, H, W, Co, pw2.CinPb, Cin, pw2.CoutPb, 9,
                             ctx_bstride=T, ctx_T=T, coff=(2, 1), ctx_fill=0.0)
            grp = []                                 # own-frame weight + the two context taps: ONE split-K launch
            if _wants_wgrad(pw2):
                grp.append(_wgrad_args(x, dout, pw2, ca_own, 1, N, H, W, Cin, pw2.CinP, Co, pw2.CoutP, 9, N, N, 0, 0.0))
            if _wants_wgrad(pw3):
                for j, coff in enumerate((-2, -1)):
                    grp.append(_wgrad_args(x, dy3, pw3, None, B, T, H, W, Cin, pw3.CinP, Co, pw3.CoutP, 9, 2 * T, T, coff,
                                           1.0, tap0=9 * j))
            if grp:
                _wgrad_launch_group(grp)
        else:
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                dadd = cfg.in_slot.take() if cfg.in_slot is not None else None       # (see GradSlot)
                if dadd is not None:             # dx = dadd + dgrad: the mp_sum epilogue with ta = tb = 1, no clip
                    _conv_launch(dout, None, pw2.wb, None, dx, plain_coef, None, 1, 1, N, H, W, Co, pw2.CinPb, Cin, pw2.CoutPb,
                                 pw2.taps, epi=_lib.EPI_MPSUM, res=dadd.contiguous(), ta=1.0, tb=1.0, clip=0.0)
                else:
                    _conv_launch(dout, None, pw2.wb, None, dx, plain_coef, None, 1, 1, N, H, W, Co, pw2.CinPb, Cin, pw2.CoutPb,
                                 pw2.taps)
            if _wants_wgrad(pw2):
                _wgrad_launch(x, dout, pw2, plain_coef, 1, N, H, W, Cin, pw2.CinP, Co, pw2.CoutP, pw2.taps, N, N, 0, 0.0)
        if _wants_wgrad(pw2) or (gated and _wants_wgrad(pw3)):
            pw2.bank.request_finish()
        if cfg.res_slot is not None and dres is not None:
            cfg.res_slot.put(dres)                   # joins the other gradient of `res` inside that consumer's kernel
            dres = None
        return dx, None, None, dca, dcb, dcs, dres, None


_const_cache = {}


def _const_vec(n, value, dev):
    """(n,) fp32 vector filled with `value` (a per-frame coefficient that is the same for every frame), cached."""
    key = (n, float(value), str(dev))
    if key not in _const_cache:
        if len(_const_cache) > 64:
            _const_cache.clear()
        _const_cache[key] = torch.full((n,), float(value), dtype=torch.float32, device=dev)
    return _const_cache[key]


def _no_res_slot(cfg):
    """A copy of `cfg` whose res_slot was served (the tail of _ConvOp.backward parks `dres` otherwise)."""
    c = ConvCfg(cfg.pw2, cfg.pw3, cfg.B, cfg.T, cfg.epi, cfg.ta, cfg.tb, cfg.clip, cfg.need_grad, cfg.in_slot, None,
                cfg.grad_private)
    return c


_sel_cache = {}


def _clean_selector(B, T, dev):
    key = (B, T, str(dev))
    if key not in _sel_cache:
        sel = torch.zeros(B, 2, T, dtype=torch.float32, device=dev)
        sel[:, 0] = 1.0
        _sel_cache[key] = sel.reshape(-1)
    return _sel_cache[key]


def gate_coefs(gate):
    """mp_sum(y2, y3, g) = ca*y2 + cb*y3 with ca = (1-g)/sqrt((1-g)^2+g^2), cb = g/sqrt(...)  (utils.py:118-123)."""
    den = torch.rsqrt((1 - gate) ** 2 + gate ** 2)
    return (1 - gate) * den, gate * den


def conv(x, pw, res=None, ta=0.0, tb=0.0, clip=0.0, cscale=None, in_slot=None, res_slot=None, grad_private=False,
         res_alias=False):
    """MPConv forward on packed weights.  Optional fused epilogues: res -> clip(ta*res + tb*conv(x));
    cscale (N,Cout) fp32 -> silu(conv(x)*cscale)/0.596.  in_slot: GradSlot of x; res_slot: GradSlot that takes d res;
    grad_private / res_alias: see ConvCfg."""
    epi = "mpsum" if res is not None else ("emb_silu" if cscale is not None else "none")
    cfg = ConvCfg(pw, None, epi=epi, ta=ta, tb=tb, clip=clip, need_grad=torch.is_grad_enabled(), in_slot=in_slot,
                  res_slot=res_slot, grad_private=grad_private, res_alias=res_alias)
    sched = pw.param
    if getattr(pw, "members", None) and not sched.requires_grad:      # a group whose first member is frozen: any trainable member
        sched = next((m.param for m in pw.members if m.param.requires_grad), sched)     # keeps the node in the graph
    return _ConvOp.apply(x, sched, None, None, None, cscale, res, cfg)


def gated_conv_train(x, gate, pw2, pw3, B, T, coefs=None, res=None, ta=0.0, tb=0.0, clip=0.0, cscale=None, grad_private=False,
                     res_slot=None, res_alias=False):
    """Training-mode MPCausal3DGatedConv.  gate: (B*2*T,) fp32 autograd tensor, or precomputed coefs=(ca, cb).
    grad_private / res_slot / res_alias: see ConvCfg."""
    ca, cb = coefs if coefs is not None else gate_coefs(gate)
    epi = "mpsum" if res is not None else ("emb_silu" if cscale is not None else "none")
    cfg = ConvCfg(pw2, pw3, B, T, epi, ta, tb, clip, torch.is_grad_enabled(), grad_private=grad_private, res_slot=res_slot,
                  res_alias=res_alias)
    return _ConvOp.apply(x, pw2.param, pw3.param, ca, cb, cscale, res, cfg)


CAT_ACT_FUSED = int(_os.environ.get("ONIRIS_CAT_ACT_FUSED", "1"))      # 0: the decoder's mp_cat + mp_silu and its 1x1 skip conv as two launches (A/B)


def conv_cat_act_ok(x, skip, pw):
    """True when oniris_conv_fwd's two-source form (OnirisConvArgs.x2) serves this evaluation-sized 1x1 conv: the register-staged
    kernel takes it (mirrors conv_dispatch_1x1 / conv1x1_glds_ok in csrc)."""
    N, H, W, C1 = x.shape
    Cin = C1 + skip.shape[-1]
    glds = BIG_TILE >= 3 and Cin % 64 == 0 and pw.CinP == Cin and Cin <= 1024 and N * H * W >= 8192
    return bool(CAT_ACT_FUSED and x.is_cuda and not torch.is_grad_enabled() and pw.taps == 1 and pw.cin == Cin and C1 % 8 == 0
                and Cin % 8 == 0 and not glds and x.is_contiguous() and skip.is_contiguous())


@torch.no_grad()
def conv_cat_act(x, skip, w1, w2, pw):
    """Evaluation only (the sampler's one-frame UNet calls): (conv1x1(xo, W), a) with xo = mp_cat(x, skip) = [w1 x, w2 skip] rounded
    to bf16 and a = mp_silu(xo), in ONE launch -- xo itself is never written (reference networks_edm2.py:230 mp_cat, :73 mp_silu,
    :85 conv_skip: the head of a decoder Block).  Bit-identical to ops.act(x, skip, w1, w2, want_xo=True) followed by ops.conv."""
    N, H, W, C1 = x.shape
    Cin = C1 + skip.shape[-1]
    Co = roundup(pw.cout, 8)
    out = torch.empty((N, H, W, Co), dtype=BF16, device=x.device)
    a = torch.empty((N, H, W, Cin), dtype=BF16, device=x.device)
    _conv_launch(x, None, pw.wf, None, out, None, None, 1, 1, N, H, W, Cin, pw.CinP, Co, pw.CoutP, 1,
                 x2=skip, act_out=a, x_split=C1, cat_w=(float(w1), float(w2)))
    return out, a


KEEP_CTX_PRODUCT = int(_os.environ.get("ONIRIS_KEEP_CTX_PRODUCT", "1"))     # A/B knob: 0 = every evaluation recomputes y3


def ctx_product_ok(H, W, Cin, Cout):
    """True when the one-frame kernel (csrc/conv_eval1.h, conv_eval1_ok) serves this layer, i.e. when the context product
    of a cached pair can be kept between the evaluations of a frame (OnirisConvArgs.ctx_prod)."""
    return bool(KEEP_CTX_PRODUCT and BIG_TILE >= 3 and not (BIG_TILE & 16) and Cin % 32 == 0 and Cin >= 32 and H % 8 == 0
                and W % 8 == 0 and Cout % 8 == 0 and 2 * H * W * Cin * 2 < (1 << 31))


@torch.no_grad()
def gated_conv_ctx_product(pair, pw2, pw3, B, out=None):
    """y3 (B, H, W, Cout) fp32 = the un-gated context product of the cached pair (B, 2, H, W, C) with pw3 -- what every
    one-frame evaluation against this pair adds, times its gate coefficient (OnirisConvArgs.ctx_prod_mode 3)."""
    _, two, H, W, Cin = pair.shape
    assert two == 2 and pair.is_contiguous() and pair.dtype == BF16
    Co = roundup(pw2.cout, 8)
    y3 = out if out is not None else torch.empty((B, H, W, Co), dtype=torch.float32, device=pair.device)
    _conv_launch(None, pair, pw2.wf, pw3.wf, None, None, None, B, 1, 1, H, W, Cin, pw2.CinP, Co, pw2.CoutP, 9,
                 ctx_bstride=2, ctx_T=2, coff=(0, 1), ctx_fill=0.0, ctx_prod=y3, ctx_prod_mode=3)
    return y3


@torch.no_grad()
def gated_conv_eval(x, gate, pw2, pw3, B, t, ctx_frames, coefs=None, res=None, ta=0.0, tb=0.0, clip=0.0, cscale=None,
                    ctx_T=None, ctx_prod=None, ctx_prod_mode=0):
    """Eval-mode gated conv: x (B*t,H,W,C); ctx_frames (B, ctx_T, H, W, C) = [2 cached frames, x frames] (ctx_T = t+2),
    or just the 2 cached frames when t == 1 (output frame 0 reads context frames 0 and 1 only).  ctx_prod (t == 1 only):
    the kept context product of that pair, written (mode 1) or read instead of being recomputed (mode 2)."""
    ctx_T = t + 2 if ctx_T is None else ctx_T
    assert ctx_frames.shape[1] == ctx_T and (ctx_T == t + 2 or t == 1)
    N, H, W, Cin = x.shape
    Co = roundup(pw2.cout, 8)
    out = torch.empty((N, H, W, Co), dtype=BF16, device=x.device)
    ca, cb = coefs if coefs is not None else gate_coefs(gate)
    ca, cb = ca.float().contiguous(), cb.float().contiguous()
    kw, ret = {}, out
    if res is not None:
        kw = dict(epi=_lib.EPI_MPSUM, res=res, ta=ta, tb=tb, clip=clip)
    elif cscale is not None:
        ret = torch.empty_like(out)
        kw = dict(epi=_lib.EPI_EMB_SILU, escale=_rows_f32(cscale), out2=ret)
    if ctx_prod_mode:
        assert t == 1 and ctx_prod.dtype == torch.float32 and tuple(ctx_prod.shape) == (B, H, W, Co) and ctx_prod.is_contiguous()
        kw.update(ctx_prod=ctx_prod, ctx_prod_mode=ctx_prod_mode)
    _conv_launch(x, ctx_frames, pw2.wf, pw3.wf, out, ca, cb, B, 1, t, H, W, Cin, pw2.CinP, Co, pw2.CoutP, 9,
                 ctx_bstride=ctx_T, ctx_T=ctx_T, coff=(0, 1), ctx_fill=0.0, **kw)
    return ret


def sampler_update(mode, x_hat, x_pred, d_io, x_aux, x_out, t_a, dt, sigma_buf=None, sigma_next=0.0):
    """One launch for the sampler's update between two UNet evaluations (reference edm2/sampler.py:66-76), fp32 tensors
    of one shape: mode 0 (Euler) d_io = (x_hat - x_pred)/t_a, x_out = x_hat + dt*d_io; mode 1 (Heun) x_hat = x_out =
    x_hat + dt*(0.5*d_io + 0.5*(x_aux - x_pred)/t_a).  sigma_buf (optional, fp32): filled with sigma_next."""
    _need_gpu(x_hat)
    for t in (x_hat, x_pred, d_io, x_out) + ((x_aux,) if x_aux is not None else ()):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == x_hat.numel()
    check(lib.oniris_sampler_update(mode, _p(x_hat), _p(x_pred), _p(d_io), _p(x_aux), _p(x_out), x_hat.numel(), float(t_a),
                                    float(dt), _p(sigma_buf), sigma_buf.numel() if sigma_buf is not None else 0,
                                    float(sigma_next), _stream()), "sampler_update")


# ------------------------------------------------------------------------------------------------------------------
# fused magnitude-preserving glue (HBM-bound single-pass kernels, csrc/elementwise.hip)

class GradSlot:
    """Side channel for the gradient of an encoder output that is ALSO a skip connection (networks_edm2.py:227-230): such a
    tensor has two consumers, and autograd would add their two gradients with one more pass over three activation-sized
    tensors (12 per step, the largest 67 MB each).  Instead the decoder-side consumer (the mp_cat inside _ActFn, which runs
    first in backward: the whole decoder precedes the encoder) parks its gradient here and reports None to autograd; the
    encoder-side consumer -- the first op of the next block: pixel-norm / resample / 1x1 skip conv -- takes it and adds
    it inside its own backward kernel (`dadd` of oniris_act_bwd / `add` of oniris_resample / the mp_sum epilogue of the
    1x1 dgrad).  One slot per skip tensor and per forward pass."""
    __slots__ = ("g", "scale")
    live = []            # the slots of the current forward pass (cleared by WeightBank.prepare, checked at the end of backward)

    def __init__(self):
        self.g, self.scale = None, 1.0
        GradSlot.live.append(self)

    @classmethod
    def check_all_taken(cls):
        """End of backward: a slot that is still filled means its encoder-side consumer never ran -- a PARTIAL backward
        (torch.autograd.grad on decoder-side inputs only, a graph cut before the consumer) -- and the skip gradient would
        be dropped silently.  ONIRIS_GRAD_SLOTS=0 routes these gradients through autograd instead."""
        left = sum(1 for s in cls.live if s.g is not None)
        for s in cls.live:
            s.g = None
        cls.live = []
        if left:
            raise RuntimeError(f"{left} skip-connection gradient(s) were parked for an encoder-side backward kernel that "
                               "never ran (partial backward through the UNet?): they would have been dropped.  Run the "
                               "whole backward, or set ONIRIS_GRAD_SLOTS=0 to let autograd join these gradients")

    def put(self, g, scale=1.0):
        """scale != 1: the parked gradient is `scale * g` with g left as it is -- the residual gradient of an mp_sum epilogue is
        ta times the gradient of its output, a scaled copy nobody needs to write (the taker folds the factor into its kernel:
        take_scaled).  A second put materialises."""
        if self.g is None:
            self.g, self.scale = g, float(scale)
        else:
            self.g = (self.g if self.scale == 1.0 else self.g * self.scale) + (g if scale == 1.0 else g * scale)
            self.scale = 1.0

    def take(self):
        g, sc, self.g, self.scale = self.g, self.scale, None, 1.0
        return g if (g is None or sc == 1.0) else g * sc

    def take_scaled(self):
        """(g, scale) for a taker whose kernel applies the factor itself."""
        g, sc, self.g, self.scale = self.g, self.scale, None, 1.0
        return g, sc


class _ActFn(torch.autograd.Function):
    """(xo, a) = act(x[, skip]):  v = cat(w1*x, w2*skip); norm: v /= eps + |v|/sqrt(C); xo = v; a = silu(v)/0.596.
    in_slot / skip_slot: GradSlots of x / of skip (see GradSlot).  rs: x is resampled first (1 = 2x2 mean, 2 = nearest
    x2; Block.forward's first line) inside the same forward launch; the backward runs the resample adjoint behind
    act_bwd, and in_slot then belongs to the un-resampled x."""

    @staticmethod
    def forward(ctx, x, skip, w1, w2, norm, want_xo, in_slot=None, skip_slot=None, rs=0, xo_slot=None):
        _need_gpu(x)
        C1 = x.shape[-1]
        C2 = skip.shape[-1] if skip is not None else 0
        if rs:
            N_, Hi, Wi = x.shape[0], x.shape[1], x.shape[2]
            Ho, Wo = (Hi // 2, Wi // 2) if rs == 1 else (Hi * 2, Wi * 2)
            oshape = (N_, Ho, Wo)
        else:
            Ho = Wo = 0
            oshape = tuple(x.shape[:-1])
        npix = 1
        for d_ in oshape:
            npix *= d_
        shape = (*oshape, C1 + C2)
        a = torch.empty(shape, dtype=BF16, device=x.device)
        # want_xo without norm / cat / resample: xo IS x -- the input itself is handed back (autograd aliases it), so that a
        # tensor with two consumers (the activation and the residual / skip-conv path of a Block) reaches this node's
        # backward as (dxo, da) and the two gradients are added inside act_bwd instead of by a torch add over three tensors
        alias = want_xo and not norm and skip is None and not rs and x.is_contiguous()
        xo = torch.empty(shape, dtype=BF16, device=x.device) if ((want_xo or norm or rs) and not alias) else None
        sden = torch.empty(npix, dtype=torch.float32, device=x.device) if norm else None
        x = x.contiguous()
        skip = skip.contiguous() if skip is not None else None
        check(lib.oniris_act_fwd(_p(x), _p(skip), _p(xo), _p(a), _p(sden), npix, C1, C2, w1, w2, int(norm), rs, Ho, Wo,
                                 _stream()), "act_fwd")
        ctx.meta = (C1, C2, w1, w2, norm, npix, (*oshape, C1), skip.shape if skip is not None else None, rs, tuple(x.shape))
        ctx.slots = (in_slot, skip_slot, xo_slot)
        ctx.set_materialize_grads(False)                 # (an output whose consumer parked its gradient in xo_slot arrives as None,
                                                         #  not as a zero tensor that would then be ADDED to the parked one)
        # without norm/cat xo would just be x itself: reuse the input for the silu' evaluation
        ctx.save_for_backward(xo if xo is not None else x, sden)
        if alias:
            return x, a
        if want_xo or norm:
            return xo, a
        return a

    @staticmethod
    def backward(ctx, *grads):
        C1, C2, w1, w2, norm, npix, xshape, sshape, rs, xin_shape = ctx.meta
        xo, sden = ctx.saved_tensors
        if len(grads) == 2:
            dxo, da = grads
        else:
            dxo, da = None, grads[0]
        if da is None:
            da = torch.zeros(xo.shape, dtype=BF16, device=xo.device)
        da = da.contiguous()
        in_slot, skip_slot, xo_slot = ctx.slots
        dxo_scale = 1.0
        if xo_slot is not None:                          # the gradient of xo parked by its consumer, possibly as (g, scale): the
            parked, sc = xo_slot.take_scaled()           # residual gradient of an mp_sum epilogue = ta * (gradient of its output)
            if parked is not None:
                if dxo is None:
                    dxo, dxo_scale = parked, sc
                else:
                    dxo = dxo + (parked if sc == 1.0 else parked * sc)
        dxo = dxo.contiguous() if dxo is not None else None
        dx = torch.empty(xshape, dtype=BF16, device=xo.device)
        dskip = torch.empty(sshape, dtype=BF16, device=xo.device) if C2 else None
        dadd = in_slot.take() if in_slot is not None else None
        if dadd is not None:
            dadd = dadd.contiguous()
            assert tuple(dadd.shape) == (tuple(xin_shape) if rs else tuple(dx.shape))
        check(lib.oniris_act_bwd(_p(da), _p(dxo), _p(xo), _p(sden), _p(dx), _p(dskip), _p(None if rs else dadd), npix, C1, C2,
                                 w1, w2, int(norm), dxo_scale, _stream()), "act_bwd")
        if rs:                                           # adjoint of the resampling (+ the second gradient of the input)
            N_, Ho, Wo = xshape[0], xshape[1], xshape[2]
            dpre = torch.empty(xin_shape, dtype=BF16, device=xo.device)
            if rs == 1:    # adjoint of the 2x2 mean: nearest x2 scaled by 1/4
                check(lib.oniris_resample(_p(dx), _p(dpre), _p(dadd), N_, Ho, Wo, C1, 1, 0.25, _stream()), "resample")
            else:          # adjoint of nearest x2: 2x2 sum = 4 * mean
                check(lib.oniris_resample(_p(dx), _p(dpre), _p(dadd), N_, Ho, Wo, C1, 0, 4.0, _stream()), "resample")
            dx = dpre
        if skip_slot is not None and dskip is not None:
            skip_slot.put(dskip)                      # joins the encoder-side gradient inside that consumer's kernel
            dskip = None
        return dx, dskip, None, None, None, None, None, None, None, None


def act(x, skip=None, w1=1.0, w2=1.0, norm=False, want_xo=False, in_slot=None, skip_slot=None, resample="keep", xo_slot=None):
    """resample 'down' / 'up': x is resampled first, in the same launch (the reference's Block.forward, :63).
    xo_slot: GradSlot in which a consumer of xo parks its gradient (GradSlot.put(g, scale)) instead of handing it to autograd."""
    rs = {"keep": 0, "down": 1, "up": 2}[resample]
    return _ActFn.apply(x, skip, float(w1), float(w2), bool(norm), bool(want_xo), in_slot, skip_slot, rs, xo_slot)


def resample_taps(f):
    """The normalised 1-D taps of a resampling filter (reference utils.py:94-100: even length, divided by its sum), or None for
    the [1, 1] filter every BASELINE configuration uses (networks_edm2.py:26) -- that one runs inside the fused kernels."""
    f = [float(v) for v in f]
    if len(f) < 2 or len(f) > 8 or len(f) % 2 != 0:
        raise NotImplementedError(f"resample filter {f}: an even number of taps, 2 .. 8")
    tot = sum(f)
    if tot == 0:
        raise ValueError(f"resample filter {f} sums to zero")
    t = tuple(v / tot for v in f)
    return None if t == (0.5, 0.5) else t


def _resample_launch(src, dst, add, N, H, W, C, mode, scale, taps):
    if taps is None:
        check(lib.oniris_resample(_p(src), _p(dst), _p(add), N, H, W, C, mode, scale, _stream()), "resample")
    else:
        arr = (ctypes.c_float * len(taps))(*taps)
        check(lib.oniris_resample_filter(_p(src), _p(dst), _p(add), N, H, W, C, mode, arr, len(taps), scale, _stream()),
              "resample_filter")


class _ResampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, in_slot=None, taps=None):
        _need_gpu(x)
        N, H, W, C = x.shape
        x = x.contiguous()
        out = torch.empty((N, H // 2, W // 2, C) if mode == 0 else (N, H * 2, W * 2, C), dtype=BF16, device=x.device)
        _resample_launch(x, out, None, N, H, W, C, mode, 1.0, taps)
        ctx.mode, ctx.slot, ctx.taps = mode, in_slot, taps
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, H, W, C = g.shape
        dadd = ctx.slot.take() if ctx.slot is not None else None        # (see GradSlot)
        if dadd is not None:
            dadd = dadd.contiguous()
        if ctx.mode == 0:      # adjoint of the strided filter: the transposed one (weights 4 f f^T) scaled by 1/4
            dx = torch.empty((N, H * 2, W * 2, C), dtype=BF16, device=g.device)
            assert dadd is None or dadd.shape == dx.shape
            _resample_launch(g, dx, dadd, N, H, W, C, 1, 0.25, ctx.taps)
        else:                  # adjoint of the transposed filter: the strided one scaled by 4
            dx = torch.empty((N, H // 2, W // 2, C), dtype=BF16, device=g.device)
            assert dadd is None or dadd.shape == dx.shape
            _resample_launch(g, dx, dadd, N, H, W, C, 0, 4.0, ctx.taps)
        return dx, None, None, None


def resample(x, mode, in_slot=None, taps=None):
    """mode 'keep' | 'down' | 'up' (utils.py:94-107); taps: resample_taps(f) -- None = the [1, 1] filter (2x2 mean / nearest x2)."""
    if mode == "keep":
        return x
    return _ResampleFn.apply(x, 0 if mode == "down" else 1, in_slot, taps)


class _SplitCols(torch.autograd.Function):
    """x (N, sum(sizes)) -> its column blocks as VIEWS (the conv epilogue and the backward pre-pass read a block in place
    through a row pitch: no copy per block); backward = one concatenation of the blocks' gradients."""

    @staticmethod
    def forward(ctx, x, sizes):
        ctx.sizes, ctx.n = tuple(sizes), x.shape[0]
        outs, o = [], 0
        xd = x.detach()
        for sz in sizes:
            outs.append(xd[:, o:o + sz])
            o += sz
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        zeros = None
        parts = []
        for g, sz in zip(grads, ctx.sizes):
            if g is None:
                if zeros is None:
                    zeros = ref.new_zeros((ctx.n, max(ctx.sizes)))
                g = zeros[:, :sz]
            parts.append(g)
        return torch.cat(parts, dim=1), None


FUSED_PRELUDE = 1              # tests set 0 AND install `prelude_reference` (tests/torch_prelude.py): the torch-autograd
prelude_reference = None       # formulation of gates / embedding / emb scales lives with the tests, not in the product


def _prelude_ref(name):
    if prelude_reference is None:
        raise RuntimeError("the conditioning prelude runs on the fused HIP kernels only (HIP tensors, FUSED_PRELUDE = 1); its "
                           "torch formulation is test infrastructure: tests/torch_prelude.py")
    return getattr(prelude_reference, name)


def direct_pack(params, holder, name):
    """parallel.ParamPack over `params` when ONE FlatParams owns them all (their values are then gathered from, and their
    gradients added into, the flat buffers without any autograd node), else None.  Cached per (holder, name)."""
    from .parallel import FlatParams
    p0 = params[0]
    flat = FlatParams.owner_of(p0)
    if flat is None:
        return None
    hit = _pack_cache.get((id(holder), name))
    if hit is not None and hit.flat is flat and len(hit.ids) == len(params) and hit.ids[0] == id(p0) and \
            hit.ids[-1] == id(params[-1]) and p0.data_ptr() == flat.flat.data_ptr() + 4 * flat.offset_of(p0):
        return hit
    base = flat.flat.data_ptr()
    for p in params:
        if FlatParams.owner_of(p) is not flat or p.dtype != torch.float32 or p.data_ptr() != base + 4 * flat.offset_of(p):
            return None
    if len(_pack_cache) > 64:
        _pack_cache.clear()
    hit = _pack_cache[(id(holder), name)] = flat.direct_pack(params)
    return hit


_pack_cache = {}


EMB_BWD_CHUNKS = 16       # ONIRIS_EMB_BWD_CHUNKS in include/oniris.h


class _EmbScaleFn(torch.autograd.Function):
    """c = 1 + c_all * gain[seg]  (oniris_emb_scale / oniris_emb_scale_bwd).  sink (parallel.ParamPack or None): where the
    gain gradient goes when `gain` has no autograd history."""

    @staticmethod
    def forward(ctx, c_all, gain, seg, start, sink=None):
        N, Ctot = c_all.shape
        c = torch.empty((N, Ctot), dtype=torch.float32, device=c_all.device)
        check(lib.oniris_emb_scale(_p(c_all), _p(gain), _p(seg), _p(c), N, Ctot, _stream()), "emb_scale")
        ctx.save_for_backward(c_all, gain, start)
        ctx.sink = sink
        return c

    @staticmethod
    def backward(ctx, dc):
        c_all, gain, start = ctx.saved_tensors
        N, Ctot = c_all.shape
        dc = dc.contiguous()
        dc_all = torch.empty_like(c_all)
        part = torch.empty((gain.numel(), EMB_BWD_CHUNKS), dtype=torch.float32, device=gain.device)
        check(lib.oniris_emb_scale_bwd(_p(dc), _p(c_all), _p(gain), _p(start), _p(dc_all), _p(part), N, Ctot, gain.numel(),
                                       _stream()), "emb_scale_bwd")
        dgain = part.sum(1).reshape(gain.shape)
        if ctx.sink is not None:
            ctx.sink.deliver(dgain)
            dgain = None
        return dc_all, dgain, None, None, None


class _EmbedPostFn(torch.autograd.Function):
    """emb = mp_silu(mp_sum(e1, e2, t)) on bf16 (N, C) rows  (oniris_embed_post / _bwd); e2 may be None."""

    @staticmethod
    def forward(ctx, e1, e2, t):
        emb = torch.empty_like(e1)
        check(lib.oniris_embed_post(_p(e1), _p(e2), _p(emb), e1.numel(), t, _stream()), "embed_post")
        ctx.save_for_backward(e1, e2)
        ctx.t = t
        return emb

    @staticmethod
    def backward(ctx, demb):
        e1, e2 = ctx.saved_tensors
        demb = demb.contiguous()
        de1 = torch.empty_like(e1)
        de2 = torch.empty_like(e2) if e2 is not None else None
        check(lib.oniris_embed_post_bwd(_p(demb), _p(e1), _p(e2), _p(de1), _p(de2), e1.numel(), ctx.t, _stream()),
              "embed_post_bwd")
        return de1, de2, None


def embed_train(c_noise, labels, fourier, pw_noise, pw_label, label_dim, t=1 / 3):
    """The UNet's embedding with gradients to the two linear weights (networks_edm2.py:204-212): one launch builds the
    bf16 inputs of both linears (Fourier features, scaled one-hot), the linears run on the packed weights, one launch
    does mp_sum + mp_silu.  Returns emb (N, 1, 1, cemb) bf16."""
    _need_gpu(c_noise)
    N, dev = c_noise.numel(), c_noise.device
    cn = fourier.freqs.numel()
    cnP = roundup(cn, 8)
    four = torch.empty((N, 1, 1, cnP), dtype=BF16, device=dev)
    oh, LP = None, 0
    if labels is not None and pw_label is not None:
        LP = roundup(label_dim, 8)
        oh = torch.empty((N, 1, 1, LP), dtype=BF16, device=dev)
        labels = labels.reshape(-1).to(torch.int64).contiguous()      # (the kernel reads `const long long*`)
    freqs, phases = fourier.freqs, fourier.phases
    if freqs.dtype != torch.float32:
        freqs, phases = freqs.float(), phases.float()
    check(lib.oniris_embed_pre(_p(c_noise), _p(labels) if oh is not None else None, _p(freqs), _p(phases), _p(four), _p(oh), N,
                               cn, cnP, label_dim, LP, _stream()), "embed_pre")
    e1 = conv(four, pw_noise)
    e2 = conv(oh, pw_label) if oh is not None else None
    return _EmbedPostFn.apply(e1, e2, float(t))


class _GatesFn(torch.autograd.Function):
    """(ca, cb) [L][N] of all gating layers (oniris_gates) with the adjoint to the packed parameters (oniris_gates_bwd)."""

    @staticmethod
    def forward(ctx, c_noise, params, nctx, T, sink=None, anchor=None):
        L, N = params.shape[0], c_noise.numel()
        out = torch.empty((2, L, N), dtype=torch.float32, device=c_noise.device)
        params = params.contiguous()
        check(lib.oniris_gates(_p(c_noise), _p(params), _p(nctx), _p(out[0]), _p(out[1]), L, N, T, _stream()), "gates")
        ctx.save_for_backward(c_noise, params, nctx)
        ctx.T, ctx.sink = T, sink
        return out[0], out[1]

    @staticmethod
    def backward(ctx, dca, dcb):
        c_noise, params, nctx = ctx.saved_tensors
        L, N = params.shape[0], c_noise.numel()
        if dca is None and dcb is None:
            return None, None, None, None, None, None
        dca = torch.zeros((L, N), dtype=torch.float32, device=params.device) if dca is None else dca.contiguous()
        dcb = torch.zeros((L, N), dtype=torch.float32, device=params.device) if dcb is None else dcb.contiguous()
        dparams = torch.empty_like(params)
        check(lib.oniris_gates_bwd(_p(c_noise), _p(params), _p(nctx), _p(dca), _p(dcb), _p(dparams), L, N, ctx.T, _stream()),
              "gates_bwd")
        if ctx.sink is not None:
            ctx.sink.deliver(dparams)
            dparams = None
        return None, dparams, None, None, None, None


def gates_train(c_noise, params, nctx, T, sink=None, anchor=None):
    """Training counterpart of gates_eval: params (L, 6) fp32 (mult0, mult1, off0, off1, min_gating, max_gating per layer)
    either WITH autograd history, or gathered by a parallel.ParamPack `sink` that also receives the gradient (`anchor`:
    any tensor that requires grad, so that autograd schedules the node); c_noise (N,) fp32 contiguous; returns ca, cb (L, N)."""
    _need_gpu(c_noise, params)
    return _GatesFn.apply(c_noise, params, nctx, T, sink, anchor)


def eval_gain_vector(gpw, gains):
    """The emb_gain parameters as one fp32 vector for the eval path (31 evaluations per generated frame): rebuilt only when
    a parameter changed (torch writes bump ._version, the raw-pointer optimizer bumps _weights_epoch).  Inside a hipGraph
    capture nothing is cached (the tensor would live in the graph's memory): UNet.prewarm_eval fills the cache before."""
    sig = (_weights_epoch, tuple(x._version for x in gains), tuple(x.data_ptr() for x in gains))
    hit = getattr(gpw, "gaincache", None)
    if hit is not None and hit[0] == sig:
        return hit[1]
    g = torch.cat([x.reshape(1) for x in gains]).float()
    if not (g.is_cuda and torch.cuda.is_current_stream_capturing()):
        gpw.gaincache = (sig, g)
    return g


def emb_scales(emb, gpw, gains):
    """All `c = emb_linear(emb) * emb_gain + 1` of a UNet (networks_edm2.py:78 in every Block) at once:
    emb (N,1,1,Cemb) bf16, gpw the row-concatenated emb_linear group (WeightBank.add_group), gains the emb_gain
    parameters in member order.  Returns one contiguous (N, cout_k) fp32 tensor per member."""
    N = emb.shape[0]
    c_all = conv(emb, gpw).reshape(N, gpw.cout)
    dev = emb.device
    if FUSED_PRELUDE and c_all.is_cuda:
        # (kept ON the group object: a dictionary keyed by id(gpw) handed a recycled id the tables of a dead net)
        cache = getattr(gpw, "embcache", None)
        if cache is not None and cache[1].device != dev:
            cache = None
        if cache is None:
            sizes, seg, start = [], [], [0]
            for k, m in enumerate(gpw.members):
                w = roundup(m.cout, 64)
                sizes.append(m.cout)
                if w != m.cout:
                    sizes.append(w - m.cout)
                seg += [k] * w
                start.append(start[-1] + w)
            assert start[-1] == gpw.cout, (start[-1], gpw.cout)
            cache = gpw.embcache = (tuple(sizes), torch.tensor(seg, dtype=torch.int32, device=dev),
                                    torch.tensor(start, dtype=torch.int32, device=dev))
        sizes, seg, start = cache
        pack = direct_pack(gains, gpw, "_gain_pack") if torch.is_grad_enabled() else None
        if pack is not None:
            g = pack.values()
        elif not torch.is_grad_enabled():
            g = eval_gain_vector(gpw, gains)
        else:
            g = torch.cat([x.reshape(1) for x in gains])
            if g.dtype != torch.float32:
                g = g.float()
        outs = _SplitCols.apply(_EmbScaleFn.apply(c_all, g, seg, start, pack), sizes)
        res, j = [], 0
        for m in gpw.members:
            res.append(outs[j])
            j += 2 if roundup(m.cout, 64) != m.cout else 1
        return res
    return _prelude_ref("emb_scales")(c_all, gpw, gains, _SplitCols)



# ------------------------------------------------------------------------------------------------------------------
# attention

def rope_tables(inv_freq, scale_vec, n_pos, device, scale_base=64):
    """fp32 cos/sin/scale tables (n_pos, 64) built from fp16-ROUNDED angles and scales exactly like
    RotaryEmbedding.make_rotary_embedding (RoPe.py:21-32: the fp16 rounding is part of the numerical spec).
    The three are row ranges of MASTER tables kept per RotaryEmbedding: cos / sin of position t do not depend on n_pos,
    and the xPos scale of position t is scale**((t - n_pos // 2) / scale_base), a function of the integer t - n_pos // 2
    only -- so a rollout, whose key count grows by one per generated frame, uploads nothing per frame (a host->device
    copy from pageable memory waits for everything queued on the stream: it used to stall the host once per frame and
    layer, with the previous frame's 31 evaluations still in the queue)."""
    # The tables live ON the `inv_freq` tensor object (a module buffer: same object for the module's life), keyed by the
    # in-place versions of both buffers: no global map from device addresses -- the allocator recycles a freed module's
    # addresses, and a dictionary keyed by them handed a 16-channel module the tables of a dead 64-channel one (round 4) --
    # and a load_state_dict() that rewrites the buffers invalidates them.
    try:
        _rope_cache = inv_freq.__dict__.setdefault("_oniris_rope_tables", {})
    except AttributeError:                                         # (a tensor type without a __dict__: no caching)
        _rope_cache = {}
    key = (str(device), id(scale_vec), scale_base, inv_freq._version, scale_vec._version, inv_freq.numel())
    m = _rope_cache.get(key)
    if m is None or m[0] < n_pos:
        cap = max(64, 2 * n_pos, 2 * (m[0] if m is not None else 0))
        inv, sv = inv_freq.detach().float().cpu(), scale_vec.detach().float().cpu()
        t = torch.arange(cap, dtype=torch.float32)
        ang = torch.outer(t, inv)
        ang = torch.cat([ang, ang], -1).to(torch.float16)
        d0 = cap // 2                                              # row r of the scale master <-> offset t - n_pos//2 = r - d0
        power = (torch.arange(-d0, cap, dtype=torch.float32)) / scale_base
        sc = sv[None, :] ** power[:, None]
        sc = torch.cat([sc, sc], -1).to(torch.float16)
        m = _rope_cache[key] = (cap, d0) + tuple(z.float().contiguous().to(device) for z in (ang.cos(), ang.sin(), sc))
    _, d0, cos, sin, sc = m
    off = d0 - n_pos // 2
    return cos[:n_pos], sin[:n_pos], sc[off:off + n_pos]


def _rope(x, xr, xt, tabs, mode, B, frames, P, C, pos_offset, pos_mod, x_bstride=0, xr_bstride=0):
    cs, sn, sc = tabs if tabs is not None else (None, None, None)
    check(lib.oniris_rope(_p(x), _p(xr), _p(xt), _p(cs), _p(sn), _p(sc), mode, B, frames, P, C, pos_offset, pos_mod,
                          x_bstride, xr_bstride, _stream()), "rope")


def _attn_args(q, k, v, qt, kt, vt, out, lse, tabs, B, heads, Lq, Lk, C, mask_mode, P, T):
    a = _lib.AttnArgs()
    a.q, a.k, a.v, a.qt, a.kt, a.vt, a.out, a.lse = _p(q), _p(k), _p(v), _p(qt), _p(kt), _p(vt), _p(out), _p(lse)
    if tabs is not None:
        num, idx, qn, qi, a.tab_block = tabs
        a.kv_num, a.kv_idx, a.q_num, a.q_idx = _p(num), _p(idx), _p(qn), _p(qi)
        a.tab_cols, a.qtab_cols = idx.shape[1], qi.shape[1]
    a.B, a.heads, a.Lq, a.Lk, a.C = B, heads, Lq, Lk, C
    a.mask_mode, a.P, a.T = mask_mode, P, T
    a.frame_kernel = (0 if FRAME_KERNEL else 1) | (0 if DECODE_STREAMS else 2) | (0 if FRAME_FWD_HALVES else 4)
    return a


# 1: dense attention inside frames of 64 / 128 / 256 tokens (FrameAttention, just_2d) on the whole-frame kernels of
# csrc/attention_frame.h; 0: on the generic grid kernels (A/B, tests)
FRAME_KERNEL = int(_os.environ.get("ONIRIS_FRAME_KERNEL", "1"))


# 1: one new frame against a KV ring below the split-KV threshold walks its key tiles in four streams per workgroup (attn_fwd_kernel<0, 4>);
# 0: one stream (rounds 1-5; A/B, tests)
DECODE_STREAMS = int(_os.environ.get("ONIRIS_DECODE_STREAMS", "1"))


# 1: a forward launch over a few 256-token frames (one frame per sequence in the cached sampler) runs two query halves per frame
# (frame_attn_fwd_kernel<1>); 0: one workgroup per frame and head (A/B, tests)
FRAME_FWD_HALVES = int(_os.environ.get("ONIRIS_FRAME_FWD_HALVES", "1"))


FRAME_BWD_FUSED = int(_os.environ.get("ONIRIS_FRAME_BWD_FUSED", "1"))      # 0: attn_delta + dQ + dK/dV as three launches (A/B, tests)


def _frame_kernel_serves(mask_mode, Lq, Lk):
    """Mirrors frame_attn_ok() (csrc/attention_frame.h) for launches without table, schedule, ring strides or split-KV."""
    return bool(FRAME_KERNEL) and mask_mode == 0 and Lq == Lk and Lq in (64, 128, 256)


# 1: FrameAttention with frames of 128 * 2^k tokens on the persistent work lists (block-diagonal table, mask_mode 1).  Measured at
# the bench shape (1024 frames of 16x16 tokens, 2 heads; scratch/r06_frame_attn_ab.py, profiles/r06_ab_frame_ws.txt): forward
# 194-202 us vs 180-196 us on the grid kernels, backward 494-507 vs 382-417 us -- two-block items do not amortise the persistent
# kernels' per-item prologue / epilogue, so the grid kernels stay the default; the path is kept under test (VERDICT r05 next #5b).
FRAME_WS = int(_os.environ.get("ONIRIS_FRAME_WS", "0"))


def _host_table(kind, n, P):
    """(kv_num, kv_idx, block) on the host: the DART training table of T = n frames ('video') or the block-diagonal table of n
    frames ('frame', frame_tables)."""
    if kind == "video":
        return train_mask_table(n, P)
    return np.ones(n, dtype=np.int32), np.arange(n, dtype=np.int32).reshape(n, 1), P


def _train_sched(T, P, n_pairs, dev, which):
    return _table_sched("video", T, P, n_pairs, dev, which)


def _table_sched(kind, n, P, n_pairs, dev, which):
    """Schedule of a table's blocks over the persistent workgroups: which = 'fwd' (query blocks of 128 rows weighted by their
    key-block count) or 'dkv' (key blocks weighted by their query-block count).  kind / n: see _host_table."""
    num, idx, blk = _host_table(kind, n, P)
    if which == "fwd":
        per = blk // 128
        w = np.repeat(num * per + 1, per)                    # a table row of `blk` tokens = per 128-row query blocks
    elif which == "dkv":                                     # items of 64 keys: 2 per 128-token block, whole query list each
        qn = mask_transpose(num, idx)[0] if kind == "video" else num          # (block-diagonal: its own transpose)
        per = blk // 128
        w = np.repeat(qn * per + 1, 2 * per)
    else:                                                    # 'dkv128': items of 128 keys (twice the work per listed block)
        qn = mask_transpose(num, idx)[0] if kind == "video" else num
        per = blk // 128
        w = np.repeat(qn * per + 1, per)
    return attn_schedule(w, n_pairs, dev)


DKV_ITEM_KEYS = int(_os.environ.get("ONIRIS_DKV_ITEM_KEYS", "0"))     # 64 / 128: force the dK/dV item size (A/B, tests); 0: by load


def _dkv_item_keys(kind, T, P, n_pairs, dev):
    """128-key items for the persistent dK/dV kernel when the launch has enough of them: the heaviest item (the first key block:
    every later query block attends it) must stay well below a workgroup's average load, or the launch takes as long as that
    one item (C2 at B = 2: 8 pairs x 64 items of up to 64 units on 256 workgroups = 35 units on average -> 64-key items;
    B = 8: 140 on average -> 128-key items)."""
    if DKV_ITEM_KEYS in (64, 128):
        return DKV_ITEM_KEYS
    num, idx, blk = _host_table(kind, T, P)
    qn = mask_transpose(num, idx)[0] if kind == "video" else num
    w = qn * (blk // 128) + 1
    n_wg = max(8, (_cu_count.get(str(dev)) or torch.cuda.get_device_properties(dev).multi_processor_count) - _cu_reserve)
    return 128 if float(w.sum()) * (blk // 128) * n_pairs / n_wg >= 1.5 * float(w.max()) else 64


def _attn_core_fwd(qr, kr, v, kind, B, T, heads, P):
    """The attention launch of a training step on prepared q (rotated, carrying the softmax scale), k, v (N, P, heads*64):
    'video' = DART training table over B sequences of 2T frames, 'frame' = dense per frame.  Returns (out, lse, tabs, meta)."""
    N, _, C = qr.shape
    dev = qr.device
    if kind == "video":
        frames = N // B
        Bq, L = B, frames * P
        mask_mode = 2
        tabs = device_tables("train", T, P, dev)
        if tabs is None:
            raise RuntimeError(f"make_train_mask returns None for T={T}, P={P} (T*P must be a multiple of 128)")
    else:
        frames, Bq, L = 1, N, P
        mask_mode, tabs = 0, None
        g = next(g for g in (8, 4, 2, 1) if N % g == 0)
        if (FRAME_WS and ATTN_PERSISTENT and P % 128 == 0 and (P & (P - 1)) == 0 and (N // g) * (P // 128) < 65536
                and g * heads < 32768):
            # FrameAttention with frames of 128 * 2^k tokens (16x16 latents: the gym net) on the persistent work lists: the N frames
            # are g pseudo-sequences of N / g frames under a block-diagonal table + the frame-causal mask (everything inside a
            # frame): 2-block items back to back on one workgroup per CU instead of a grid of 2-block workgroups
            frames, Bq, L, mask_mode = N // g, g, (N // g) * P, 1
            tabs = frame_tables(frames, P, dev)
    fl = _attn_flops(kind, N if kind != "video" else Bq, T, heads, P if kind != "video" else L, P)
    out = torch.empty((N, P, C), dtype=BF16, device=dev)
    lse = torch.empty((Bq, heads, L), dtype=torch.float32, device=dev)
    a = _attn_args(qr, kr, v, None, None, None, out, lse, tabs, Bq, heads, L, L, C, mask_mode, P, T)
    ks = 2 if (mask_mode != 0 and L >= 2048) else 1
    name = "frame_attn_fwd_kernel" if _frame_kernel_serves(mask_mode, L, L) else f"attn_fwd_kernel<MODE={mask_mode},KS={ks}>"
    if mask_mode != 0 and ATTN_PERSISTENT and tabs[1].shape[1] <= 64 and L % 128 == 0 and Bq * heads < 32768:
        # persistent kernel: query blocks of 128 rows, cost = key blocks of its table row + 1 (fixed per-item work)
        sched = _table_sched(kind, T if kind == "video" else frames, P, Bq * heads, dev, "fwd")
        a.sched, a.sched_wgs, a.sched_slots = _p(sched[0]), sched[1], sched[2]
        name = f"attn_fwd_ws_kernel<MODE={mask_mode}>"
    _profiled(name, fl, lambda: check(lib.oniris_attn_fwd(ctypes.byref(a), _stream()), "attn_fwd"))
    return out, lse, tabs, (kind, B, T, heads, Bq, L, frames, P, C, mask_mode)


def _attn_core_bwd(qr, kr, v, out, lse, dout, tabs, meta):
    """dq (w.r.t. the unscaled q), dk, dv of _attn_core_fwd."""
    kind, B, T, heads, Bq, L, frames, P, C, mask_mode = meta
    dev = qr.device
    if _frame_kernel_serves(mask_mode, L, L) and FRAME_BWD_FUSED:
        # dense attention inside frames of 64 / 128 / 256 tokens: delta, dQ, dK, dV in ONE launch that reads q, k, v, out, dout once
        dq, dk, dv = torch.empty_like(qr), torch.empty_like(kr), torch.empty_like(v)
        a = _attn_args(qr, kr, v, None, None, None, out, lse, None, Bq, heads, L, L, C, mask_mode, P, T)
        a.dout, a.dq, a.dk, a.dv = _p(dout), _p(dq), _p(dk), _p(dv)
        fl = _attn_flops(kind, B, T, heads, P, P)
        _profiled("frame_attn_bwd_kernel", 3.5 * fl, lambda: check(lib.oniris_frame_attn_bwd(ctypes.byref(a), _stream()), "frame_attn_bwd"))
        return dq, dk, dv
    delta = torch.empty((Bq, heads, L), dtype=torch.float32, device=dev)
    dkv_ws = (mask_mode != 0 and ATTN_PERSISTENT and ATTN_DKV_PERSISTENT and tabs[3].shape[1] <= 64 and L % 128 == 0
              and Bq * heads < 32768)
    tab_n = T if kind == "video" else frames            # what the table was built from (frames of the training half / of a pseudo-sequence)
    neg = torch.empty((2, Bq, heads, L), dtype=torch.float32, device=dev) if dkv_ws else None     # -lse | -delta
    check(lib.oniris_attn_bwd_prep(_p(dout), _p(out), _p(delta), None, _p(lse) if dkv_ws else None, _p(neg), Bq, heads, L, C,
                                   _stream()), "attn_bwd_prep")
    dq, dk, dv = torch.empty_like(qr), torch.empty_like(kr), torch.empty_like(v)
    a = _attn_args(qr, kr, v, None, None, None, out, lse, tabs, Bq, heads, L, L, C, mask_mode, P, T)
    a.dout, a.delta, a.dq, a.dk, a.dv = _p(dout), _p(delta), _p(dq), _p(dk), _p(dv)
    fl = _attn_flops(kind, B if kind != "video" else Bq, T, heads, P if kind != "video" else L, P)
    ks = 2 if (mask_mode != 0 and L >= 2048) else 1
    if dkv_ws and ATTN_DQ_PERSISTENT and tabs[1].shape[1] <= 64:
        # persistent dQ kernel on the forward's work list (query blocks, longest first); reads the NEGATED row constants
        sched = _table_sched(kind, tab_n, P, Bq * heads, dev, "fwd")
        a.sched, a.sched_wgs, a.sched_slots = _p(sched[0]), sched[1], sched[2]
        a.lse, a.delta = _p(neg[0]), _p(neg[1])
        _profiled(f"attn_bwd_dq_ws_kernel<MODE={mask_mode}>", 1.5 * fl,
                  lambda: check(lib.oniris_attn_bwd_dq(ctypes.byref(a), _stream()), "attn_bwd_dq"))
        a.sched, a.sched_wgs, a.sched_slots = None, 0, 0
        a.lse, a.delta = _p(lse), _p(delta)
    else:
        _profiled("frame_attn_dq_kernel" if _frame_kernel_serves(mask_mode, L, L) else f"attn_bwd_dq_kernel<MODE={mask_mode},KS={ks}>", 1.5 * fl,
                  lambda: check(lib.oniris_attn_bwd_dq(ctypes.byref(a), _stream()), "attn_bwd_dq"))
    if dkv_ws:
        # persistent kernel: items of 64 keys with their whole query list, longest first over one workgroup per CU:
        # dK / dV leave the kernel finished (no fp32 partial sums, no reduction launch); it reads the NEGATED row constants
        keys = _dkv_item_keys(kind, tab_n, P, Bq * heads, dev)
        sched = _table_sched(kind, tab_n, P, Bq * heads, dev, "dkv128" if keys == 128 else "dkv")
        a.sched, a.sched_wgs, a.sched_slots = _p(sched[0]), sched[1], sched[2]
        a.dkv_item_keys = keys
        a.lse, a.delta = _p(neg[0]), _p(neg[1])
        _profiled(f"attn_bwd_dkv_ws_kernel<MODE={mask_mode}>", 2.0 * fl,
                  lambda: check(lib.oniris_attn_bwd_dkv(ctypes.byref(a), _stream()), "attn_bwd_dkv"))
    else:
        # causal tables: split every key block's query list so that no workgroup walks more than ~32 sub-tiles
        nch = max(1, min(ATTN_DKV_CHUNKS, L // ATTN_DKV_MIN_L)) if mask_mode == 2 else 1
        if nch > 1:
            part = torch.empty((2, nch, Bq, L, C), dtype=torch.float32, device=dev)
            a.dkv_part, a.dkv_chunks = _p(part), nch
        _profiled(f"attn_bwd_dkv_kernel<MODE={mask_mode}>", 2.0 * fl,
                  lambda: check(lib.oniris_attn_bwd_dkv(ctypes.byref(a), _stream()), "attn_bwd_dkv"))     # (+ the partial-sum reduce)
    return dq, dk, dv


FRAME_QKV_FUSED = int(_os.environ.get("ONIRIS_FRAME_QKV_FUSED", "1"))      # 0: qkv normalisation as passes of their own around the frame kernels (A/B, tests)


class _FrameAttentionQkvFn(torch.autograd.Function):
    """FrameAttention's core on frames of 64 / 128 / 256 tokens straight from the attn_qkv output (oniris_frame_attn_qkv_fwd / _bwd,
    csrc/attention_frame.h): qkv (N, P, 3C) bf16 -> (N, P, C); the per-head normalisation and its adjoint run inside the two
    launches, nothing but qkv, out and lse is kept for the backward."""

    @staticmethod
    def forward(ctx, qkv, heads):
        _need_gpu(qkv)
        qkv = qkv.contiguous()
        N, P, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty((N, P, C), dtype=BF16, device=qkv.device)
        lse = torch.empty((N, heads, P), dtype=torch.float32, device=qkv.device)
        fl = _attn_flops("frame", N, 1, heads, P, P)
        # HBM-bound (128 FLOP per byte at 256 tokens per frame): qkv read once, out + lse written
        _profiled("frame_attn_qkv_fwd_kernel", fl,
                  lambda: check(lib.oniris_frame_attn_qkv_fwd(_p(qkv), _p(out), _p(lse), N, P, heads, _stream()), "frame_attn_qkv_fwd"),
                  nbytes=2.0 * N * P * 4 * C + 4.0 * N * heads * P)
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        N, P, _ = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        fl = _attn_flops("frame", N, 1, ctx.heads, P, P)
        C = qkv.shape[2] // 3
        _profiled("frame_attn_qkv_bwd_kernel", 3.5 * fl,
                  lambda: check(lib.oniris_frame_attn_qkv_bwd(_p(qkv), _p(out), _p(lse), _p(dout), _p(dqkv), N, P, ctx.heads, _stream()),
                                "frame_attn_qkv_bwd"),
                  nbytes=2.0 * N * P * 8 * C + 4.0 * N * ctx.heads * P)            # qkv, out, dout read; dqkv written
        return dqkv, None


class _AttentionFn(torch.autograd.Function):
    """qkv (N, P, 3C) bf16 (channel = s*C + head*64 + c)  ->  attention output (N, P, C).
    kind: 'video' (DART training mask + RoPE over frames, B sequences of 2T frames) or 'frame' (dense per frame)."""

    @staticmethod
    def forward(ctx, qkv, kind, B, T, heads, rope_bufs, need_grad):
        _need_gpu(qkv)
        N, P, C3 = qkv.shape
        C = C3 // 3
        if C != 64 * heads:
            raise NotImplementedError(f"_AttentionFn: 64-channel heads (got {C} channels / {heads} heads: _AttentionHdFn)")
        dev = qkv.device
        q = torch.empty((N, P, C), dtype=BF16, device=dev)
        k, v = torch.empty_like(q), torch.empty_like(q)
        fused_rope = kind == "video" and FUSED_ROPE
        if not fused_rope:
            check(lib.oniris_qkv_norm(_p(qkv), _p(q), _p(k), _p(v), N * P, C, 0, 0, 0, _stream()), "qkv_norm")
        tabs_r = rope_tables(rope_bufs[0], rope_bufs[1], T, dev) if kind == "video" else None
        if fused_rope:              # normalisation + rotation in one pass over qkv (frames = 2T: position = frame mod T)
            cs_, sn_, sc_ = tabs_r
            check(lib.oniris_qkv_norm_rope(_p(qkv), _p(q), _p(k), _p(v), _p(cs_), _p(sn_), _p(sc_), N * P, C, P, T, _stream()),
                  "qkv_norm_rope")
            qr, kr = q, k
        elif kind == "video":
            frames = N // B
            qr, kr = torch.empty_like(q), torch.empty_like(k)
            _rope(q, qr, None, tabs_r, 1, B, frames, P, C, 0, T)
            _rope(k, kr, None, tabs_r, 2, B, frames, P, C, 0, T)
        else:
            qr, kr = q, k
        out, lse, tabs, meta = _attn_core_fwd(qr, kr, v, kind, B, T, heads, P)
        ctx.meta = meta
        ctx.tabs, ctx.tabs_r = tabs, tabs_r
        ctx.save_for_backward(qkv, qr, kr, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, qr, kr, v, out, lse = ctx.saved_tensors
        kind, B, T, heads, Bq, L, frames, P, C, mask_mode = ctx.meta
        dout = dout.contiguous()
        dq, dk, dv = _attn_core_bwd(qr, kr, v, out, lse, dout, ctx.tabs, ctx.meta)
        dqkv = torch.empty_like(qkv)
        N = qkv.shape[0]
        if kind == "video" and FUSED_ROPE:
            cs_, sn_, sc_ = ctx.tabs_r
            check(lib.oniris_qkv_norm_rope_bwd(_p(qkv), _p(dq), _p(dk), _p(dv), _p(dqkv), _p(cs_), _p(sn_), _p(sc_), N * P, C, P, T,
                                               _stream()), "qkv_norm_rope_bwd")
            return dqkv, None, None, None, None, None, None
        if kind == "video":
            dqn, dkn = torch.empty_like(dq), torch.empty_like(dk)
            _rope(dq, dqn, None, ctx.tabs_r, 3, Bq, frames, P, C, 0, T)
            _rope(dk, dkn, None, ctx.tabs_r, 4, Bq, frames, P, C, 0, T)
        else:
            dqn, dkn = dq, dk
        check(lib.oniris_qkv_norm_bwd(_p(qkv), _p(dqn), _p(dkn), _p(dv), _p(dqkv), N * P, C, _stream()), "qkv_norm_bwd")
        return dqkv, None, None, None, None, None, None


HEAD_DIMS_PADDED = (8, 16, 24, 32, 40, 48, 56)     # served through the 64-channel kernels on zero-padded heads (csrc/attention_hd.hip)


HEAD_DIM_WIDE_MAX = 256                            # 64 < d <= 256, d % 8 == 0: the generic fp32 attention kernel (fp32.wide_heads_*; slow)


def _head_dim(C, heads):
    d = C // max(heads, 1)
    wide = 64 < d <= HEAD_DIM_WIDE_MAX and d % 8 == 0
    if heads <= 0 or C != heads * d or (d != 64 and d not in HEAD_DIMS_PADDED and not wide):
        raise NotImplementedError(f"attention head dimension {C}/{heads}: 64 (native), a multiple of 8 below 64 (padded into the "
                                  f"64-channel kernels) or a multiple of 8 up to {HEAD_DIM_WIDE_MAX} (generic fp32 attention kernel)")
    return d


def _pad_heads(t, heads, d):
    """(N, P, heads*d) -> (N, P, heads*64) with zeros in channels d..63 of every head."""
    N, P, _ = t.shape
    out = torch.zeros((N, P, heads, 64), dtype=t.dtype, device=t.device)
    out[..., :d] = t.reshape(N, P, heads, d)
    return out.view(N, P, heads * 64)


def _unpad_heads(t, heads, d):
    N, P, _ = t.shape
    return t.view(N, P, heads, 64)[..., :d].reshape(N, P, heads * d)


class _AttentionHdFn(torch.autograd.Function):
    """_AttentionFn for heads of 8 / 16 / 32 channels: per-head norm (+ rotary embedding) into zero-padded 64-channel heads
    (oniris_qkv_norm_hd), the SAME attention launches, the padding dropped from the result."""

    @staticmethod
    def forward(ctx, qkv, kind, B, T, heads, rope_bufs, need_grad):
        _need_gpu(qkv)
        N, P, C3 = qkv.shape
        d = _head_dim(C3 // 3, heads)
        dev = qkv.device
        qkv = qkv.contiguous()
        q = torch.empty((N, P, heads * 64), dtype=BF16, device=dev)
        k, v = torch.empty_like(q), torch.empty_like(q)
        tabs_r = rope_tables(rope_bufs[0], rope_bufs[1], T, dev) if kind == "video" else (None, None, None)
        rope = 3 if kind == "video" else 0
        check(lib.oniris_qkv_norm_hd(_p(qkv), _p(q), _p(k), _p(v), _p(tabs_r[0]), _p(tabs_r[1]), _p(tabs_r[2]), N * P, heads, d, P,
                                     T if kind == "video" else 1, 0, rope, T if kind == "video" else 1, _stream()), "qkv_norm_hd")
        out, lse, tabs, meta = _attn_core_fwd(q, k, v, kind, B, T, heads, P)
        ctx.meta, ctx.tabs, ctx.tabs_r, ctx.hd = meta, tabs, tabs_r, (d, rope)
        ctx.save_for_backward(qkv, q, k, v, out, lse)
        return _unpad_heads(out, heads, d)

    @staticmethod
    def backward(ctx, dout):
        qkv, q, k, v, out, lse = ctx.saved_tensors
        kind, B, T, heads, Bq, L, frames, P, C, mask_mode = ctx.meta
        d, rope = ctx.hd
        dq, dk, dv = _attn_core_bwd(q, k, v, out, lse, _pad_heads(dout.contiguous(), heads, d), ctx.tabs, ctx.meta)
        dqkv = torch.empty_like(qkv)
        N = qkv.shape[0]
        cs_, sn_, sc_ = ctx.tabs_r
        check(lib.oniris_qkv_norm_hd_bwd(_p(qkv), _p(dq), _p(dk), _p(dv), _p(dqkv), _p(cs_), _p(sn_), _p(sc_), N * P, heads, d, P,
                                         T if kind == "video" else 1, 0, rope, T if kind == "video" else 1, _stream()),
              "qkv_norm_hd_bwd")
        return dqkv, None, None, None, None, None, None


def attention_train(qkv, kind, B, T, heads, rope_bufs=None):
    d = _head_dim(qkv.shape[-1] // 3, heads)
    if d > 64:                                      # no BASELINE configuration: generality (networks_edm2.py:28,39), not speed
        from . import fp32 as _fp32
        return _fp32.wide_heads_train(qkv, kind, B, T, heads, rope_bufs)
    if (d == 64 and kind == "frame" and FRAME_KERNEL and FRAME_QKV_FUSED and qkv.shape[1] in (64, 128, 256)
            and not (FRAME_WS and qkv.shape[1] % 128 == 0)):
        return _FrameAttentionQkvFn.apply(qkv, heads)
    fn = _AttentionFn if d == 64 else _AttentionHdFn
    return fn.apply(qkv, kind, B, T, heads, rope_bufs, torch.is_grad_enabled())


class KVRing:
    """Preallocated K / V storage of one VideoAttention layer during a rollout: [B][cap frames * P][C]
    bf16 each, `n` committed frames.  The cache entry the modules hand around stays the reference's (K, V) pair of
    (B, n*P, C) tensors -- they are VIEWS of the ring (`K._oniris_ring`), so nothing is concatenated or copied per
    evaluation: the qkv kernel writes the new frames' k, v straight behind the committed ones (an evaluation with
    update_cache=False leaves them uncommitted: the next one overwrites them), RoPE reads the ring through a batch
    stride, the decode kernel reads V in place.  Appending in place is only done for the cache that owns the ring's
    latest state; a continuation from an older (K, V) pair -- two futures from one cache -- gets a ring of its own."""

    def __init__(self, B, P, C, cap, device):
        self.B, self.P, self.C, self.cap, self.n = B, P, C, cap, 0
        self.K = torch.empty((B, cap * P, C), dtype=BF16, device=device)
        self.V = torch.empty((B, cap * P, C), dtype=BF16, device=device)
        # ROTATED image of K for ONE key count (RoPe.py:55-57 re-rotates every key whenever the count changes): the
        # committed frames are rotated once per generated frame (rotate_committed, from UNet.prewarm_eval), the new frame's
        # key is added by the fused qkv kernel of each of the frame's 31 evaluations -- instead of one pass over the whole
        # ring per evaluation.  kr_state = (committed frames, key count) the image is valid for.
        self.KR = torch.empty((B, cap * P, C), dtype=BF16, device=device)
        self.kr_state = None

    @torch.no_grad()
    def rotate_committed(self, rope_bufs):
        n, nk = self.n, self.n + 1
        if self.kr_state == (n, nk):
            return
        if n > 0:
            tabs_r = rope_tables(rope_bufs[0], rope_bufs[1], nk, self.K.device)
            bstride = self.cap * self.P * self.C
            _rope(self.K, self.KR, None, tabs_r, 2, self.B, n, self.P, self.C, 0, nk, x_bstride=bstride, xr_bstride=bstride)
        self.kr_state = (n, nk)

    def views(self):
        k, v = self.K[:, :self.n * self.P], self.V[:, :self.n * self.P]
        k._oniris_ring = self
        k._oniris_n = self.n
        return k, v

    @staticmethod
    def of(kv_cache, B, P, C, t_new, device, grow=True):
        """The ring that holds kv_cache (None: empty) with room for t_new more frames -- the cache's own ring when it is
        its latest state and large enough, else a new one (the old frames are copied once)."""
        n = 0 if kv_cache is None else kv_cache[0].shape[1] // P
        ring = getattr(kv_cache[0], "_oniris_ring", None) if kv_cache is not None else None
        if (ring is not None and ring.n == n and getattr(kv_cache[0], "_oniris_n", -1) == n and ring.cap >= n + t_new
                and ring.B == B and ring.P == P and ring.C == C):
            return ring
        if not grow:
            return None
        new = KVRing(B, P, C, max(16, 2 * (n + t_new)), device)
        if n:
            new.K[:, :n * P].copy_(kv_cache[0])
            new.V[:, :n * P].copy_(kv_cache[1])
        new.n = n
        return new


DECODE_SPLIT_MIN_KEYS = int(_os.environ.get("ONIRIS_DECODE_SPLIT_MIN", "2048"))   # split-KV decode from this many keys on


def _decode_splits(a, B, heads, Lq, Lk, dev):
    """One new frame against a long KV ring: ceil(Lq / 128) * heads * B workgroups (4 at B = 1) would walk all keys serially
    (~0.5 us per 64-key tile: 130 us per layer at 264 cached frames); deal the key tiles to kv_splits workgroups each."""
    if Lk < DECODE_SPLIT_MIN_KEYS:
        return
    wgs = -(-Lq // 128) * heads * B
    ns = max(1, min(64, Lk // 512, 512 // max(1, wgs)))
    if ns > 1:
        ws = torch.empty((ns, B, heads, Lq, 65), dtype=torch.float32, device=dev)
        a.split_ws, a.kv_splits = _p(ws), ns
        a._keep = ws                                # (keeps the workspace alive until the launch is queued)


FUSED_QKV_EVAL = int(_os.environ.get("ONIRIS_FUSED_QKV_EVAL", "1"))     # 0: conv + qkv_norm[_rope_eval] as separate launches (A/B aid)


def _qkv_eval(x, pw, q, k, v, kr, tabs, ntok, C, kv_tpb, kv_bstride, kv_off, pos):
    cs_, sn_, sc_ = tabs if tabs is not None else (None, None, None)
    check(lib.oniris_qkv_eval(_p(x), _p(pw.wf), _p(q), _p(k), _p(v), _p(kr), _p(cs_), _p(sn_), _p(sc_), ntok, C, pw.CinP,
                              kv_tpb, kv_bstride, kv_off, pos, _stream()), "qkv_eval")


@torch.no_grad()
def attention_eval_x(x, pw, B, heads, rope_bufs, kv_cache, update_cache, P):
    """attention_eval from the layer INPUT x (N, H, W, C) and the packed attn_qkv weight: for one new frame against a
    prepared KV ring the 1x1 convolution, the normalisation and the rotation are one launch (oniris_qkv_eval), 3 launches
    per VideoAttention layer and evaluation instead of 4; every other case runs the convolution and attention_eval."""
    N, C = x.shape[0], x.shape[-1]
    t = N // B
    dev = x.device
    ring = KVRing.of(kv_cache, B, P, C, t, dev) if (FUSED_QKV_EVAL and t == 1 and C == 64 * heads and pw.cout == 3 * C) else None
    if ring is None or ring.kr_state != (ring.n, ring.n + 1):
        qkv = conv(x, pw).reshape(N, P, 3 * C)
        return attention_eval(qkv, B, heads, rope_bufs, kv_cache, update_cache, P)
    n, nk = ring.n, ring.n + 1
    bstride = ring.cap * P * C
    q = torch.empty((N, P, C), dtype=BF16, device=dev)
    _qkv_eval(x.contiguous(), pw, q, ring.K, ring.V, ring.KR, rope_tables(rope_bufs[0], rope_bufs[1], nk, dev), N * P, C, P,
              bstride, n * P, n)
    if update_cache:
        ring.n = nk
        new_cache = ring.views()
    else:
        new_cache = kv_cache
    out = torch.empty((N, P, C), dtype=BF16, device=dev)
    a = _attn_args(q, ring.KR, ring.V, None, None, None, out, None, None, B, heads, P, nk * P, C, 0, P, 0)
    a.v_bstride = a.k_bstride = bstride
    _decode_splits(a, B, heads, P, nk * P, dev)
    check(lib.oniris_attn_fwd(ctypes.byref(a), _stream()), "attn_fwd")
    return out, new_cache


@torch.no_grad()
def frame_attention_eval(x, pw, heads):
    """FrameAttention without autograd (attention_modules.py:105-119) from the layer input: qkv convolution + normalisation
    in one launch, dense attention inside every frame."""
    N, H, W, C = x.shape
    P = H * W
    dev = x.device
    if not (FUSED_QKV_EVAL and C == 64 * heads and pw.cout == 3 * C):
        return attention_train(conv(x, pw).reshape(N, P, 3 * C), "frame", N, 1, heads)
    q = torch.empty((N, P, C), dtype=BF16, device=dev)
    k, v = torch.empty_like(q), torch.empty_like(q)
    _qkv_eval(x.contiguous(), pw, q, k, v, None, None, N * P, C, 0, 0, 0, 0)
    out = torch.empty((N, P, C), dtype=BF16, device=dev)
    lse = torch.empty((N, heads, P), dtype=torch.float32, device=dev)
    a = _attn_args(q, k, v, None, None, None, out, lse, None, N, heads, P, P, C, 0, P, 1)
    check(lib.oniris_attn_fwd(ctypes.byref(a), _stream()), "attn_fwd")
    return out


@torch.no_grad()
def _attention_eval_hd(qkv, B, heads, rope_bufs, kv_cache, update_cache, P):
    """attention_eval for heads of 8 / 16 / 32 channels (zero-padded into the 64-channel kernels, csrc/attention_hd.hip).  The
    cache is the reference's: normalised, UN-rotated k and v of all frames so far -- here (B, frames*P, heads*64) padded --
    and every call re-rotates all keys for the grown key count (RoPe.py:55-57); no KV ring on this path."""
    N, _, C3 = qkv.shape
    d = _head_dim(C3 // 3, heads)
    dev = qkv.device
    t = N // B
    Cp = heads * 64
    n = 0 if kv_cache is None else kv_cache[0].shape[1] // P
    nk = n + t
    cs_, sn_, sc_ = rope_tables(rope_bufs[0], rope_bufs[1], nk, dev)
    q = torch.empty((N, P, Cp), dtype=BF16, device=dev)
    knew, vnew = torch.empty_like(q), torch.empty_like(q)
    # q: normalised, scaled and rotated at positions n .. nk-1; k, v of the new frames: normalised only
    check(lib.oniris_qkv_norm_hd(_p(qkv.contiguous()), _p(q), _p(knew), _p(vnew), _p(cs_), _p(sn_), _p(sc_), N * P, heads, d, P, nk, n,
                                 1, t, _stream()), "qkv_norm_hd")
    knew, vnew = knew.view(B, t * P, Cp), vnew.view(B, t * P, Cp)
    if kv_cache is not None:
        K = torch.cat([kv_cache[0][:, :n * P], knew], dim=1)
        V = torch.cat([kv_cache[1][:, :n * P], vnew], dim=1)
    else:
        K, V = knew, vnew
    K, V = K.contiguous(), V.contiguous()
    new_cache = (K, V) if update_cache else kv_cache
    kr = torch.empty_like(K)
    check(lib.oniris_rope_hd(_p(K), _p(kr), _p(cs_), _p(sn_), _p(sc_), B * nk * P, heads, d, P, nk, 0, 2, nk, _stream()), "rope_hd")
    Lq, Lk = t * P, nk * P
    out = torch.empty((N, P, Cp), dtype=BF16, device=dev)
    if t == 1:
        mask_mode, tabs = 0, None                               # one new frame: dense SDPA over all keys (:69-70)
    elif Lq == Lk:
        mask_mode, tabs = 1, device_tables("infer", t, P, dev)  # causal prefill (:72-75)
    else:
        raise NotImplementedError("The inference mask is not implemented for this case")
    a = _attn_args(q, kr, V, None, None, None, out, None, tabs, B, heads, Lq, Lk, Cp, mask_mode, P, 0)
    if t == 1:
        _decode_splits(a, B, heads, Lq, Lk, dev)
    check(lib.oniris_attn_fwd(ctypes.byref(a), _stream()), "attn_fwd")
    return _unpad_heads(out, heads, d), new_cache


@torch.no_grad()
def attention_eval(qkv, B, heads, rope_bufs, kv_cache, update_cache, P):
    """Eval-mode VideoAttention core (attention_modules.py:51-77): qkv (B*t, P, 3C) of the NEW frames.
    kv_cache: (K, V) normalised, un-rotated, (B, t_cached*P, C) or None.  Returns out (B*t,P,C), new cache."""
    if _head_dim(qkv.shape[-1] // 3, heads) > 64:
        from . import fp32 as _fp32
        return _fp32.wide_heads_eval(qkv, B, heads, rope_bufs, kv_cache, update_cache, P)
    if _head_dim(qkv.shape[-1] // 3, heads) != 64:
        return _attention_eval_hd(qkv, B, heads, rope_bufs, kv_cache, update_cache, P)
    N, P_, C3 = qkv.shape
    C = C3 // 3
    dev = qkv.device
    t = N // B
    ring = KVRing.of(kv_cache, B, P, C, t, dev)
    n = ring.n
    q = torch.empty((N, P, C), dtype=BF16, device=dev)
    bstride = ring.cap * P * C
    nk = n + t
    if t == 1 and ring.kr_state == (n, nk):
        # one new frame, and the ring's rotated image is ready for this key count (UNet.prewarm_eval): normalisation + the
        # rotation of the new q / k in ONE launch, dense attention straight over the ring -- 4 launches per layer and
        # evaluation instead of 6, and no pass over all cached keys
        cs_, sn_, sc_ = rope_tables(rope_bufs[0], rope_bufs[1], nk, dev)
        check(lib.oniris_qkv_norm_rope_eval(_p(qkv), _p(q), _p(ring.K), _p(ring.V), _p(ring.KR), _p(cs_), _p(sn_), _p(sc_),
                                            N * P, C, P, bstride, n * P, n, _stream()), "qkv_norm_rope_eval")
        if update_cache:
            ring.n = nk
            new_cache = ring.views()
        else:
            new_cache = kv_cache
        out = torch.empty((N, P, C), dtype=BF16, device=dev)
        a = _attn_args(q, ring.KR, ring.V, None, None, None, out, None, None, B, heads, P, nk * P, C, 0, P, 0)
        a.v_bstride = a.k_bstride = bstride
        _decode_splits(a, B, heads, P, nk * P, dev)
        check(lib.oniris_attn_fwd(ctypes.byref(a), _stream()), "attn_fwd")
        return out, new_cache
    check(lib.oniris_qkv_norm(_p(qkv), _p(q), _p(ring.K), _p(ring.V), N * P, C, t * P, bstride, n * P, _stream()), "qkv_norm")
    if update_cache:
        ring.n = nk
        new_cache = ring.views()
    else:
        new_cache = kv_cache
    Lq, Lk = t * P, nk * P
    tabs_r = rope_tables(rope_bufs[0], rope_bufs[1], nk, dev)
    qr = torch.empty_like(q)
    kr = torch.empty((B, Lk, C), dtype=BF16, device=dev)
    _rope(q, qr, None, tabs_r, 1, B, t, P, C, nk - t, nk)
    _rope(ring.K, kr, None, tabs_r, 2, B, nk, P, C, 0, nk, x_bstride=bstride)
    out = torch.empty((N, P, C), dtype=BF16, device=dev)
    if t == 1:
        mask_mode, tabs = 0, None                               # one new frame: dense SDPA over all keys (:69-70)
    elif Lq == Lk:
        mask_mode = 1                                           # causal prefill (:72-75)
        tabs = device_tables("infer", t, P, dev)
    else:
        raise NotImplementedError("The inference mask is not implemented for this case")
    a = _attn_args(qr, kr, ring.V, None, None, None, out, None, tabs, B, heads, Lq, Lk, C, mask_mode, P, 0)
    a.v_bstride = bstride
    if t == 1:
        _decode_splits(a, B, heads, Lq, Lk, dev)
    check(lib.oniris_attn_fwd(ctypes.byref(a), _stream()), "attn_fwd")
    return out, new_cache


# ------------------------------------------------------------------------------------------------------------------
# optimizer

# Channels of the packed UNet input (image channels + the ones channel, zero-padded).  32 (round 5): the stem conv and its weight
# gradient then have the 32-channel shape the streaming kernels take (conv_stream.h, conv_plain_stream.h, conv_wgrad_stream.h, and
# conv_eval1.h in the sampler) instead of the register-staged kernels' 16 -- twice the input bytes, at 2-3x their rate.  16: A/B.
IN_PAD = int(_os.environ.get("ONIRIS_IN_PAD", "32"))


def dart_input(images, noise, sigma, S, sigma_data, want_c_noise=False):
    """Packed UNet input of a DART training step (oniris_dart_input): images (B,T,C,H,W), noise (B,S*T,C,H,W), sigma
    (B,S*T) fp32 -> (B*S*T, H, W, IN_PAD) bf16 = c_in * (images + sigma*noise) with the ones channel, zeros behind it.
    want_c_noise: also return c_noise = log(sigma)/4 like sigma (networks_edm2.py:291), from the same launch."""
    _need_gpu(images, noise, sigma)                 # (noise None: c_in * images, Precond's input side in eval)
    B, T, C, H, W = images.shape
    xcl = torch.empty((B * S * T, H, W, IN_PAD), dtype=BF16, device=images.device)
    cn = torch.empty((B, S * T), dtype=torch.float32, device=images.device) if want_c_noise else None
    check(lib.oniris_dart_input(_p(images), _p(noise), _p(sigma), _p(xcl), B, S, T, C, H, W, float(sigma_data), _p(cn), IN_PAD,
                                _stream()), "dart_input")
    return (xcl, cn) if want_c_noise else xcl


@torch.no_grad()
def precond_out(F, x, sigma, out_gain, sigma_data):
    """D = c_skip * x + c_out * out_gain * F (Precond.forward's output side, eval): F (N,H,W,8) bf16 raw UNet output,
    x (B,t,C,H,W) fp32, sigma (B,t) -> D like x."""
    _need_gpu(F, x, sigma)
    B, t, C, H, W = x.shape
    assert F.dtype == BF16 and F.shape == (B * t, H, W, 8) and F.is_contiguous() and x.dtype == torch.float32 and x.is_contiguous()
    D = torch.empty_like(x)
    og = out_gain.detach().float().reshape(1)
    check(lib.oniris_precond_out(_p(F), _p(x), _p(sigma.float().contiguous()), _p(og), _p(D), B * t, C, H, W,
                                 float(sigma_data), _stream()), "precond_out")
    return D


@torch.no_grad()
def embed_eval(c_noise, labels, fourier, w_noise, w_label, label_dim):
    """(N, 1, 1, cemb) bf16 embedding of the UNet in eval, one launch (oniris_embed_eval)."""
    _need_gpu(c_noise, w_noise)
    N, (cemb, cn) = c_noise.numel(), w_noise.shape
    emb = torch.empty((N, 1, 1, cemb), dtype=BF16, device=c_noise.device)
    lab = labels.reshape(-1).to(torch.int64).contiguous() if (labels is not None and w_label is not None) else None
    check(lib.oniris_embed_eval(_p(c_noise), _p(lab), _p(fourier.freqs.float()), _p(fourier.phases.float()), _p(w_noise),
                                _p(w_label if lab is not None else None), _p(emb), N, cn, cemb, int(label_dim), _stream()),
          "embed_eval")
    return emb


@torch.no_grad()
def gates_eval(c_noise, params, nctx, T):
    """(ca, cb) [L][N] for all gating layers in one launch (oniris_gates); params (L,6) fp32, nctx (L,) int32 or None."""
    _need_gpu(c_noise, params)
    L, N = params.shape[0], c_noise.numel()
    out = torch.empty((2, L, N), dtype=torch.float32, device=c_noise.device)
    check(lib.oniris_gates(_p(c_noise), _p(params), _p(nctx), _p(out[0]), _p(out[1]), L, N, T, _stream()), "gates")
    return out[0], out[1]


class _DartLoss(torch.autograd.Function):
    """losses[b,t] = mean_{c,h,w} (c_skip*x + c_out*out_gain*F - images)^2 over the noised half (edm2/loss.py:37-38 with
    Precond's output scaling, networks_edm2.py:293-297); F = raw channels-last UNet output.  Backward writes dF in
    bf16 directly (zero for the clean slots) and d out_gain."""

    @staticmethod
    def forward(ctx, F, out_gain, images, noise, sigma, S, sigma_data):
        _need_gpu(F, images, noise, sigma)
        B, T, C, H, W = images.shape
        assert F.dtype == BF16 and F.shape == (B * S * T, H, W, 8) and F.is_contiguous()
        og = out_gain.detach().float().reshape(1).contiguous()
        losses = torch.empty((B, T), dtype=torch.float32, device=F.device)
        check(lib.oniris_dart_loss(_p(F), _p(images), _p(noise), _p(sigma), _p(og), _p(losses), B, S, T, C, H, W,
                                   float(sigma_data), _stream()), "dart_loss")
        ctx.save_for_backward(F, og, images, noise, sigma)
        ctx.cfg = (S, float(sigma_data), out_gain.shape)
        return losses

    @staticmethod
    def backward(ctx, g):
        F, og, images, noise, sigma = ctx.saved_tensors
        S, sd, gshape = ctx.cfg
        B, T, C, H, W = images.shape
        dF = torch.empty_like(F)
        part = torch.empty((B, T), dtype=torch.float32, device=F.device)
        check(lib.oniris_dart_loss_bwd(_p(F), _p(images), _p(noise), _p(sigma), _p(og), _p(g.float().contiguous()), _p(dF),
                                       _p(part), B, S, T, C, H, W, sd, _stream()), "dart_loss_bwd")
        return dF, part.sum().reshape(gshape), None, None, None, None, None


def dart_loss(F, out_gain, images, noise, sigma, S, sigma_data):
    return _DartLoss.apply(F, out_gain, images, noise, sigma, S, sigma_data)


class _LossTail(torch.autograd.Function):
    """(loss, un-weighted loss) = tail of EDM2Loss (oniris_loss_tail): lambda(sigma) weighting, division by the fitted mean
    loss, both means, and the (sigma, loss, position) history append -- one launch, nothing read back by the host."""

    @staticmethod
    def forward(ctx, mse, sigma, coef, history, sigma_data, T_off):
        _need_gpu(mse, sigma, coef)
        B, T = mse.shape
        assert mse.dtype == torch.float32 and mse.is_contiguous() and sigma.dtype == torch.float32 and sigma.stride(1) == 1
        out = torch.empty(2, dtype=torch.float32, device=mse.device)
        dcoef = torch.empty((B, T), dtype=torch.float32, device=mse.device)
        rs = rl = rp = cnt = None
        cap = 0
        if history is not None:
            rs, rl, rp, cnt = history
            cap = rs.numel()
        c = coef.detach().reshape(-1)
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        check(lib.oniris_loss_tail(_p(mse), _p(sigma), _p(c), _p(out), _p(dcoef), _p(rs), _p(rl), _p(rp), _p(cnt), cap, B, T,
                                   sigma.stride(0), T_off, (c.numel() + 1) // 2, float(sigma_data), _stream()), "loss_tail")
        ctx.save_for_backward(dcoef)
        loss, unweighted = out[0], out[1]
        ctx.mark_non_differentiable(unweighted)
        return loss, unweighted

    @staticmethod
    def backward(ctx, g, _g_unweighted):
        (dcoef,) = ctx.saved_tensors
        return dcoef * g, None, None, None, None, None


def loss_tail(mse, sigma, coef, history, sigma_data):
    """mse (B, T) fp32 per-frame MSE of the noised half; sigma (B, S*T) fp32 (the LAST T columns are read);
    coef: the Fourier coefficients of the fitted mean loss; history: (ring_sigma, ring_loss, ring_pos, count) device
    tensors of MultiNoiseLoss or None.  Returns (loss, un-weighted loss) as 0-d device tensors."""
    return _LossTail.apply(mse, sigma, coef, history, float(sigma_data), sigma.shape[1] - mse.shape[1])


SQNORM_WS = 1024          # ONIRIS_SQNORM_WS in include/oniris.h


def sqnorm_(g, norm_buf):
    """norm_buf[0] <- sum(g^2) (deterministic two-stage reduction; norm_buf: 1 + SQNORM_WS floats)."""
    _need_gpu(g, norm_buf)
    check(lib.oniris_sqnorm(_p(g), g.numel(), _p(norm_buf), _stream()), "sqnorm")


def adamw_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, max_norm=None, norm_buf=None,
           emas=(), norm_ready=False):
    """Optimizer side of a training step on flat fp32 buffers (gym_train.py:105-108): optional gradient-norm clipping
    (norm_buf: 1 + SQNORM_WS floats of scratch, receives sum(g^2) in [0]), AdamW, and the power-function EMA update
    of up to two tracked copies -- emas = [(flat_tensor, 1 - beta), ...]."""
    _need_gpu(p, g, m, v)
    global _weights_epoch
    _weights_epoch += 1
    if max_norm is None and not emas and step >= 1:
        check(lib.oniris_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                               grad_scale, _stream()), "adamw")
        return
    assert len(emas) <= 2
    gn = None
    if max_norm is not None and step >= 1:
        assert norm_buf is not None and norm_buf.numel() >= 1 + SQNORM_WS and norm_buf.dtype == torch.float32
        if not norm_ready:             # (norm_ready: the caller ran sqnorm_ over the WHOLE gradient buffer already)
            check(lib.oniris_sqnorm(_p(g), g.numel(), _p(norm_buf), _stream()), "sqnorm")
        gn = norm_buf
    e = list(emas) + [(None, 0.0)] * (2 - len(emas))
    check(lib.oniris_adamw_clip_ema(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                    grad_scale, _p(gn), float(max_norm or 0.0), _p(e[0][0]), float(e[0][1]),
                                    _p(e[1][0]), float(e[1][1]), _stream()), "adamw_clip_ema")
