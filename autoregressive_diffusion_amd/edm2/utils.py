"""Magnitude-preserving primitives on channels-last bf16 tensors + BetterModule (checkpoint helpers).

Mirrors the interface of the reference's edm2/utils.py (normalize :83-88, resample :94-107, mp_silu :112,
mp_sum :118-123, mp_cat :128-134, MPFourier :139-150, BetterModule :13-72) -- written from scratch.  These are the
HBM-bound glue ops between the HIP kernels; they run as device elementwise ops on (N, H, W, C) bf16 tensors.
"""
import math
import numpy as np
import torch
from torch import nn

EPS = 1e-4
SILU_SCALE = 1.0 / 0.596
BF16 = torch.bfloat16


def to_cl(x, pad_to=None):
    """(N, C, H, W) any float dtype -> (N, H, W, C[padded]) bf16 contiguous."""
    x = x.permute(0, 2, 3, 1)
    if pad_to is not None and x.shape[-1] < pad_to:
        x = torch.nn.functional.pad(x, (0, pad_to - x.shape[-1]))
    return x.to(BF16).contiguous()


def from_cl(x, dtype=torch.float32):
    return x.permute(0, 3, 1, 2).to(dtype).contiguous()


def normalize(x, dim=None, eps=EPS):
    """x / (eps + |x|_dim * sqrt(n_norm / n_x)): the reference's magnitude normalisation (utils.py:83-88), part of its
    public utils surface (its tests import it); the hot path runs it inside the HIP kernels."""
    if dim is None:
        dim = list(range(1, x.ndim))
    norm = torch.linalg.vector_norm(x, dim=dim, keepdim=True, dtype=torch.float32)
    norm = eps + norm * math.sqrt(norm.numel() / x.numel())
    return x / norm.to(x.dtype)


def normalize_cl(x):
    """pixel norm over the channel (last) dim: x / (eps + |x| / sqrt(C))   (utils.py:83-88 with dim=1)."""
    n = torch.linalg.vector_norm(x.float(), dim=-1, keepdim=True)
    return (x.float() / (EPS + n * (1.0 / math.sqrt(x.shape[-1])))).to(x.dtype)


def mp_silu(x):
    return torch.nn.functional.silu(x) * SILU_SCALE


def mp_sum(a, b, t=0.5):
    """(1 - t) a + t b, renormalised (utils.py:118-123); t may be a per-sample (B,) / per-sample-and-channel (B, C) tensor."""
    if torch.is_tensor(t) and t.dim() > 0:
        return bmult(a + bmult(b - a, t), ((1 - t) ** 2 + t ** 2) ** -0.5)
    return (a + (b - a) * t) * (1.0 / math.sqrt((1 - t) ** 2 + t ** 2))


def mp_cat_cl(a, b, t=0.5):
    Na, Nb = a.shape[-1], b.shape[-1]
    C = math.sqrt((Na + Nb) / ((1 - t) ** 2 + t ** 2))
    return torch.cat([a * (C / math.sqrt(Na) * (1 - t)), b * (C / math.sqrt(Nb) * t)], dim=-1)


def resample_cl(x, mode="keep"):
    """f = [1,1]: 'down' = 2x2 mean, 'up' = nearest x2 (utils.py:94-107)."""
    if mode == "keep":
        return x
    N, H, W, C = x.shape
    if mode == "down":
        return x.reshape(N, H // 2, 2, W // 2, 2, C).float().mean(dim=(2, 4)).to(x.dtype)
    assert mode == "up"
    return x[:, :, None, :, None, :].expand(N, H, 2, W, 2, C).reshape(N, 2 * H, 2 * W, C)


# ---- the rest of the reference's public `edm2.utils` names (utils.py:94-107, 118-134, 153-235).  Host-side helpers on plain
# (N, C, ...) tensors for code OUTSIDE the accelerated path that imports them from here -- the reference's VAE
# (vae/vae.py:13: bmult), its VAE training scripts (cs_vae_train.py:19: GaussianLoss), its debugging aids.  None of them is
# called by the modules of this package.

def bmult(x, t):
    """x scaled per sample (t: (B,)) or per sample and channel (t: (B, C)); a 0-d t is an ordinary scalar (utils.py:153-158)."""
    if t.dim() == 0:
        return x * t
    if t.dim() not in (1, 2):
        raise ValueError("bmult: t must be 0-, 1- or 2-dimensional")
    return x * t.reshape(*t.shape, *([1] * (x.dim() - t.dim())))


def mp_cat(a, b, dim=1, t=0.5):
    """Magnitude-preserving concatenation along `dim` (utils.py:128-134); channels-last twin: mp_cat_cl."""
    Na, Nb = a.shape[dim], b.shape[dim]
    C = math.sqrt((Na + Nb) / ((1 - t) ** 2 + t ** 2))
    return torch.cat([a * (C / math.sqrt(Na) * (1 - t)), b * (C / math.sqrt(Nb) * t)], dim=dim)


def resample(x, f=[1, 1], mode="keep"):
    """(N, C, H, W): 'down' / 'up' by 2 with the separable filter f (reference utils.py:94-107: even length, normalised; the
    networks use [1, 1]: 2x2 mean / every pixel repeated 2x2).  Host-side helper on plain tensors: two 1-D depthwise passes
    (rows, then columns) -- the strided ones for 'down', the transposed ones with 2 f per axis for 'up'."""
    if mode == "keep":
        return x
    taps = torch.as_tensor([float(v) for v in f], dtype=x.dtype, device=x.device)
    if taps.numel() % 2 != 0 or taps.numel() < 2:
        raise ValueError("resample: the filter needs an even number of taps")
    taps = taps / taps.sum()
    if mode not in ("down", "up"):
        raise ValueError(f"resample: unknown mode {mode!r}")
    c, L = x.shape[1], taps.numel()
    pad = (L - 1) // 2
    col = taps.reshape(1, 1, L, 1).expand(c, 1, L, 1).contiguous()
    row = taps.reshape(1, 1, 1, L).expand(c, 1, 1, L).contiguous()
    if mode == "down":
        y = torch.nn.functional.conv2d(x, col, groups=c, stride=(2, 1), padding=(pad, 0))
        return torch.nn.functional.conv2d(y, row, groups=c, stride=(1, 2), padding=(0, pad))
    y = torch.nn.functional.conv_transpose2d(x, 2 * col, groups=c, stride=(2, 1), padding=(pad, 0))
    return torch.nn.functional.conv_transpose2d(y, 2 * row, groups=c, stride=(1, 2), padding=(0, pad))


def GaussianLoss(mean, logvar, target, eps=1e-4):
    """Mean negative log-likelihood of `target` under N(mean, exp(logvar)), with the constant 0.918 ~ ln(2 pi) / 2 the
    reference adds (utils.py:209-210; `eps` is unused there as well)."""
    nll = 0.5 * (logvar + (mean - target).square() * torch.exp(-logvar)) + 0.918
    return nll.mean()


def nan_hook(module, input, output):
    """Forward hook: raise as soon as a module emits a NaN (utils.py:165-174)."""
    outs = output if isinstance(output, (tuple, list)) else (output,)
    for o in outs:
        if torch.is_tensor(o) and torch.isnan(o).any():
            raise Exception(f"NaN detected in output of {module.__class__.__name__}")


class nan_inspector:
    """`with nan_inspector(model): ...` -- nan_hook on every submodule for the duration of the block (utils.py:177-206)."""

    def __init__(self, model):
        self.model, self.handles = model, []

    def __enter__(self):
        self.handles = [m.register_forward_hook(nan_hook) for m in self.model.modules() if m is not self.model]
        return self

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        self.handles = []
        return False


def compare_caches(cache1, cache2, rtol=1e-4, atol=1e-4, verbose=True):
    """True when two cache structures (nested dicts / lists / tuples of tensors and numbers) agree within the tolerances;
    with `verbose` the first difference is printed with its path (utils.py:214-235)."""
    def walk(a, b, path):
        if type(a) is not type(b):
            return f"{path}: type {type(a).__name__} vs {type(b).__name__}"
        if isinstance(a, dict):
            if a.keys() != b.keys():
                return f"{path}: keys {sorted(map(repr, a.keys() ^ b.keys()))} on one side only"
            for k in a:
                d = walk(a[k], b[k], f"{path}[{k!r}]")
                if d:
                    return d
            return None
        if isinstance(a, (list, tuple)):
            if len(a) != len(b):
                return f"{path}: length {len(a)} vs {len(b)}"
            for i, (u, v) in enumerate(zip(a, b)):
                d = walk(u, v, f"{path}[{i}]")
                if d:
                    return d
            return None
        if torch.is_tensor(a):
            if a.shape != b.shape:
                return f"{path}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
            if not torch.allclose(a.float(), b.float().to(a.device), rtol=rtol, atol=atol, equal_nan=True):
                return f"{path}: max |diff| {(a.float() - b.float().to(a.device)).abs().max().item():.3g}"
            return None
        if isinstance(a, float):
            return None if math.isclose(a, b, rel_tol=rtol, abs_tol=atol) else f"{path}: {a} vs {b}"
        return None if a == b else f"{path}: {a!r} vs {b!r}"
    diff = walk(cache1, cache2, "cache")
    if diff and verbose:
        print("compare_caches:", diff)
    return diff is None


def strip_runtime_state(self):
    """`__getstate__` of this package's modules (copy.deepcopy / pickle): what the package attached at run time -- handles into
    the weight bank (`pw`), cached plans and packs (`_oniris_*`), a pending batched gate (`_gate_pre`) -- stays behind; the
    copy rebuilds it at its first forward.  (The EMA trackers of the reference loops deep-copy the whole net, phema.py:95:
    without this every copy carried its own packed weights and ~6 GB of split-K slabs.)"""
    state = self.__dict__.copy()
    for k in [k for k in state if k.startswith("_oniris_") or k == "_gate_pre"]:
        del state[k]
    if "pw" in state:
        state["pw"] = None
    return state


class MPFourier(nn.Module):
    def __init__(self, num_channels, bandwidth=1):
        super().__init__()
        self.register_buffer("freqs", 2 * np.pi * torch.randn(num_channels) * bandwidth)
        self.register_buffer("phases", 2 * np.pi * torch.rand(num_channels))

    def forward(self, x):
        y = torch.outer(x.float(), self.freqs.float()) + self.phases.float()
        return (y.cos() * math.sqrt(2)).to(x.dtype)


def _s3_split(url):
    """'s3://bucket/some/key' -> ('bucket', 'some/key')."""
    rest = url[len("s3://"):]
    bucket, _, key = rest.partition("/")
    if not bucket or not key:
        raise ValueError(f"not an s3://bucket/key URL: {url!r}")
    return bucket, key


def _s3_client():
    try:
        import boto3
    except ImportError as e:                      # (the reference imports boto3 unconditionally: utils.py:16,39)
        raise ImportError("s3:// checkpoints need the `boto3` package (as in the reference, edm2/utils.py:16-31,39-58); "
                          "install it or pass a local path") from e
    return boto3.client("s3")


def _numpy_safe_globals():
    """The numpy reconstructors a checkpoint's `kwargs` may need: scalars (np.float64 mean / std of a VAE, np.int64 sizes),
    small ndarrays and their dtype objects.  Data-only callables: allow-listing them does not open the pickle machinery."""
    import numpy as np
    core = getattr(np, "_core", None) or np.core
    allow = [core.multiarray.scalar, core.multiarray._reconstruct, np.dtype, np.ndarray]
    dtypes = getattr(np, "dtypes", None)
    if dtypes is not None:
        allow += [getattr(dtypes, n) for n in dir(dtypes) if n.endswith("DType")]
    return allow


def _load_checkpoint_file(path):
    """{"state_dict": tensors, "kwargs": constructor arguments} (reference utils.py:15-64).  The reference loads with
    weights_only=False (utils.py:59: arbitrary code execution from a downloaded file); nothing in the format needs that, so:
    (1) the restricted unpickler; (2) if a constructor argument is a numpy scalar / array (statistics passed to a VAE's
    constructor end up in `kwargs` as they are), the restricted unpickler again with numpy's data reconstructors allow-listed;
    (3) anything else is refused with a message that names the opt-in: ONIRIS_TRUST_CHECKPOINT=1 loads like the reference."""
    import os
    import pickle
    if os.environ.get("ONIRIS_TRUST_CHECKPOINT") == "1":
        return torch.load(path, weights_only=False)
    try:
        return torch.load(path, weights_only=True)
    except pickle.UnpicklingError as first:
        try:
            with torch.serialization.safe_globals(_numpy_safe_globals()):
                return torch.load(path, weights_only=True)
        except pickle.UnpicklingError:
            raise pickle.UnpicklingError(
                f"{path}: the checkpoint holds objects beyond tensors, plain Python values and numpy scalars / arrays, which the "
                "restricted loader refuses.  If you trust the file, set ONIRIS_TRUST_CHECKPOINT=1 to load it the way the reference "
                f"does (torch.load(weights_only=False), edm2/utils.py:59).  First refusal: {str(first).splitlines()[0]}") from first


class BetterModule(nn.Module):
    """save_to_state_dict / from_pretrained with the reference's {"state_dict", "kwargs"} file format (utils.py:13-72): local
    paths, or s3://bucket/key URLs through boto3 like the reference (uploaded from a temporary file; downloads are kept under
    /tmp/cache/autoregressive_diffusion_models/ and reused).  The reference's VAE inherits this class through the `edm2` shim
    (vae/vae.py:13), and its scripts load both models from S3 (gym_train.py:33, generation_code.py:30,34)."""

    S3_CACHE_DIR = "/tmp/cache/autoregressive_diffusion_models/"

    def save_to_state_dict(self, path):
        data = {"state_dict": self.state_dict(), "kwargs": self.kwargs}
        path = str(path)
        if not path.startswith("s3://"):
            torch.save(data, path)
            return
        import os
        import tempfile
        bucket, key = _s3_split(path)
        client = _s3_client()
        fd, tmp = tempfile.mkstemp(suffix=".pt")
        os.close(fd)
        try:
            torch.save(data, tmp)
            client.upload_file(tmp, bucket, key)
        finally:
            os.remove(tmp)

    @classmethod
    def from_pretrained(cls, checkpoint):
        if isinstance(checkpoint, str):
            if checkpoint.startswith("s3://"):
                import os
                bucket, key = _s3_split(checkpoint)
                os.makedirs(cls.S3_CACHE_DIR, exist_ok=True)
                local = os.path.join(cls.S3_CACHE_DIR, os.path.basename(key))
                if not os.path.exists(local):
                    _s3_client().download_file(bucket, key, local)
                checkpoint = local
            checkpoint = _load_checkpoint_file(checkpoint)
        model = cls(**checkpoint["kwargs"])
        model.load_state_dict(checkpoint["state_dict"])
        return model

    @property
    def _ddp_params_and_buffers_to_ignore(self):
        """torch.nn.parallel.DistributedDataParallel reads exactly this attribute of the module it is handed (its
        constructor, torch/nn/parallel/distributed.py: `parameters_to_ignore`; nothing else does) -- the one place where
        `DDP(unet, device_ids=[local_rank], output_device=local_rank, find_unused_parameters=True)` (cs_train.py:53-54) can
        be noticed.  torch's reducer learns about a gradient from the parameter's AccumulateGrad node; the conv / attention
        / embedding weight gradients of this net are written into `.grad` through raw pointers by ONE kernel after the
        backward pass (weights.hip: weight_bwd_kernel), so the reducer would never see them.  The two kinds of parameters
        are therefore split:
          * the kernel-owned weights (every NormalizedWeight) are named here -- torch's reducer and its construction-time
            broadcast leave them alone -- and an inner `parallel.OnirisDDP` on the SAME process group takes them: re-homed
            in one flat buffer (broadcast from rank 0 right here), exchanged by the staged tensor hooks during backward
            and at its end, honouring `no_sync()` through torch DDP's own `require_backward_grad_sync`;
          * everything autograd accumulates itself (gates, emb_gain, out_gain) stays an ordinary torch parameter that
            torch's reducer buckets and averages as for any model.
        A torch.optim optimizer over `parameters()` works on both (OnirisDDP's foreign-optimizer cycle).
        Every OTHER reader (inspect.getmembers, hasattr, debuggers, attribute-copying wrappers) sees a plain missing
        attribute: the split happens only when the caller is torch's distributed.py."""
        import sys
        frame = sys._getframe(1)
        if not frame.f_code.co_filename.replace("\\", "/").endswith("torch/nn/parallel/distributed.py"):
            raise AttributeError("_ddp_params_and_buffers_to_ignore")
        # ... and only when this module really holds kernel-written weights: the reference's VAE is a BetterModule too
        # (vae/vae.py:13 takes the class from here) and is an ordinary torch model that torch DDP handles alone
        from .conv import NormalizedWeight
        owned = [m.weight for m in self.modules() if isinstance(m, NormalizedWeight) and m.weight.requires_grad]
        if not owned:
            raise AttributeError("_ddp_params_and_buffers_to_ignore")
        # ... and with them every parameter whose gradient this package's fused passes deliver (the gates of the gated convs and
        # emb_gain: gathered / delivered as packs of the flat buffers; out_gain: applied by the fused loss OUTSIDE the wrapper's
        # forward, where torch's reducer would mark it unused and then see its gradient arrive).  torch's reducer keeps what is left
        # -- for the UNet the four parameters of `out_res`, which the reference evaluates and never uses (networks_edm2.py:197) --
        # and it must keep at least one parameter: otherwise only the kernel-owned weights are taken.
        takers = self._ddp_fused_parameters() if hasattr(self, "_ddp_fused_parameters") else []
        rest = {id(p) for p in owned} | {id(p) for p in takers}
        if any(p.requires_grad and id(p) not in rest for p in self.parameters()):
            owned = owned + [p for p in takers if p.requires_grad]
        outer = frame.f_locals.get("self")
        inner = self.__dict__.get("_oniris_inner_ddp")
        if inner is None or isinstance(inner, str) or inner._torch_ddp is None or inner._torch_ddp() is not outer:
            # (torch reads the attribute twice -- hasattr, then the value: built on the first read)
            from ..parallel import FlatParams, OnirisDDP
            try:
                inner = OnirisDDP(self, process_group=getattr(outer, "process_group", None),
                                  flat=FlatParams(self, only=owned), torch_ddp=outer)
            except AttributeError as e:      # (torch asks with hasattr(): an AttributeError from in here would read as "no
                raise RuntimeError(f"building the gradient exchange for torch DDP failed: {e!r}") from e   # such attribute")
            self.__dict__["_oniris_inner_ddp"] = inner
        ids = {id(p) for p in inner.flat.params}
        names = [n for n, p in self.named_parameters() if id(p) in ids]
        # The buffers of these nets are constants (MPFourier's frequencies / phases, the rotary embedding's inv_freq / scale):
        # the inner engine broadcast rank 0's once at construction, as OnirisDDP does.  torch DDP would broadcast them again
        # in EVERY forward (broadcast_buffers=True is its default and what cs_train.py:54 gets) -- an in-place rewrite that
        # bumps their version counters, on which the rotary tables are cached (ops.rope_tables): 8 table rebuilds with host
        # round trips per step, ~20 ms of the 84 ms step measured under the wrapper (profiles/r05_torchloop.txt).
        names += [n for n, b in self.named_buffers() if b is not None]
        # (torch's reducer tests f"{module_name}.{param_name}" against this list, which for a parameter of the ROOT module is
        # ".name", while its broadcast and its parameter filter test the plain "name": a root-level parameter goes in twice)
        return names + ["." + n for n in names if "." not in n]

    def _ddp_inner(self):
        """The inner OnirisDDP when torch's DistributedDataParallel wraps this module (None otherwise).  A COPY of a wrapped
        module (copy.deepcopy: the EMA trackers of the reference loops, phema.py:95) has no exchange of its own: fine for the
        evaluations such copies are made for, an error as soon as it is trained on more than one rank."""
        inner = self.__dict__.get("_oniris_inner_ddp")
        if inner is None:
            return None
        if isinstance(inner, str):
            import torch.distributed as dist
            if (torch.is_grad_enabled() and self.training and dist.is_available() and dist.is_initialized()
                    and dist.get_world_size() > 1):
                raise RuntimeError("this network is a copy of one that torch's DistributedDataParallel wrapped: the copy has "
                                   "no gradient exchange for its kernel-owned weights -- wrap it in DistributedDataParallel "
                                   "itself before training it")
            return None
        return inner

    def __getstate__(self):
        """copy.deepcopy / pickle: everything this package attached at run time (`_oniris_*`: weight bank with its packed
        buffers and split-K slabs, cached plans, stage hooks, the inner DDP engine with its process group) stays behind --
        the copy rebuilds what it needs at its first forward."""
        lost = self.__dict__.get("_oniris_inner_ddp") is not None
        state = strip_runtime_state(self)
        if lost:
            state["_oniris_inner_ddp"] = "lost"
        return state

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def n_params(self):
        return sum(p.numel() for p in self.parameters())
