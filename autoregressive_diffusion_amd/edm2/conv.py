"""MPConv / MPCausal3DGatedConv / Gating / NormalizedWeight with the reference's module tree, parameter names and
call signatures (edm2/conv.py:8-127), computing through the HIP implicit-GEMM kernels.

Public `forward`s take/return NCHW tensors like the reference; `_cl` variants work on channels-last bf16 and are
what UNet uses internally.  Weights are normalised + packed ONCE per top-level forward by the WeightBank of the
outermost module (see `weights_ready`)."""
import threading
import torch
from torch import nn

from .. import ops
from .. import fp32 as _fp32
from .utils import to_cl, from_cl, BF16, strip_runtime_state

_tls = threading.local()


def _bank_of(root):
    bank = root.__dict__.get("_oniris_bank")
    mods = root.__dict__.get("_oniris_mods")          # the module tree of a built net is fixed: walk it once
    if mods is None:                                   # (~0.5 ms of host time per forward for the gym net otherwise)
        mods = root.__dict__["_oniris_mods"] = [m for m in root.modules() if isinstance(m, NormalizedWeight)]
    if bank is None or bank._n_mods != len(mods) or any(m.pw is None or m.pw.param is not m._parameters["weight"] or
                                                        m.pw.bank is not bank for m in mods):
        bank = ops.WeightBank()
        pairs = {id(w) for c in root.modules() if isinstance(c, MPCausal3DGatedConv) for w in (c.last_frame_conv.weight, c.weight)}
        for m in mods:
            m.pw = bank.add(m.weight, perm3=m.perm3, gated_pair=id(m) in pairs)
            m.pw.bank = bank
        bank._n_mods = len(mods)
        # 1x1 weights that read the same input (UNet: the emb_linear of every Block) become one row-concatenated GEMM
        for sub in root.modules():
            if hasattr(sub, "_oniris_weight_groups"):
                sub.__dict__["_oniris_groups"] = [bank.add_group([m.pw for m in grp]) for grp in sub._oniris_weight_groups()]
        root.__dict__["_oniris_bank"] = bank
    return bank


class weights_ready:
    """Context manager: the OUTERMOST module forward normalises+packs all of its weights (one kernel launch);
    nested forwards reuse them.  Mirrors 'NormalizedWeight.forward runs on every use' (conv.py:14-21) at the
    granularity the fixed point of the forced normalisation allows (once per step)."""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        depth = getattr(_tls, "depth", 0)
        if depth == 0:
            _bank_of(self.module).prepare(self.module.training)
        _tls.depth = depth + 1

    def __exit__(self, *exc):
        _tls.depth -= 1
        return False


class NormalizedWeight(nn.Module):
    def __init__(self, in_channels, out_channels, kernel):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, *kernel))
        self.perm3 = False
        self.pw = None

    __getstate__ = strip_runtime_state

    def forward(self, gain=1):
        """The effective weight as a tensor, for code that asks the module itself (reference conv.py:14-21: forced normalisation
        of the stored parameter in training mode, then normalize(w) * gain / sqrt(fan_in), gradients through the second
        normalisation).  Plain torch on whatever device the parameter lives on; the convolutions of this package never call it --
        they read the packed bf16 copies `weights_ready` prepares for all weights in one launch."""
        from .utils import normalize
        w = self.weight.to(torch.float32)
        if self.training:
            with torch.no_grad():
                self.weight.copy_(normalize(w))
        fan_in = w[0].numel()
        return normalize(w) * (gain / fan_in ** 0.5)


class MPConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel, dilation=1):
        super().__init__()
        self.out_channels = out_channels
        self.weight = NormalizedWeight(in_channels, out_channels, kernel)
        assert dilation == 1

    def _cl(self, x, res=None, ta=0.0, tb=0.0, clip=0.0, in_slot=None, res_slot=None):
        return ops.conv(x, self.weight.pw, res, ta, tb, clip, in_slot=in_slot, res_slot=res_slot)

    def forward(self, x, gain=1):
        if _fp32.active():                                      # Precond(use_fp16=False) / force_fp32=True: fp32 end to end
            return _fp32.mpconv(self, x, gain)
        with weights_ready(self):
            if x.ndim == 2:                                     # linear (conv.py:38-39)
                pad = (-x.shape[1]) % 8
                xin = torch.nn.functional.pad(x, (0, pad)).to(BF16)[:, None, None, :].contiguous()
                y = self._cl(xin)[:, 0, 0, :].to(x.dtype)
            else:
                y = from_cl(self._cl(to_cl(x, pad_to=-(-x.shape[1] // 8) * 8)), x.dtype)
                if self.weight.perm3:          # attn_qkv: the packed rows are (s m c); the public output keeps the
                    n, c3, h, w = y.shape      # reference's channel order (m c s) (attention_modules.py:48)
                    y = y.reshape(n, 3, c3 // 3, h, w).transpose(1, 2).reshape(n, c3, h, w)
            return y * gain

    @torch.no_grad()
    def load_from_2d(self, state_dict):
        self.weight.weight.copy_(state_dict)


DEFER_CACHE_SHIFT = [False]         # set by edm2/sampler.py around the capture of a frame's evaluation graph


class Gating(nn.Module):
    """edm2/conv.py:104-127 (tiny, fp32, stays in torch autograd)."""

    def __init__(self):
        super().__init__()
        self.offset = nn.Parameter(torch.tensor([0., 0.]))
        self.mult = nn.Parameter(torch.tensor([1.5, -0.5]))
        self.max_gating = nn.Parameter(torch.tensor(-5.))
        self.min_gating = nn.Parameter(torch.tensor(-5.))

    def forward(self, c_noise, n_context_frames=0, just_2d=False):
        B, tt = c_noise.shape
        T = tt // 2 if self.training else tt
        if just_2d:
            pos = torch.zeros_like(c_noise)
        else:
            pos = (torch.arange(B * tt, device=c_noise.device) % T).reshape(B, tt) + n_context_frames
            pos = pos.to(c_noise.dtype).log1p()
        sv = c_noise * self.mult[0] + self.offset[0] + pos * self.mult[1] + self.offset[1]
        lo, hi = torch.sigmoid(self.min_gating), torch.sigmoid(self.max_gating)
        return lo + (1 - lo) * hi * torch.sigmoid(sv), n_context_frames + T


_nctx_cache = {}


def nctx_tensor(n_ctx, dev):
    """Per-layer frame counters as a device vector: one host->device copy per distinct frame count (the sampler
    evaluates the net 31 times per frame; cached so that the evaluation is hipGraph-capturable)."""
    key = (tuple(n_ctx), str(dev))
    if key not in _nctx_cache:
        prev = _nctx_cache.get((tuple(n - 1 for n in n_ctx), str(dev)))
        if len(_nctx_cache) > 64:
            _nctx_cache.clear()
        # a rollout bumps every counter by one per generated frame: derive the vector on the device (an upload from
        # pageable memory would wait for everything queued on the stream -- the previous frame's 31 evaluations)
        _nctx_cache[key] = prev + 1 if prev is not None else torch.tensor(list(n_ctx), device=dev).reshape(-1, 1, 1)
    return _nctx_cache[key]


_gate_param_cache = {}


def _packed_gate_params(convs, dev):
    """(L, 6) fp32 = mult0, mult1, off0, off1, min_gating, max_gating of every gating layer; cached until a parameter
    changes (torch writes bump ._version; the raw-pointer optimizer bumps ops._weights_epoch)."""
    ps = [p for m in convs for p in (m.gating.mult, m.gating.offset, m.gating.min_gating, m.gating.max_gating)]
    key = (id(convs[0]), len(convs), str(dev))
    sig = (ops._weights_epoch, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps))
    hit = _gate_param_cache.get(key)
    if hit is None or hit[0] != sig:
        with torch.no_grad():
            packed = torch.stack([torch.cat([m.gating.mult.reshape(2), m.gating.offset.reshape(2), m.gating.min_gating.reshape(1),
                                             m.gating.max_gating.reshape(1)]) for m in convs]).float().contiguous().to(dev)
        if len(_gate_param_cache) > 16:
            _gate_param_cache.clear()
        hit = _gate_param_cache[key] = (sig, packed)
    return hit[1]


_nctx_i32_cache = {}


def _nctx_i32(n_ctx, dev):
    key = (tuple(n_ctx), str(dev))
    if key not in _nctx_i32_cache:
        prev = _nctx_i32_cache.get((tuple(n - 1 for n in n_ctx), str(dev)))
        if len(_nctx_i32_cache) > 64:
            _nctx_i32_cache.clear()
        _nctx_i32_cache[key] = prev + 1 if prev is not None else torch.tensor(list(n_ctx), dtype=torch.int32, device=dev)
    return _nctx_i32_cache[key]


def batched_gates(convs, c_noise, caches, training):
    """All Gating modules of a net in ONE vectorised evaluation (identical math to Gating.forward, conv.py:113-127):
    replaces ~60 x 20 tiny elementwise launches per step by ~20.  Returns per-layer (ca, cb, n_new)."""
    B, tt = c_noise.shape
    T = tt // 2 if training else tt
    dev = c_noise.device
    n_ctx = [int(c.get("n_context_frames", 0)) if c else 0 for c in caches]
    if not training and not torch.is_grad_enabled() and c_noise.is_cuda:
        # eval: one HIP launch (oniris_gates) instead of ~20 torch ones; the packed gating parameters are rebuilt only
        # when a parameter changed, the frame counters are the cached device vector of nctx_tensor
        params = _packed_gate_params(convs, dev)
        ca, cb = ops.gates_eval(c_noise.float().contiguous(), params, _nctx_i32(n_ctx, dev) if any(n_ctx) else None, T)
        return [(a, b, n + T) for a, b, n in zip(ca.unbind(0), cb.unbind(0), n_ctx)]
    if ops.FUSED_PRELUDE and c_noise.is_cuda:
        # one forward and one backward launch (oniris_gates / oniris_gates_bwd) around an (L, 6) pack of the parameters;
        # the torch formulation below costs ~28 forward and ~60 backward launches on (L, B, tt) = 15 K-element tensors
        pack = None
        if torch.is_grad_enabled():        # parameters in one FlatParams: gathered from / delivered into its flat buffers
            gp = convs[0].__dict__.get("_oniris_gate_params")      # (module attribute lookups: ~1 us each, 300 of them)
            if gp is None or len(gp) != 4 * len(convs) or gp[0] is not convs[0].gating._parameters["mult"]:
                gp = convs[0].__dict__["_oniris_gate_params"] = [p for m in convs for p in (
                    m.gating.mult, m.gating.offset, m.gating.min_gating, m.gating.max_gating)]
            pack = ops.direct_pack(gp, convs[0], "_gate_pack")
        if pack is not None:
            P, anchor = pack.values().view(len(convs), 6), convs[0].gating.mult
        else:
            P = torch.cat([torch.stack([m.gating.mult for m in convs]), torch.stack([m.gating.offset for m in convs]),
                           torch.stack([m.gating.min_gating for m in convs])[:, None],
                           torch.stack([m.gating.max_gating for m in convs])[:, None]], dim=1).float()
            anchor = None
        ca, cb = ops.gates_train(c_noise.float().reshape(-1).contiguous(), P, _nctx_i32(n_ctx, dev) if any(n_ctx) else None, T,
                                 pack, anchor)
        return [(a, b, n + T) for a, b, n in zip(ca.unbind(0), cb.unbind(0), n_ctx)]
    return ops._prelude_ref("batched_gates")(convs, c_noise, caches, training, n_ctx, T, nctx_tensor)


class MPCausal3DGatedConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel):
        super().__init__()
        assert len(kernel) == 3
        self.out_channels = out_channels
        self.in_channels = in_channels
        self.last_frame_conv = MPConv(in_channels, out_channels, kernel[1:])
        self.weight = NormalizedWeight(in_channels, out_channels, (kernel[0] - 1, kernel[1], kernel[2]))
        self.gating = Gating()

    __getstate__ = strip_runtime_state

    def _cl(self, x, batch_size, c_noise, cache=None, update_cache=False, just_2d=False, **epi):
        """x (B*t, H, W, C) bf16 -> (y, cache).  cache['activations'] is (B, 2, H, W, C) bf16.
        **epi: fused epilogue, either cscale=(N,Cout) fp32 [silu(y*cscale)/0.596] or res/ta/tb/clip [mp_sum+clip];
        grad_private (training, see ops.ConvCfg): the output's gradient will be a tensor only this op's backward reads."""
        grad_private = epi.pop("grad_private", False)
        slot_kw = dict(res_slot=epi.pop("res_slot", None), res_alias=epi.pop("res_alias", False))
        if just_2d:
            self.__dict__.pop("_gate_pre", None)
            train_kw = dict(grad_private=grad_private, **slot_kw) if self.training else {}
            return ops.conv(x, self.last_frame_conv.weight.pw, **epi, **train_kw), cache
        if cache is None:
            cache = {}
        pre = self.__dict__.pop("_gate_pre", None)            # (ca, cb, n_new) batched by UNet.forward for all layers
        if pre is not None:
            coefs, gate, n_new = (pre[0], pre[1]), None, pre[2]
        else:
            gate, n_new = self.gating(c_noise.float(), cache.get("n_context_frames", 0))
            gate, coefs = gate.reshape(-1), None
        if update_cache:
            cache["n_context_frames"] = n_new
        N, H, W, C = x.shape
        pw2, pw3 = self.last_frame_conv.weight.pw, self.weight.pw
        if self.training:
            T = N // (2 * batch_size)
            return ops.gated_conv_train(x, gate, pw2, pw3, batch_size, T, coefs, grad_private=grad_private, **slot_kw, **epi), cache
        t = N // batch_size
        pad = cache.get("activations")
        had_pair = pad is not None
        if pad is None:
            pad = torch.ones(batch_size, 2, H, W, C, dtype=BF16, device=x.device)
        if t == 1:              # one generated frame (the sampler's 31 evaluations): its context IS the cached pair
            if update_cache and DEFER_CACHE_SHIFT[0]:
                # captured evaluation of the sampler's frame graph: the shifted pair is built once, after the frame's last
                # replay (sampler.finish_cache), from the old pair and this layer's input -- not by every replay
                cache["activations"] = pad
                cache["_pending_frame"] = x.reshape(batch_size, 1, H, W, C)
            elif update_cache:
                cache["activations"] = torch.cat([pad[:, 1:], x.reshape(batch_size, 1, H, W, C)], dim=1)
            # The context product of the cached pair is the same in every evaluation against that pair (the sampler runs 31
            # per generated frame, sampler.py:50-76; the reference's F.conv3d recomputes it each time, conv.py:84-86): kept
            # in the cache entry beside the pair it belongs to -- computed by UNet.prewarm_eval or by the first evaluation,
            # read by all later ones, dropped (by identity of the pair) as soon as the cache moves on.
            # (... or the weights do: the third entry is the packed-weight signature object of the bank, replaced by every
            # re-packing -- WeightBank.prepare)
            kept = cache.get("_ctx_product") if had_pair else None
            wsig = getattr(pw3.bank, "_packed_sig", None)
            if kept is not None and kept[1] is pad and kept[2] is wsig and wsig is not None:
                epi = dict(epi, ctx_prod=kept[0], ctx_prod_mode=2)
            elif (had_pair and wsig is not None and pad.is_contiguous() and ops.ctx_product_ok(H, W, C, pw2.cout)
                  and not torch.cuda.is_current_stream_capturing()):
                y3 = torch.empty((batch_size, H, W, ops.roundup(pw2.cout, 8)), dtype=torch.float32, device=x.device)
                cache["_ctx_product"] = (y3, pad, wsig)
                epi = dict(epi, ctx_prod=y3, ctx_prod_mode=1)
            return ops.gated_conv_eval(x, gate, pw2, pw3, batch_size, 1, pad.contiguous(), coefs, ctx_T=2, **epi), cache
        ctx = torch.cat([pad, x.reshape(batch_size, t, H, W, C)], dim=1).contiguous()
        if update_cache:
            cache["activations"] = ctx[:, -2:].clone()
        return ops.gated_conv_eval(x, gate, pw2, pw3, batch_size, t, ctx, coefs, **epi), cache

    def keep_ctx_product(self, cache):
        """Compute the context product of `cache`'s pair now (one context-phases-only launch), so that the evaluations that
        follow -- captured into a hipGraph or not -- only walk their own phases.  No-op when it is already there."""
        pad = cache.get("activations") if cache else None
        if pad is None or not pad.is_contiguous() or pad.dtype != BF16:
            return
        pw2, pw3 = self.last_frame_conv.weight.pw, self.weight.pw
        wsig = getattr(getattr(pw3, "bank", None), "_packed_sig", None)
        if wsig is None:
            return
        kept = cache.get("_ctx_product")
        if kept is not None and kept[1] is pad and kept[2] is wsig:
            return
        B, _, H, W, C = pad.shape
        if not ops.ctx_product_ok(H, W, C, pw2.cout):
            return
        cache["_ctx_product"] = (ops.gated_conv_ctx_product(pad, pw2, pw3, B), pad, wsig)

    def forward(self, x, emb, batch_size, c_noise, cache=None, update_cache=False, just_2d=False):
        if _fp32.active():
            return _fp32.gated_conv(self, x, emb, batch_size, c_noise, cache, update_cache, just_2d)
        with weights_ready(self):
            cl = to_cl(x, pad_to=-(-x.shape[1] // 8) * 8)
            if cache is not None and "activations" in cache and cache["activations"].shape[-1] != cl.shape[-1]:
                raise ValueError("cache does not belong to this layer")
            y, cache = self._cl(cl, batch_size, c_noise, cache, update_cache, just_2d)
            return from_cl(y, x.dtype), cache

    @torch.no_grad()
    def load_from_2d(self, weight):
        if isinstance(weight, dict):
            weight = weight["weight"]
        self.last_frame_conv.load_from_2d(weight)
