"""Block / UNet / Precond with the reference's constructor arguments, forward signatures, module tree and
state_dict keys (edm2/networks_edm2.py:19-297), running on the MI355X kernels: activations stay channels-last
bf16 between layers, every conv / attention is a HIP kernel behind the C-ABI, weights are normalised and packed
once per forward."""
import inspect
import math
from contextlib import nullcontext
import torch
from torch import nn
import torch.nn.functional as F

from .. import ops
from .. import fp32 as _fp32
from .loss_weight import MultiNoiseLoss
from .utils import BetterModule, MPFourier, to_cl, from_cl, BF16, strip_runtime_state
from .conv import MPConv, MPCausal3DGatedConv, Gating, weights_ready, batched_gates
from .attention import FrameAttention, VideoAttention


def unwrap_ddp(m):
    """The module behind torch's DistributedDataParallel (cs_train.py:54 hands the WRAPPER to Precond); anything else as it is."""
    return m.module if isinstance(m, torch.nn.parallel.DistributedDataParallel) else m


class Block(nn.Module):
    def __init__(self, in_channels, out_channels, emb_channels, flavor="enc", resample_mode="keep",
                 resample_filter=[1, 1], attention=False, channels_per_head=64, dropout=0, res_balance=0.3,
                 attn_balance=0.3, clip_act=256):
        super().__init__()
        # [1, 1] (every configuration of the reference, :26) runs inside the fused activation kernel; any other even-length filter
        # (utils.py:94-107) goes through its own pass, oniris_resample_filter
        self._taps = ops.resample_taps(resample_filter)
        self.out_channels, self.in_channels = out_channels, in_channels
        self.flavor, self.resample_filter, self.resample_mode = flavor, resample_filter, resample_mode
        self.num_heads = out_channels // channels_per_head if attention else 0
        self.dropout, self.res_balance, self.attn_balance, self.clip_act = dropout, res_balance, attn_balance, clip_act
        self.emb_gain = nn.Parameter(torch.zeros([]))
        self.emb_linear = MPConv(emb_channels, out_channels, kernel=[])
        self.conv_res0 = MPCausal3DGatedConv(out_channels if flavor == "enc" else in_channels, out_channels, [3, 3, 3])
        self.conv_res1 = MPCausal3DGatedConv(out_channels, out_channels, [3, 3, 3])
        self.conv_skip = MPConv(in_channels, out_channels, kernel=[1, 1]) if in_channels != out_channels else None
        if attention == "video":
            self.attn = VideoAttention(out_channels, self.num_heads, attn_balance)
        else:
            self.attn = FrameAttention(out_channels, self.num_heads, attn_balance)

    __getstate__ = strip_runtime_state

    def _cl(self, x, emb, batch_size, c_noise, cache=None, update_cache=False, just_2d=False, skip=None, cat_w=None,
            c=None, in_slot=None, skip_slot=None, private_out=False):
        """x (N,H,W,C) bf16, emb (N,1,1,cemb) bf16; skip/cat_w: the decoder's mp_cat operand, fused into the first
        activation kernel.  private_out: the caller consumes the block's output through ops of this package that join all of
        its gradients in one kernel (UNet.forward with GradSlots) -- see ops.ConvCfg.grad_private.  Elementwise chains of the reference's Block.forward (:62-94) run as fused HIP kernels:
        [mp_cat | pixel norm] + mp_silu -> act;  *c + mp_silu -> conv_res0 epilogue;  mp_sum (+clip) -> conv_res1 /
        attn_proj epilogue."""
        if cache is None:
            cache = {}
        # in_slot / skip_slot (ops.GradSlot): x / skip is an encoder output with a second consumer; the block's first op
        # adds the other consumer's gradient inside its own backward kernel, the mp_cat parks the skip gradient for it
        # the resampling (reference :63) runs inside the block's first activation kernel when that is the next op
        rs = self.resample_mode
        if rs != "keep" and (self._taps is not None or skip is not None or (self.flavor == "enc" and self.conv_skip is not None)):
            x, in_slot, rs = ops.resample(x, rs, in_slot, self._taps), None, "keep"
        # xs: the activation's xo is the residual of conv_res1 (no 1x1 skip conv in between): that conv parks the residual gradient
        # as (gradient of its output, ta) and the activation's backward kernel applies the factor -- ta * g is never written
        skip_conv_done = False
        want_xs = (private_out and ops.ALIAS2 and self.training and torch.is_grad_enabled()
                   and not (self.flavor == "dec" and self.conv_skip is not None))
        if self.flavor == "enc" and self.conv_skip is not None:
            x, in_slot = self.conv_skip._cl(x, in_slot=in_slot), None
        # ... only when the activation's backward will RUN and take what was parked: an input that carries no gradient (a frozen
        # stem / frozen first encoder blocks while fine-tuning) has no backward node, and a parked gradient would be left over at
        # the end of backward (GradSlot.check_all_taken would blame a partial backward) -- then conv_res1 returns the
        # residual gradient to autograd, which drops it
        xs = ops.GradSlot() if (want_xs and (x.requires_grad or (skip is not None and skip.requires_grad))) else None
        if self.flavor == "enc":
            x, a = ops.act(x, norm=True, in_slot=in_slot, resample=rs, xo_slot=xs)     # x <- pixel norm(x); a = mp_silu(x)
        elif (skip is not None and self.conv_skip is not None and not self.training
              and ops.conv_cat_act_ok(x, skip, self.conv_skip.weight.pw)):
            # evaluation (one generated frame: every launch is ~5 us of latency around almost no work): mp_cat, mp_silu and the
            # 1x1 skip conv of the concatenation in ONE launch; the concatenated tensor itself is never written
            x, a = ops.conv_cat_act(x, skip, cat_w[0], cat_w[1], self.conv_skip.weight.pw)
            skip_conv_done = True
        elif skip is not None:
            x, a = ops.act(x, skip, cat_w[0], cat_w[1], want_xo=True, in_slot=in_slot, skip_slot=skip_slot, xo_slot=xs)   # x <- mp_cat(x, skip); a = mp_silu(x)
        else:
            x, a = ops.act(x, want_xo=True, in_slot=in_slot, resample=rs, xo_slot=xs)  # (no resampling: x comes back as an alias of itself)
        N = x.shape[0]
        if c is None:          # (the UNet hands in all of its blocks' scales from one grouped GEMM: ops.emb_scales)
            c = (self.emb_linear._cl(emb).reshape(N, -1).float() * self.emb_gain + 1)      # (N, Cout) fp32
        y, cache["conv_res0"] = self.conv_res0._cl(a, batch_size, c_noise, cache.get("conv_res0"), update_cache,
                                                   just_2d, cscale=c)    # y = mp_silu(conv(a) * c)
        if self.training and self.dropout != 0:
            y = F.dropout(y, p=self.dropout)
        if self.flavor == "dec" and self.conv_skip is not None and not skip_conv_done:
            x = self.conv_skip._cl(x)
        t = self.res_balance
        den = 1.0 / math.sqrt((1 - t) ** 2 + t ** 2)
        clip = float(self.clip_act) if self.clip_act is not None else 0.0
        x, cache["conv_res1"] = self.conv_res1._cl(y, batch_size, c_noise, cache.get("conv_res1"), update_cache, just_2d,
                                                   res=x, ta=(1 - t) * den, tb=t * den,
                                                   clip=clip if self.num_heads == 0 else 0.0,
                                                   grad_private=private_out,
                                                   **(dict(res_slot=xs, res_alias=True) if xs is not None else {}))
        if self.num_heads > 0:
            x, cache["attn"] = self.attn._cl(x, batch_size, cache.get("attn"), update_cache, just_2d, clip=clip)
        else:
            cache["attn"] = None
        return x, cache

    @torch.no_grad()
    def load_from_2d(self, state_dict):
        """state_dict of the matching 2-D EDM2 block (keys `conv_res0.weight`, `emb_linear.weight`, `conv_res1.weight`,
        `emb_gain` [, `conv_skip.weight`, `attn_qkv.weight`, `attn_proj.weight`]); reference networks_edm2.py:96-110."""
        sd = {(k[:-len(".weight")] if k.endswith(".weight") else k): v for k, v in state_dict.items()}
        if "attn_qkv" in sd:
            self.attn.attn_qkv.weight.weight.copy_(sd["attn_qkv"])
            self.attn.attn_proj.weight.weight.copy_(sd["attn_proj"])
        if "emb_gain" in sd:
            self.emb_gain.copy_(sd["emb_gain"])
        for name, child in self.named_children():
            if callable(getattr(child, "load_from_2d", None)):
                child.load_from_2d(sd[name])

    def forward(self, x, emb, batch_size, c_noise, cache=None, update_cache=False, just_2d=False):
        if _fp32.active():
            return _fp32.block(self, x, emb, batch_size, c_noise, cache, update_cache, just_2d)
        with weights_ready(self):
            pad = (-emb.shape[1]) % 8
            e = F.pad(emb, (0, pad)).to(BF16)[:, None, None, :].contiguous()
            y, cache = self._cl(to_cl(x), e, batch_size, c_noise, cache, update_cache, just_2d)
            return from_cl(y, x.dtype), cache


class UNet(BetterModule):
    def __init__(self, img_resolution, img_channels, label_dim, model_channels, channel_mult=[1, 2, 2, 4],
                 channel_mult_noise=None, channel_mult_emb=None, num_blocks=3, video_attn_resolutions=[8],
                 frame_attn_resolutions=[16], label_balance=0.5, concat_balance=0.5, **block_kwargs):
        super().__init__()
        self.img_resolution, self.img_channels, self.label_dim = img_resolution, img_channels, label_dim
        cblock = [model_channels * x for x in channel_mult]
        cnoise = model_channels * channel_mult_noise if channel_mult_noise is not None else cblock[0]
        cemb = model_channels * channel_mult_emb if channel_mult_emb is not None else max(cblock)
        self.label_balance, self.concat_balance = label_balance, concat_balance
        self.out_res = Gating()
        self.out_gain = nn.Parameter(torch.zeros([]))
        self.emb_fourier_sigma = MPFourier(cnoise)
        self.emb_noise = MPConv(cnoise, cemb, kernel=[])
        self.emb_fourier_time = MPFourier(cnoise)
        self.emb_time = MPConv(cnoise, cemb, kernel=[])
        self.emb_label = MPConv(label_dim, cemb, kernel=[]) if label_dim != 0 else None

        def attn_kind(res):
            return "video" if res in video_attn_resolutions else "frame" if res in frame_attn_resolutions else False

        self.enc = nn.ModuleDict()
        cout = img_channels + 1
        for level, channels in enumerate(cblock):
            res = img_resolution >> level
            if level == 0:
                cin, cout = cout, channels
                self.enc[f"{res}x{res}_conv"] = MPCausal3DGatedConv(cin, cout, kernel=[3, 3, 3])
            else:
                self.enc[f"{res}x{res}_down"] = Block(cout, cout, cemb, flavor="enc", resample_mode="down", **block_kwargs)
            for idx in range(num_blocks):
                cin, cout = cout, channels
                self.enc[f"{res}x{res}_block{idx}"] = Block(cin, cout, cemb, flavor="enc", attention=attn_kind(res),
                                                            **block_kwargs)
        self.dec = nn.ModuleDict()
        skips = [block.out_channels for block in self.enc.values()]
        for level, channels in reversed(list(enumerate(cblock))):
            res = img_resolution >> level
            if level == len(cblock) - 1:
                self.dec[f"{res}x{res}_in0"] = Block(cout, cout, cemb, flavor="dec", attention="video", **block_kwargs)
                self.dec[f"{res}x{res}_in1"] = Block(cout, cout, cemb, flavor="dec", **block_kwargs)
            else:
                self.dec[f"{res}x{res}_up"] = Block(cout, cout, cemb, flavor="dec", resample_mode="up", **block_kwargs)
            for idx in range(num_blocks + 1):
                cin = cout + skips.pop()
                cout = channels
                self.dec[f"{res}x{res}_block{idx}"] = Block(cin, cout, cemb, flavor="dec", attention=attn_kind(res),
                                                            **block_kwargs)
        self.out_conv = MPCausal3DGatedConv(cout, img_channels, kernel=[3, 3, 3])
        frame = inspect.currentframe()
        args, _, _, values = inspect.getargvalues(frame)
        self.kwargs = {arg: values[arg] for arg in args if arg != "self"}

    _oniris_cl_io = True        # forward(..., _cl_io=(B, tt)) takes / returns channels-last bf16 (edm2/loss.py fast path)

    def forward(self, x, c_noise, conditioning=None, cache=None, update_cache=False, just_2d=False, _cl_io=None):
        inner = self._ddp_inner()          # torch DistributedDataParallel around this net (utils.BetterModule)
        if inner is None:
            return self._forward(x, c_noise, conditioning, cache, update_cache, just_2d, _cl_io)
        inner.inner_pre_forward()
        out = self._forward(x, c_noise, conditioning, cache, update_cache, just_2d, _cl_io)
        inner.inner_post_forward(out)
        return out

    def _forward(self, x, c_noise, conditioning=None, cache=None, update_cache=False, just_2d=False, _cl_io=None):
        """_cl_io = (B, tt): `x` is already the packed UNet input (B*tt, H, W, ops.IN_PAD) bf16 with the ones channel
        (ops.dart_input) and the raw channels-last output (B*tt, H, W, 8k) bf16 is returned WITHOUT out_gain -- the
        fused DART loss applies it (ops.dart_loss).  Default: the reference signature (:191)."""
        if _fp32.active():
            if _cl_io is not None:
                raise RuntimeError("the packed bf16 input / output form of UNet.forward does not exist in the fp32 path")
            return _fp32.unet(self, x, c_noise, conditioning, cache, update_cache, just_2d)
        if cache is None:
            cache = {}
        with weights_ready(self):
            B, tt = x.shape[:2] if _cl_io is None else _cl_io
            n_ctx = cache.get("n_context_frames", 0)
            n_new = n_ctx + (tt // 2 if self.training else tt)           # out_res(...)'s frame counter; its gate value is
                                                                         # unused in the reference too (:197), not evaluated here
            if update_cache:
                cache["n_context_frames"] = n_new
            c_noise = c_noise.float()
            cn = c_noise.reshape(-1)
            wn = self.emb_noise.weight.weight
            if not self.training and not torch.is_grad_enabled() and cn.is_cuda and wn.dtype == torch.float32 and wn.shape[1] <= 512:
                # eval: Fourier features, both embedding linears (raw fp32 weights, normalised inside), mp_sum and mp_silu
                # in ONE fp32 launch (~28 tiny launches otherwise, 31 times per generated frame)
                emb = ops.embed_eval(cn.contiguous(), conditioning if self.emb_label is not None else None, self.emb_fourier_sigma,
                                     wn, self.emb_label.weight.weight if self.emb_label is not None else None, self.label_dim)
            elif ops.FUSED_PRELUDE and cn.is_cuda and wn.shape[0] % 8 == 0:
                # training: Fourier features + one-hot in one launch, the two linears on the packed weights, mp_sum +
                # mp_silu in one launch (and one for their adjoint) instead of ~40 + ~50 tiny torch launches
                with_label = self.emb_label is not None and conditioning is not None
                emb = ops.embed_train(cn.contiguous(), conditioning if with_label else None, self.emb_fourier_sigma,
                                      self.emb_noise.weight.pw, self.emb_label.weight.pw if with_label else None,
                                      self.label_dim)
            else:
                emb = ops._prelude_ref("embedding")(self, cn, conditioning)
            # input: (B,t,C,H,W) -> channels-last with the extra all-ones channel (:221), padded to ops.IN_PAD channels
            N = B * tt
            if _cl_io is None:
                xc = x.reshape(N, *x.shape[2:])
                xc = torch.cat([xc, torch.ones_like(xc[:, :1])], dim=1)
                xcl = to_cl(xc, pad_to=max(ops.IN_PAD, -(-xc.shape[1] // 16) * 16))
            else:
                xcl = x
            if not just_2d:
                self._prime_gates(c_noise, cache)
            eb = self.__dict__.get("_oniris_emb_blocks")
            if eb is None:
                blocks = self._emb_blocks()
                eb = self.__dict__["_oniris_emb_blocks"] = (blocks, [b.emb_gain for b in blocks])
            blocks, gains = eb
            if gains[0] is not blocks[0]._parameters["emb_gain"]:           # (a parameter object was replaced)
                gains = eb[1][:] = [b.emb_gain for b in blocks]
            cs = dict(zip(map(id, blocks), ops.emb_scales(emb, self.__dict__["_oniris_groups"][0], gains)))
            skips = []
            # one GradSlot per encoder output (training): it is read by the next block AND by a decoder block (skip)
            use_slots = torch.is_grad_enabled() and self.training and ops.GRAD_SLOTS
            slot = None
            stage_hooks = self.__dict__.get("_oniris_stage_hooks") or {}    # OnirisDDP: {("enc" | "dec", name): tensor hook}
            for name, block in self.enc.items():
                if isinstance(block, Block):
                    xcl, cache["enc", name] = block._cl(xcl, emb, B, c_noise, cache.get(("enc", name)), update_cache, just_2d,
                                                        c=cs[id(block)], in_slot=slot, private_out=bool(use_slots))
                else:
                    xcl, cache["enc", name] = block._cl(xcl, B, c_noise, cache.get(("enc", name)), update_cache, just_2d)
                # (an output of a frozen encoder prefix carries no gradient: nobody would take what the decoder parks for it)
                slot = ops.GradSlot() if (use_slots and xcl.requires_grad) else None
                skips.append((xcl, slot))
                cb = stage_hooks.get(("enc", name))
                if cb is not None and xcl.requires_grad:              # everything downstream of here is final when
                    xcl.register_hook(cb)                             # this gradient arrives (an OnirisDDP stage)
            for name, block in self.dec.items():
                skip, cat_w, skip_slot = None, None, None
                if "block" in name:                                   # mp_cat(x, skip, t) fused into the block's act kernel
                    skip, skip_slot = skips.pop()
                    Na, Nb, t = xcl.shape[-1], skip.shape[-1], self.concat_balance
                    Cn = math.sqrt((Na + Nb) / ((1 - t) ** 2 + t ** 2))
                    cat_w = (Cn / math.sqrt(Na) * (1 - t), Cn / math.sqrt(Nb) * t)
                # (the last encoder output enters the decoder as its main input: `slot` is still that tensor's slot)
                xcl, cache["dec", name] = block._cl(xcl, emb, B, c_noise, cache.get(("dec", name)), update_cache, just_2d,
                                                    skip=skip, cat_w=cat_w, c=cs[id(block)], in_slot=slot, skip_slot=skip_slot,
                                                    private_out=bool(use_slots))
                slot = None
                cb = stage_hooks.get(("dec", name))
                if cb is not None and xcl.requires_grad:
                    xcl.register_hook(cb)
            xcl, cache["out_conv"] = self.out_conv._cl(xcl, B, c_noise, cache.get("out_conv"), update_cache, just_2d)
            if _cl_io is not None:
                return xcl, cache
            out = from_cl(xcl[..., :self.img_channels], torch.float32)
            out = out.reshape(B, tt, *out.shape[1:]) * self.out_gain
            return out, cache

    def _emb_blocks(self):
        return [b for b in list(self.enc.values()) + list(self.dec.values()) if isinstance(b, Block)]

    def _oniris_weight_groups(self):
        """Weights packed row-concatenated (one GEMM for all): every Block's emb_linear reads the same embedding."""
        return [[b.emb_linear.weight for b in self._emb_blocks()]]

    def _oniris_param_classes(self):
        """Layout hint for parallel.FlatParams: {id(param): 0 every step | 1 only 3-D steps | 2 never} -- which steps give
        a parameter its gradient (just_2d skips the context weights and gates, conv.py:60; out_res / emb_time are
        evaluated but unused, networks_edm2.py:197,205-207)."""
        cls = {}
        for m in self.modules():
            if isinstance(m, MPCausal3DGatedConv):
                cls[id(m.weight.weight)] = 1
                for p in m.gating.parameters():
                    cls[id(p)] = 1
        for p in list(self.out_res.parameters()) + list(self.emb_time.parameters()):
            cls[id(p)] = 2
        return cls

    def _ddp_fused_parameters(self):
        """Autograd-visible parameters whose gradients the fused passes of this package deliver (BetterModule.
        _ddp_params_and_buffers_to_ignore): gates of every gated conv, emb_gain of every Block, out_gain."""
        ps = [self.out_gain]
        for m in self.modules():
            if isinstance(m, MPCausal3DGatedConv):
                ps += list(m.gating.parameters())
            elif isinstance(m, Block):
                ps.append(m.emb_gain)
        return ps

    def _oniris_overlap_stages(self, max_stages=6, head_frac=0.06):
        """[((side, block name), [parameters])] for OnirisDDP's early gradient exchanges, in the order the stages become
        final during backward.  The kernel-owned weights (their .grad is written by weight_bwd, not by autograd) of every
        block AFTER a given one (forward order) are final as soon as the backward pass has produced the gradient of that
        block's output -- a tensor hook there starts their exchange beside the backward kernels still to come.  Backward
        walks out_conv, the decoder from its last block to its first, then the encoder from its last block to its first;
        the walk is cut into at most `max_stages` stages of about equal parameter count (the deep decoder / encoder levels
        hold the parameters, the outer levels the time), stopping where at most `head_frac` of them are left: those, and
        everything autograd accumulates itself (gates, emb_gain: evaluated once for all blocks at the top of forward; the
        grouped emb_linear weights), go out in the final exchange."""
        from .conv import NormalizedWeight

        def owned(mod, skip=()):
            return [m.weight for m in mod.modules() if isinstance(m, NormalizedWeight) and id(m.weight) not in skip
                    and m.weight.requires_grad]
        late = {id(b.emb_linear.weight.weight) for b in self._emb_blocks()}
        # blocks in BACKWARD order with the activation whose gradient marks "everything up to here is done": the output of
        # the block that precedes them in forward order
        fwd = [(("enc", n), b) for n, b in self.enc.items()] + [(("dec", n), b) for n, b in self.dec.items()]
        order = [(fwd[-1][0], owned(self.out_conv, late))]                # out_conv is final once dec[-1]'s output has its gradient
        for i in range(len(fwd) - 1, 0, -1):
            order.append((fwd[i - 1][0], owned(fwd[i][1], late)))         # block i is final once block i-1's output has its gradient
        total = sum(w.numel() for _, ws in order for w in ws) + sum(w.numel() for w in owned(fwd[0][1], late))
        if total == 0:
            return []
        per_stage = total * (1.0 - head_frac) / max_stages
        stages, cur, acc, done = [], [], 0, 0
        for key, ws in order:
            cur += ws
            acc += sum(w.numel() for w in ws)
            if acc >= per_stage and cur:
                stages.append((key, cur))
                done += acc
                cur, acc = [], 0
                if len(stages) == max_stages or done >= total * (1.0 - head_frac):
                    break
        return stages

    def _oniris_overlap_plan(self, head_frac=0.12):
        """One-stage form (round 2): (encoder block name, [parameters behind it])."""
        st = self._oniris_overlap_stages(max_stages=1, head_frac=head_frac)
        return (st[0][0], st[0][1]) if st else None

    def prewarm_eval(self, cache):
        """Build, OUTSIDE any hipGraph capture, the small per-frame-count device tables the next cached one-frame
        evaluation will ask for (RoPE tables for the grown key length, the gates' frame-counter vector): they are
        host->device uploads and must not happen inside a capture (edm2/sampler.py _GraphedDenoiser)."""
        from .conv import nctx_tensor, _nctx_i32, _packed_gate_params
        dev = self.out_gain.device
        convs, caches = self._gate_layers(cache)
        n_ctx = [int(c.get("n_context_frames", 0)) if c else 0 for c in caches]
        if any(n_ctx):
            nctx_tensor(n_ctx, dev)
            _nctx_i32(n_ctx, dev)
        _packed_gate_params(convs, dev)
        # context products of the cached pairs, once per frame (conv.MPCausal3DGatedConv.keep_ctx_product) -- only while the
        # packed eval weights of the last evaluation are still the parameters' (otherwise the next evaluation computes them)
        pw = convs[0].weight.pw if convs else None
        bank = getattr(pw, "bank", None)
        if not self.training and bank is not None and bank.packed_valid():
            for conv, c in zip(convs, caches):
                if c:
                    conv.keep_ctx_product(c)
        eb, groups = self.__dict__.get("_oniris_emb_blocks"), self.__dict__.get("_oniris_groups")
        if eb is not None and groups:
            ops.eval_gain_vector(groups[0], eb[1])                 # (cached outside the capture, see there)
        for side, blocks in (("enc", self.enc), ("dec", self.dec)):
            for name, block in blocks.items():
                att = getattr(block, "attn", None)
                kv = (cache.get((side, name)) or {}).get("attn") if isinstance(block, Block) else None
                if att is None or kv is None or not hasattr(att, "rope") or "_tokens_per_frame" not in att.__dict__:
                    continue
                if att.channels != 64 * att.num_heads:            # (padded small heads: no KV ring, ops._attention_eval_hd)
                    continue
                P = att.__dict__["_tokens_per_frame"]
                nk = kv[0].shape[1] // P + 1
                ops.rope_tables(att.rope.inv_freq, att.rope.scale, nk, dev)
                # room for the next frame in the layer's KV ring (grown here, never inside a capture)
                ring = ops.KVRing.of(kv, kv[0].shape[0], P, kv[0].shape[2], 1, dev)
                if ring is not getattr(kv[0], "_oniris_ring", None):
                    cache[(side, name)]["attn"] = ring.views()
                # the committed keys rotated ONCE for the next frame count (every evaluation of the frame reads them)
                ring.rotate_committed((att.rope.inv_freq, att.rope.scale))

    def _gate_layers(self, cache):
        convs, caches = [], []

        def visit(mod, c):
            if isinstance(mod, MPCausal3DGatedConv):
                convs.append(mod); caches.append(c)
            else:
                c = c or {}
                convs.extend([mod.conv_res0, mod.conv_res1]); caches.extend([c.get("conv_res0"), c.get("conv_res1")])
        for name, block in self.enc.items():
            visit(block, cache.get(("enc", name)))
        for name, block in self.dec.items():
            visit(block, cache.get(("dec", name)))
        visit(self.out_conv, cache.get("out_conv"))
        return convs, caches

    def _prime_gates(self, c_noise, cache):
        """Evaluate the gates of all gated convs at once and hand each layer its (ca, cb, counter)."""
        plan = self.__dict__.get("_oniris_gate_plan")          # the module tree of a built net is fixed: walk it once
        if plan is None:
            plan = []                                          # (conv, key of its block's cache, key inside it | None)

            def visit(mod, key):
                if isinstance(mod, MPCausal3DGatedConv):
                    plan.append((mod, key, None))
                else:
                    plan.extend([(mod.conv_res0, key, "conv_res0"), (mod.conv_res1, key, "conv_res1")])
            for name, block in self.enc.items():
                visit(block, ("enc", name))
            for name, block in self.dec.items():
                visit(block, ("dec", name))
            visit(self.out_conv, "out_conv")
            self.__dict__["_oniris_gate_plan"] = plan
            self.__dict__["_oniris_gate_convs"] = [m for m, _, _ in plan]
        convs = self.__dict__["_oniris_gate_convs"]
        if cache:
            caches = []
            for _, key, sub in plan:
                c = cache.get(key)
                caches.append(c if sub is None else (c or {}).get(sub))
        else:
            caches = [None] * len(convs)
        for m, pre in zip(convs, batched_gates(convs, c_noise, caches, self.training)):
            m.__dict__["_gate_pre"] = pre

    def no_sync(self):
        return nullcontext()

    @torch.no_grad()
    def load_from_2d(self, unet):
        """Import the weights of a 2-D EDM2 UNet (NVIDIA's image model; reference networks_edm2.py:238-258, used by
        test.py:28): blocks are matched BY ORDER within enc / dec, every 2-D conv lands in the own-frame path
        (`last_frame_conv`) of the gated conv, the context weights and the gates keep their values; the single
        Fourier embedding feeds both the sigma and the time embedding; `emb_time` is left alone."""
        for mine, theirs in ((self.enc, unet.enc), (self.dec, unet.dec)):
            for m3, m2 in zip(mine.children(), theirs.children()):
                m3.load_from_2d(m2.state_dict())
        sd = unet.state_dict()
        self.emb_noise.load_from_2d(sd["emb_noise.weight"])
        if self.label_dim != 0:
            self.emb_label.load_from_2d(sd["emb_label.weight"])
        for emb in (self.emb_fourier_sigma, self.emb_fourier_time):
            emb.freqs.copy_(sd["emb_fourier.freqs"])
            emb.phases.copy_(sd["emb_fourier.phases"])
        self.out_conv.load_from_2d(sd["out_conv.weight"])
        self.out_gain.copy_(sd["out_gain"])


class Precond(BetterModule):
    def __init__(self, unet, use_fp16=True, sigma_data=0.5):
        super().__init__()
        self.unet, self.use_fp16, self.sigma_data = unet, use_fp16, sigma_data
        self.noise_weight = MultiNoiseLoss()

    def _ddp_fused_parameters(self):
        core = getattr(self.unet, "module", self.unet)
        return core._ddp_fused_parameters() if hasattr(core, "_ddp_fused_parameters") else []

    def forward(self, x, sigma, conditioning=None, force_fp32=False, cache=None, update_cache=False, just_2d=False):
        inner = self._ddp_inner()          # torch DistributedDataParallel around the Precond itself
        if inner is None:
            return self._forward(x, sigma, conditioning, force_fp32, cache, update_cache, just_2d)
        inner.inner_pre_forward()
        out = self._forward(x, sigma, conditioning, force_fp32, cache, update_cache, just_2d)
        inner.inner_post_forward(out)
        return out

    def _forward(self, x, sigma, conditioning=None, force_fp32=False, cache=None, update_cache=False, just_2d=False):
        if cache is None:
            cache = {}
        cache["shape"] = x.shape
        x = x.to(torch.float32)
        if force_fp32 or not self.use_fp16:
            # the reference's precision switch (networks_edm2.py:285,294: dtype = fp32 unless use_fp16 and not force_fp32): the whole
            # net in fp32 on fp32 activations -- fp32.py (HIP fp32 contractions; a verification mode, never the timed path).  Caches
            # written by such a call are fp32 and belong to fp32 calls only.
            sg = sigma.to(torch.float32)[:, :, None, None, None]
            sd = self.sigma_data
            c_skip = sd ** 2 / (sg ** 2 + sd ** 2)
            c_out = sg * sd / (sg ** 2 + sd ** 2).sqrt()
            c_in = 1 / (sd ** 2 + sg ** 2).sqrt()
            c_noise = sigma.to(torch.float32).reshape(sigma.shape[:2]).log() / 4
            with _fp32.fp32_arithmetic():
                F_x, cache = self.unet.forward(c_in * x, c_noise, conditioning, cache, update_cache, just_2d)
            return c_skip * x + c_out * F_x.to(torch.float32), cache
        core = unwrap_ddp(self.unet)       # (torch's DistributedDataParallel hides the UNet's attributes; the CALL still goes through it)
        if (not torch.is_grad_enabled() and x.is_cuda and x.is_contiguous() and x.shape[2] <= 8 and sigma.shape == x.shape[:2]
                and getattr(core, "_oniris_cl_io", False) and getattr(core, "img_channels", -1) == x.shape[2]):
            # eval (the sampler's 31 evaluations per frame): the sigma-preconditioning around the UNet as two HIP passes
            # instead of ~25 tiny torch launches -- c_in * x packed channels-last with the ones channel, and
            # D = c_skip * x + c_out * out_gain * F read straight from the raw channels-last output
            sg = sigma.to(torch.float32).contiguous()
            xcl, c_noise = ops.dart_input(x, None, sg, 1, self.sigma_data, want_c_noise=True)
            Fcl, cache = self.unet.forward(xcl, c_noise, conditioning, cache, update_cache, just_2d,
                                           _cl_io=tuple(x.shape[:2]))
            return ops.precond_out(Fcl, x, sg, core.out_gain, self.sigma_data), cache
        sigma = sigma.to(torch.float32)[:, :, None, None, None]
        sd = self.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
        c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
        c_noise = sigma.reshape(sigma.shape[:2]).log() / 4
        F_x, cache = self.unet.forward(c_in * x, c_noise, conditioning, cache, update_cache, just_2d)
        return c_skip * x + c_out * F_x.to(torch.float32), cache
