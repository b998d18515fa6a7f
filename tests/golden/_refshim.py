"""Harness that lets the read-only reference (/root/reference) run on CPU in THIS container.

Only used by tests/golden/make_golden.py (fixture generation).  Nothing here ships to the GPU box as a
dependency: the generated .npz fixtures are data (inputs + expected outputs).

Adaptations (none touch reference files):
  1. device="cuda" -> "cpu" rewriting through a TorchFunctionMode + no-op .cuda()/.to("cuda").
  2. edm2.attention.attention_modules.compiled_flex_attention replaced by a dense SDPA whose mask is
     block_mask.to_dense() (expanded to token granularity) AND mask_mod  -- this is what the compiled
     FlexAttention kernel computes, and it supports autograd on CPU.  The claim is CHECKED at generation time:
     `reference_compiled_flex` runs the same module calls through the reference's real torch.compile'd function (forward,
     no_grad) and make_golden.py asserts equality to 1e-6 and stores those outputs (`*_y_compiledflex`) in G6 / G6b.
"""
import sys
import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"


def _fix(v):
    if isinstance(v, str) and v.startswith("cuda"):
        return "cpu"
    if isinstance(v, torch.device) and v.type == "cuda":
        return torch.device("cpu")
    return v


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs:
            kwargs["device"] = _fix(kwargs["device"])
        args = tuple(_fix(a) for a in args)
        return func(*args, **kwargs)


_mode = None


def install():
    """Import the reference with the shim active; returns the `edm2` package."""
    global _mode
    if _mode is None:
        _mode = CudaToCpu()
        _mode.__enter__()
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
    # The reference must win over this repository's own top-level `edm2/` compatibility package.  The reference's
    # edm2 has no __init__.py (a namespace package), so a regular package of that name ANYWHERE on sys.path beats it:
    # the repository root goes to the end of sys.path, after the import of the reference package.
    import os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    mine = [p for p in sys.path if os.path.abspath(p or ".") == root]
    for p in mine:
        sys.path.remove(p)
    while REF in sys.path:
        sys.path.remove(REF)
    sys.path.insert(0, REF)
    for name in [m for m in sys.modules if m == "edm2" or m.startswith("edm2.")]:
        del sys.modules[name]
    import edm2  # noqa
    where = list(getattr(edm2, "__path__", []))
    assert where and all(w.startswith(REF) for w in where), f"fixtures must come from the reference, got {where}"
    import edm2.networks_edm2  # noqa
    from edm2.attention import attention_modules as am

    def dense_flex(q, k, v, score_mod=None, block_mask=None):
        Lq, Lk = q.shape[-2], k.shape[-2]
        qi = torch.arange(Lq)[:, None]
        ki = torch.arange(Lk)[None, :]
        if block_mask is not None:
            dense = block_mask.to_dense()[0, 0].bool()          # (nq_blocks, nk_blocks)
            bq, bk = block_mask.BLOCK_SIZE
            allowed = dense.repeat_interleave(bq, 0).repeat_interleave(bk, 1)[:Lq, :Lk]
            if block_mask.mask_mod is not None:
                allowed = allowed & block_mask.mask_mod(0, 0, qi, ki)
        else:
            neg = score_mod(torch.zeros(Lq, Lk), 0, 0, qi, ki)
            allowed = torch.isfinite(neg)
        return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=allowed)

    if not hasattr(am, "_reference_compiled_flex_attention"):
        am._reference_compiled_flex_attention = am.compiled_flex_attention      # the reference's own @torch.compile'd function
    am.compiled_flex_attention = dense_flex
    sys.path.append(root)
    return edm2


class reference_compiled_flex:
    """`with reference_compiled_flex(): y = module(x)` -- the module call runs the reference's OWN `compiled_flex_attention`
    (attention_modules.py:85-88: torch.compile(flex_attention); inductor's CPU backend in this container, forward / no_grad
    only -- with autograd it raises, which is why the fixtures' gradients come from `dense_flex`).  make_golden.py uses it to
    assert, at generation time, that `dense_flex` (block table AND mask_mod) IS what the compiled kernel computes,
    and stores the compiled kernel's output in the fixture."""

    def __enter__(self):
        from edm2.attention import attention_modules as am
        self.am, self.saved = am, am.compiled_flex_attention
        am.compiled_flex_attention = am._reference_compiled_flex_attention
        return self

    def __exit__(self, *exc):
        self.am.compiled_flex_attention = self.saved
        return False
