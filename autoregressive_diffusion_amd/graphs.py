"""hipGraph capture of a static-shape training micro-step (forward + backward of the denoiser), replayed every step.

The reference relies on torch.compile for its attention kernel only; here the whole launch sequence (~1500 small and
large kernels per step, all enqueued on one HIP stream with caller-allocated buffers and no host synchronisation)
is captured once per step flavour (3-D / 2-D) and replayed, which removes the per-launch host cost.  The optimizer and
the gradient all-reduce stay outside the graph (their arguments change per step / they talk to RCCL)."""
import torch

_capture_stream = None


class GraphedStep:
    def __init__(self, fn, warmup=3, params=None, flat=None):
        """fn(): runs forward+backward on STATIC input tensors and returns a (loss) tensor.
        params: parameters whose .grad autograd (re)creates inside fn (FlatParams(lazy_small=True) sets them to None
        before every backward): the tensors the CAPTURED backward assigned are the ones every replay writes, so they
        are re-attached as .grad after each replay (otherwise FlatParams.gather() sees the stale aliases of the
        previous step and the replayed gradients of ~190 small parameters are silently dropped)."""
        self.fn, self.warmup = fn, warmup
        self.params, self._grads = list(params or ()), []
        # flat: the FlatParams of the net -- which weights the captured backward produced gradients for is host-side
        # bookkeeping (FlatAdamW skips parameters without gradient like torch.optim); a replay runs no Python, so the
        # record of the captured backward is re-applied after every replay
        from .parallel import FlatParams
        if flat is None and self.params:
            flat = FlatParams.owner_of(self.params[0])
        # GraphedStep(fn) with neither params= nor flat=: without a FlatParams the replays could not re-apply the gradient
        # bookkeeping and FlatAdamW would silently freeze every kernel-owned weight after the capture step.  The owner is
        # then found at capture time: whichever live FlatParams the captured backward marked (snapshot_touched non-empty).
        self.flat, self._touched = flat, []
        self.graph, self.out, self.calls = None, None, 0
        # ONE side stream for the warm-up and the capture of EVERY GraphedStep: autograd pins each parameter's
        # AccumulateGrad node to the stream it was first used on; a node living on another stream would run outside
        # the capture (its work silently missing from the replay).
        global _capture_stream
        if _capture_stream is None:
            _capture_stream = torch.cuda.Stream()
        self.stream = _capture_stream

    def __call__(self):
        if self.graph is not None:
            # replay on the capture stream, fenced by events against the caller's stream on both sides: on ROCm 7.0 a
            # graph launched directly behind eager kernels of the same stream was observed to start before they
            # finished (AdamW of step n racing the zero_grad/weight_prep of step n+1)
            cur = torch.cuda.current_stream()
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.graph.replay()
            cur.wait_stream(self.stream)
            for p, g in self._grads:
                p.grad = g
            for f, snap in self._touched:
                f.restore_touched(snap)
            return self.out
        self.calls += 1
        if self.calls <= self.warmup:              # eager warm-up: builds tables, sets kernel attributes, fills caches
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                out = self.fn()
            torch.cuda.current_stream().wait_stream(self.stream)
            return out
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            out = self.fn()
        self.graph, self.out = g, out
        self._grads = [(p, p.grad) for p in self.params if p.grad is not None]
        from .parallel import FlatParams
        cands = [self.flat] if self.flat is not None else FlatParams.live()
        self._touched = [(f, snap) for f, snap in ((f, f.snapshot_touched()) for f in cands) if snap[0] or snap[1]]
        if self.flat is None and len(self._touched) == 1:
            self.flat = self._touched[0][0]
        return self.__call__()
