import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, threading
from autoregressive_diffusion_amd import ops
orig = ops.WeightBank._finish
def patched(self):
    print("  _finish: thread", threading.current_thread().name, "stream", torch.cuda.current_stream().cuda_stream, flush=True)
    return orig(self)
ops.WeightBank._finish = patched
orig_b = ops._ConvOp.backward
def pb(ctx, g):
    if not hasattr(pb, "done"):
        pb.done = True
        print("  conv backward: thread", threading.current_thread().name, "stream", torch.cuda.current_stream().cuda_stream, flush=True)
    return orig_b(ctx, g)
ops._ConvOp.backward = staticmethod(pb)
from edm2.conv import MPConv
m = MPConv(32, 32, [3, 3]).cuda().train()
x = torch.randn(2, 32, 8, 8, device="cuda", requires_grad=True)
s = torch.cuda.Stream()
print("default stream", torch.cuda.current_stream().cuda_stream, "side", s.cuda_stream)
with torch.cuda.stream(s):
    print(" forward on", torch.cuda.current_stream().cuda_stream)
    y = m(x)
    y.sum().backward()
torch.cuda.synchronize()
