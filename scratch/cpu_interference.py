"""Does a busy GPU slow the launching host thread down?  Pure-Python / torch-CPU-side loops timed with the GPU idle and with
a long queue of kernels in flight (big matmuls vs many small kernels)."""
import time, torch
dev = "cuda"
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
s = torch.randn(1024, device=dev)
x = torch.randn(64, 64)
def py_loop(n=20000):
    t0 = time.perf_counter(); acc = 0
    for i in range(n):
        acc += i * i % 7
    return (time.perf_counter() - t0) * 1e3
def torch_loop(n=3000):
    t0 = time.perf_counter()
    for i in range(n):
        y = x.reshape(-1)
    return (time.perf_counter() - t0) * 1e3
def alloc_loop(n=2000):
    t0 = time.perf_counter()
    for i in range(n):
        y = torch.empty(1 << 20, device=dev)
    return (time.perf_counter() - t0) * 1e3
for name, fill in (("idle", None), ("big kernels queued", "big"), ("small kernels queued", "small")):
    torch.cuda.synchronize()
    if fill == "big":
        for _ in range(60): b = a @ a           # ~60 x 0.8 ms
    elif fill == "small":
        for _ in range(3000): s.add_(1.0)
    r = (py_loop(), torch_loop(), alloc_loop())
    t0 = time.perf_counter(); torch.cuda.synchronize(); left = (time.perf_counter() - t0) * 1e3
    print(f"{name:22s} python loop {r[0]:6.2f} ms  reshape loop {r[1]:6.2f} ms  empty loop {r[2]:6.2f} ms   (GPU still busy for {left:.1f} ms afterwards)")
