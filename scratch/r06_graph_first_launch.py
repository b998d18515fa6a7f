"""What does the FIRST launch of a freshly instantiated hipGraph cost, and does it wait for work that is already queued?
Graph A: 100 kernels of ~10 us (torch.cuda._sleep).  Queue 30 replays of A (~30 ms of GPU work), then launch a NEW graph B (same shape)
for the first time: host time of that call, and the GPU-side distance between the end of A's last kernel and B's first (events)."""
import time, torch
dev = torch.device("cuda")
x = torch.zeros(1024, device=dev)
side = torch.cuda.Stream()
pool = torch.cuda.graph_pool_handle()
def capture():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        g.capture_begin(pool=pool)
        for _ in range(100):
            torch.cuda._sleep(20000)
            x.add_(1.0)
        g.capture_end()
    return g
def t():
    return time.perf_counter()
A = capture()
with torch.cuda.stream(side):
    A.replay()
torch.cuda.synchronize()
for label, other_stream, warm in (("same stream", False, False), ("other stream", True, False), ("same stream, B launched once before (warm)", False, True)):
    B = capture()
    st = torch.cuda.Stream() if other_stream else side
    if warm:
        with torch.cuda.stream(st):
            B.replay()
        torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(side):
        t0 = t()
        for _ in range(30):
            A.replay()
        e0.record()
        tA = t() - t0
    with torch.cuda.stream(st):
        if other_stream:
            st.wait_event(e0)
        t1 = t()
        B.replay()
        tB = t() - t1
        e1.record()
        t2 = t()
        B.replay()
        tB2 = t() - t2
        e2.record()
    torch.cuda.synchronize()
    print(f"{label}: 30 replays of A enqueued in {tA * 1e3:.2f} ms (host); first launch of B: host {tB * 1e3:.2f} ms, second {tB2 * 1e3:.2f} ms; "
          f"GPU: end of A -> end of B's first replay {e0.elapsed_time(e1):.2f} ms, second replay {e1.elapsed_time(e2):.2f} ms")
    del B
