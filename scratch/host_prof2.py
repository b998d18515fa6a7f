"""torch.profiler (CPU activity) over the steady-state training steps: host cost per op / autograd node."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from torch.profiler import profile, ProfilerActivity
import bench
B, T = int(os.environ.get("PB", 1)), int(os.environ.get("PT", 8))
sys.argv = ["bench.py", "--steps", "16", "--warmup", "6", "--cpu-frames", "0", "--no-profile", "--batch", str(B), "--frames", str(T)]
# wrap the timed region: bench.main() calls time.perf_counter() right before / after it; simplest is to profile all of main
# after a first full run has warmed every cache
bench.main()
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    bench.main()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=110, max_name_column_width=70))
