"""cProfile (main thread: forward + optimizer; the backward runs on autograd's thread) of many steady-state steps."""
import cProfile, pstats, sys, os, io
B, T = os.environ.get("PB", "2"), os.environ.get("PT", "64")
sys.argv = ["bench.py", "--steps", "60", "--warmup", "4", "--cpu-frames", "0", "--no-profile", "--batch", B, "--frames", T]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:7000])
