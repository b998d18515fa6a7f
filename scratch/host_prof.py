"""cProfile of the host side of the training step (where do the ~18 ms of enqueue per step go)."""
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--steps", "12", "--warmup", "4", "--cpu-frames", "0", "--no-profile"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45); print(s.getvalue()[:9000])
