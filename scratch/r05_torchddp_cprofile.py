"""Host profile of the reference loop under torch DistributedDataParallel (one RCCL rank): where the ~20 ms per step over the
un-wrapped loop go.  python scratch/r05_torchddp_cprofile.py [torch|oniris] -> gpurun_out/r05_cprofile_<wrapper>.txt"""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wrapper = sys.argv[1] if len(sys.argv) > 1 else "torch"
dist_ = len(sys.argv) > 2 and sys.argv[2] == "dist"
if dist_:
    os.environ["ONIRIS_FORCE_DIST"] = "1"
sys.argv = ["bench.py", "--wrapper", wrapper, "--steps", "8", "--warmup", "4", "--cpu-frames", "0", "--no-extra", "--no-profile"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
s2 = io.StringIO()
pstats.Stats(pr, stream=s2).sort_stats("tottime").print_stats(45)
s3 = io.StringIO()
pstats.Stats(pr, stream=s3).sort_stats("tottime").print_callers("built-in method torch.empty|run_backward|method 'cpu'")
s2.write("\n\n======== callers\n" + s3.getvalue())
open("gpurun_out/r05_cprofile_%s%s.txt" % (wrapper, "_dist" if dist_ else ""), "w").write(s.getvalue() + "\n\n======== tottime\n" + s2.getvalue())
