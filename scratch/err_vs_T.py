"""Error of one full-gym-net 3-D training step against the fp32 oracle as the sequence grows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import test_model_gpu as t
for T in [int(a) for a in sys.argv[1:]] or [8, 16, 32]:
    t.test_cs_shaped_unet_vs_oracle(f"gym-full-net-T{T}", t.GYM_FULL, T, True)
