"""Dispatch-coverage guard: every kernel instantiation that bench.py's timed regions launch
must have been launched -- and its result compared with the oracle -- by a passing `-m gpu` test of this session.

How: the library notes every launch (kernel handle -> demangled instantiation name with its template arguments, plus a tag for
variants picked at run time inside one instantiation such as 'nt-stores'; include/oniris.h `oniris_census`).  tests/conftest.py
switches the census on around every GPU test that is not marked `selfcheck` and keeps the union of what PASSING tests launched
(ORACLE_CENSUS).  This file runs last, calls bench.py's OWN `train()` / `rollout()` at the shapes of the default run (the headline
gym B = 8, T = 64 step and the `extra` records), and asserts `launched by bench  is a subset of  launched under the oracle`.

Size-keyed choices this closes (they are why the guard exists): 128-key dK/dV work items (ops._dkv_item_keys: only at B = 8),
non-temporal instantiations and output stores (csrc/misc.cpp oniris_ew_nt_bytes: tensors >= 96 MiB), the streaming plain conv
(>= 512 tiles), split-K choices of the weight-gradient kernels."""
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.slow]
_bench_census = {}


def _bench_module():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _run_region(name, fn):
    from autoregressive_diffusion_amd import ops
    ops.census_start()
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        seen = ops.census_stop()
    torch.cuda.empty_cache()
    _bench_census[name] = seen
    return seen


REGIONS = ["gym_t64_b8", "gym_t64_b2", "cs_t32", "cs_t64", "rollout_256", "rollout_b8"]


@pytest.mark.parametrize("region", REGIONS)
def test_timed_region_launches_only_oracle_covered_kernels(region):
    """One 3:1 cycle (4 steps: 2-D, 3-D, 3-D, 3-D incl. the fused clip + AdamW + EMA pass) of bench.train() at the shape of
    the region, or bench.rollout() as the default run calls it; the set of (instantiation, tag) it launches against the set
    recorded under passing oracle tests."""
    import conftest
    bench = _bench_module()
    if not conftest.ORACLE_CENSUS:
        pytest.skip("no oracle census in this session (run the whole `-m gpu` suite: this guard reads what the other tests launched)")
    dev = torch.device("cuda", 0)
    wd = bench.Watchdog(0, 0)
    args = types.SimpleNamespace(batch=8, frames=None, accum=1, wrapper="oniris")
    if region == "gym_t64_b8":
        fn = lambda: bench.train(args, "gym", 4, 0, 0, 1, dev, wd, light=True, light_batch=8)
    elif region == "gym_t64_b2":
        fn = lambda: bench.train(args, "gym", 4, 0, 0, 1, dev, wd, light=True, light_batch=2)
    elif region == "cs_t32":
        fn = lambda: bench.train(args, "cs", 4, 0, 0, 1, dev, wd, light=True, light_batch=2)
    elif region == "cs_t64":
        fn = lambda: bench.train(args, "cs", 4, 0, 0, 1, dev, wd, light=True, light_batch=2, light_frames=64)
    elif region == "rollout_256":
        fn = lambda: bench.rollout(types.SimpleNamespace(batch=1, ctx_frames=8, gen_frames=256), quiet=True)
    else:
        fn = lambda: bench.rollout(types.SimpleNamespace(batch=8, ctx_frames=8, gen_frames=8), quiet=True)
    try:
        seen = _run_region(region, fn)
    finally:
        wd.stop()
    assert seen, "the census saw no launch: is the library's launch hook compiled in?"
    missing = sorted(k for k in seen if k not in conftest.ORACLE_CENSUS)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "dispatch_coverage.json"), "w") as f:
        json.dump({"regions": {r: {k: v for k, v in sorted(c.items())} for r, c in _bench_census.items()},
                   "not_oracle_covered": {r: sorted(k for k in c if k not in conftest.ORACLE_CENSUS) for r, c in _bench_census.items()},
                   "oracle_census_size": len(conftest.ORACLE_CENSUS)}, f, indent=1)
    print(f"{region}: {len(seen)} distinct (instantiation, tag) kinds launched, {len(missing)} not under the oracle")
    for k in sorted(seen):
        print(f"   {seen[k]:7d}  {k}   <- {conftest.ORACLE_CENSUS_TESTS.get(k, 'NOT COVERED')}")
    assert not missing, f"{region}: launched by bench.py but by no passing oracle-comparing test: {missing}"
