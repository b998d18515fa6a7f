import json
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a minute or more (full-size CPU-oracle comparisons); still part of -m gpu")
    config.addinivalue_line("markers", "selfcheck: compares the HIP path with itself (fused vs unfused, cached vs uncached, "
                                       "causality, plumbing) -- its launches do NOT count as oracle-covered in the dispatch census")


def pytest_collection_modifyitems(config, items):
    import torch
    # the dispatch-coverage guard reads what every other test launched: it goes last
    items.sort(key=lambda it: 1 if "test_zz_dispatch_coverage" in it.nodeid else 0)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---------------------------------------------------------------------------------------------------------------------
# Dispatch census: while an oracle-comparing GPU test runs, the library notes every kernel
# instantiation it launches (include/oniris.h: oniris_census).  A test's launches join ORACLE_CENSUS only if the test
# PASSED and is not marked `selfcheck`.  tests/test_zz_dispatch_coverage.py then launches what bench.py's timed regions
# launch and asserts that set to be a subset.
ORACLE_CENSUS = {}          # 'kernel instantiation[ [tag]]' -> launches under passing oracle tests
ORACLE_CENSUS_TESTS = {}    # the same key -> first test that launched it


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    import torch
    record = ("gpu" in item.keywords and "selfcheck" not in item.keywords and torch.cuda.is_available()
              and "test_zz_dispatch_coverage" not in item.nodeid)
    if record:
        from autoregressive_diffusion_amd import ops
        ops.census_start()
    outcome = yield
    if record:
        seen = ops.census_stop()
        if outcome.excinfo is None:
            for k, n in seen.items():
                ORACLE_CENSUS[k] = ORACLE_CENSUS.get(k, 0) + n
                ORACLE_CENSUS_TESTS.setdefault(k, item.nodeid)


def pytest_sessionfinish(session, exitstatus):
    if ORACLE_CENSUS:
        out = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "oracle_census.json"), "w") as f:
                json.dump({k: dict(launches=ORACLE_CENSUS[k], first_test=ORACLE_CENSUS_TESTS[k]) for k in sorted(ORACLE_CENSUS)},
                          f, indent=1)
        except OSError:
            pass


@pytest.fixture(params=["default", "nt0"])
def nt_policy(request):
    """Runs an oracle test twice: with the product's non-temporal threshold (96 MiB: oracle-sized tensors take the default-policy
    instantiations) and with the threshold forced to 0, which sends the same tensors through the <NT = true> instantiations and the
    'nt-stores' output paths that bench.py's B = 8 step launches (csrc/misc.cpp oniris_ew_nt_bytes)."""
    from autoregressive_diffusion_amd import ops
    if request.param == "default":
        yield request.param
        return
    old = ops.set_ew_nt_bytes(0)
    try:
        yield request.param
    finally:
        ops.set_ew_nt_bytes(old)
