"""Host-side logic of bench.py that runs without a GPU: the CPU-baseline pinning (BASELINE.md §2 protocol) and the build fingerprint
that ties a PMC summary to the sources it was measured on."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_cpu_busy_reports_fractions():
    busy = bench._cpu_busy(0.05)
    if not busy:
        pytest.skip("/proc/stat not readable here")
    assert all(-0.01 <= v <= 1.01 for v in busy.values())
    assert set(busy) >= set(os.sched_getaffinity(0)) or len(busy) > 0


@pytest.mark.parametrize("n", [1, 2, 4, 1024])
def test_pick_cpus_stays_inside_the_affinity_mask(n):
    allowed = sorted(os.sched_getaffinity(0))
    cpus = bench._pick_cpus(n)
    assert cpus is not None
    assert len(cpus) == min(n, len(allowed))
    assert len(set(cpus)) == len(cpus)
    assert set(cpus) <= set(allowed)
    assert cpus == sorted(cpus)


def test_pick_cpus_prefers_distinct_physical_cores():
    allowed = sorted(os.sched_getaffinity(0))

    def core_of(c):
        try:
            txt = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            return c
        return int(txt.replace("-", ",").split(",")[0])
    cores = {core_of(c) for c in allowed}
    n = min(2, len(cores))
    cpus = bench._pick_cpus(n)
    assert len({core_of(c) for c in cpus}) == n


def test_build_fingerprint_covers_kernel_sources_only(tmp_path):
    a = bench.build_fingerprint()
    assert len(a) == 16 and int(a, 16) >= 0
    assert a == bench.build_fingerprint()   # deterministic; bench.py / tests / docs are not part of it
