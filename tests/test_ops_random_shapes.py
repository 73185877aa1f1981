"""Property tests: the conv kernels (forward, input gradient, weight gradient) and the InstanceNorm rows at RANDOM small shapes --
channel counts that are no multiple of anything, odd lengths, every tap count 1..8, stride 1 / 2 -- against the same torch
restatements as the table-driven tests (model.py:21-32 and its autograd).  Derandomised: the same shapes every run."""
import pytest
import torch

pytest.importorskip("hypothesis")   # (test-only dependency; the product does not need it)
from hypothesis import given, settings, strategies as st, HealthCheck  # noqa: E402

from tests import test_ops_conv as C
from tests.emu_util import backend

GPU = pytest.mark.gpu

shape = st.tuples(st.integers(1, 5),                       # B
                  st.integers(1, 72),                      # Cin
                  st.integers(1, 140),                     # Cout
                  st.integers(5, 80),                      # T
                  st.integers(1, 8),                       # KS
                  st.sampled_from([1, 1, 2]))              # stride


def _ok(B, Cin, Cout, T, KS, stride):
    return T > KS // 2 + 1 and B * Cin * Cout * T * KS < 6e6   # (reflect padding needs pad < T; keep the simulator's time per example small)


def _run(kind, which, s):
    B, Cin, Cout, T, KS, stride = s
    if not _ok(*s):
        return
    if which == "fwd":
        C.test_conv_fwd_matches_pad_conv(kind, B, Cin, Cout, T, KS, stride, 0)
    elif which == "dgrad":
        C.test_conv_dgrad_matches_autograd(kind, B, Cin, Cout, T, KS, stride, 0)
    else:
        C.test_conv_wgrad_matches_autograd(kind, B, Cin, Cout, T, KS, stride)


@pytest.mark.parametrize("which", ["fwd", "dgrad", "wgrad"])
def test_conv_random_shapes_on_the_simulator(which):
    backend("emu")

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(shape)
    def run(s):
        _run("emu", which, s)
    run()


@GPU
@pytest.mark.parametrize("which", ["fwd", "dgrad", "wgrad"])
def test_conv_random_shapes_on_the_gpu(which):
    backend("gpu")

    @settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(shape)
    def run(s):
        _run("gpu", which, s)
    run()


# ---- InstanceNorm / AdaIN / activation / residual rows (model.py:296,341,77-83,316-320,362-369)
from tests import test_ops_rowops as R   # noqa: E402

in_shape = st.tuples(st.integers(1, 6), st.integers(1, 20), st.integers(2, 300), st.booleans(), st.sampled_from([0, 1, 2, 5]))


def _run_in(kind, s):
    B, Cc, T, affine, res_mode = s
    if res_mode == 5 and T % 2:     # an upsampled residual has an even length
        T += 1
    if B * Cc * T > 20000:
        return
    R.test_instnorm_fwd_bwd(kind, B, Cc, T, affine, res_mode)


def test_instnorm_random_shapes_on_the_simulator():
    backend("emu")

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(in_shape)
    def run(s):
        _run_in("emu", s)
    run()


@GPU
def test_instnorm_random_shapes_on_the_gpu():
    backend("gpu")

    @settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(in_shape)
    def run(s):
        _run_in("gpu", s)
    run()


# ---- the bf16 pair-storage instances of the same kernels (compute_dtype "bf16": channel pairs in one dword): even channel counts,
# lengths that are multiples of 4 (the engine's shape rule for that mode)
from tests import test_bf16_pairs as BP   # noqa: E402

pair_shape = st.tuples(st.integers(1, 4), st.integers(1, 36).map(lambda v: 2 * v), st.integers(1, 70).map(lambda v: 2 * v),
                       st.integers(2, 20).map(lambda v: 4 * v), st.integers(1, 8), st.sampled_from([1, 1, 2]))


def _run_pairs(kind, which, s):
    B, Cin, Cout, T, KS, stride = s
    if not _ok(*s) or (stride == 2 and (T // 2) % 4):   # (the strided output is a pair tensor too: its length obeys the same rule)
        return
    if which == "fwd":
        BP.test_pairs_conv_fwd(kind, B, Cin, Cout, T, KS, stride, 0, False)
    elif which == "dgrad":
        BP.test_pairs_conv_dgrad(kind, B, Cin, Cout, T, KS, stride, 0)
    else:
        BP.test_pairs_conv_wgrad(kind, B, Cin, Cout, T, KS, stride)


@pytest.mark.parametrize("which", ["fwd", "dgrad", "wgrad"])
def test_pair_conv_random_shapes_on_the_simulator(which):
    backend("emu")

    @settings(max_examples=80, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(pair_shape)
    def run(s):
        _run_pairs("emu", which, s)
    run()
