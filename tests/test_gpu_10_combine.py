"""GPU tests of call combining (KOSK_COMBINE, include/kosk_mi355x.h): several caller threads, one handle each, their resident
calls served by merged pipeline runs.  In a fresh child process like every multi-threaded / multi-handle case."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


@pytest.mark.parametrize("k", [2, 3, 4])
def test_combined_calls_equal_uncombined_bytes_under_six_threads(k, torch_cuda, gpu_child):
    """tests/gpu_child_cases.py: combined_calls -- pk, sk, proof images, digest tables, verify bits and fail masks of every caller
    equal an uncombined handle's byte for byte (and the oracle's for three proofs), while combine_stats shows that the calls
    really ran merged."""
    out = gpu_child("from tests.gpu_child_cases import combined_calls; combined_calls(%d)" % k)
    assert "combined_calls ok %d" % k in out


def test_combined_calls_under_glibc_heap_checking(torch_cuda, gpu_child):
    out = gpu_child("from tests.gpu_child_cases import combined_calls; combined_calls(3, threads=5, rounds=3, min_merge=0.0)",
                    env={"MALLOC_CHECK_": "3", "MALLOC_PERTURB_": "165", "LIBC_FATAL_STDERR_": "1"})
    assert "combined_calls ok 3" in out


def test_cohort_members_come_and_go(torch_cuda, gpu_child):
    out = gpu_child("from tests.gpu_child_cases import combined_members_come_and_go; combined_members_come_and_go(3)")
    assert "combined_members_come_and_go ok 3" in out
