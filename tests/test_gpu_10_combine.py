"""GPU tests of call combining (kosk_options::combine, include/kosk_mi355x.h): several caller threads, one handle each, their resident
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


@pytest.mark.parametrize("callers", [6, 4, 3])
def test_line_of_record_shape(callers, torch_cuda, gpu_child):
    """tests/gpu_child_cases.py: line_of_record_shape -- bench.py's arrangement (kosk_options::combine = 6 by default, 4 and 3 for the side
    runs: that many caller threads x 46 Kyber-768 proofs on device tapes read in place by merged 276- / 184- / 138-proof runs, raw
    resident entry points, resident public keys) against an uncombined handle byte for byte and against the oracle (proof images, keys,
    both digest tables)."""
    out = gpu_child("from tests.gpu_child_cases import line_of_record_shape; line_of_record_shape(callers=%d)" % callers)
    assert "line_of_record_shape ok 3 46 %d callers per run %d.00" % (callers, callers) in out


def test_round_hooks_inside_a_cohort(torch_cuda, gpu_child):
    """tests/gpu_child_cases.py: cohort_round_hooks -- a merged member's hook fires on the run leader's thread with the member's own block;
    kosk_options::hooks_unmerged keeps a hooked handle's calls (and its hook) on its own thread; unequal batch sizes."""
    out = gpu_child("from tests.gpu_child_cases import cohort_round_hooks; cohort_round_hooks()")
    assert "cohort_round_hooks ok 2" in out


def test_member_calls_larger_than_its_block_stay_in_its_block(torch_cuda, gpu_child):
    """tests/gpu_child_cases.py: member_big_batch_stays_in_its_block -- a cohort member's host-buffer calls of 2 * per + 1 proofs are
    chunked by its own batch size and never touch the neighbouring members' blocks of the shared workspace."""
    out = gpu_child("from tests.gpu_child_cases import member_big_batch_stays_in_its_block; member_big_batch_stays_in_its_block(3)")
    assert "member_big_batch_stays_in_its_block ok 3" in out
