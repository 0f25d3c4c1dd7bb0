"""CPU sanitizer targets (the reference's CMakeLists.txt:6-7 has none): the oracle and the product's host
file csrc/kosk_host.cpp under ASan+UBSan (K = 2,3,4 prove / verify / tamper, host keygen, Fiat-Shamir batches) and the
thread pool under TSan (back-to-back small jobs: the hand-off that used to race)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(target):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "--no-print-directory", target], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        pytest.fail("make %s failed:\n%s" % (target, r.stdout[-3000:]))


def test_asan_ubsan_oracle_and_host_code():
    _build("asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_san", "san_driver_asan"), "full"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0 and "san_driver full: ok" in r.stdout, r.stdout[-3000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout


def test_tsan_thread_pool_and_batched_fiat_shamir():
    _build("tsan")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_san", "san_driver_tsan"), "pool", "20000"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "san_driver pool: ok" in r.stdout, r.stdout[-3000:]
    assert "ThreadSanitizer" not in r.stdout


@pytest.mark.parametrize("san", ["tsan", "asan"])
def test_lane_threads_and_chunk_dealing_under_sanitizers(san):
    """csrc/kosk_lanes.hpp (the lane threads and chunk dealing behind kosk_verifiable_keygen_batch / kosk_verify_batch with
    KOSK_STREAMS > 1) on fake sub-contexts: every unit exactly once, failing and throwing chunks contained, no data race."""
    _build(san)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_san", "san_driver_" + san), "lanes", "1500"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "san_driver lanes: ok" in r.stdout, r.stdout[-3000:]
    assert "ThreadSanitizer" not in r.stdout and "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout


@pytest.mark.parametrize("san", ["tsan", "asan"])
def test_call_combiner_under_sanitizers(san):
    """csrc/kosk_combine.hpp (the combiner behind the merged resident calls of a KOSK_COMBINE cohort) with fake executors: every
    call served once by a run of neighbouring members of its own kind, short batches only at a run's end, exceptions contained,
    opposite-phase callers fall into step, a lone caller is not delayed; no data race, no leak."""
    _build(san)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_san", "san_driver_" + san), "combine", "300"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "san_driver combine: ok" in r.stdout, r.stdout[-3000:]
    assert "ThreadSanitizer" not in r.stdout and "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout
