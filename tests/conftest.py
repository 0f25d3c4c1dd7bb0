import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.environ.setdefault("LIBC_FATAL_STDERR_", "1")  # glibc's fatal messages (heap checks, stack protector) to stderr, not to /dev/tty
    # where the reference tree is mounted (the build container) the pins against the reference compiled in place are REQUIRED:
    # tests/test_oracle_vs_ref.py fails instead of skipping when oracle/_ref is missing.  Elsewhere (the GPU box has no
    # /root/reference; the prebuilt oracle/_ref travels with the tree) set KOSK_REQUIRE_REF=1 by hand to get the same.
    if os.path.isdir("/root/reference/kyber"):
        os.environ.setdefault("KOSK_REQUIRE_REF", "1")
    # the oracle is test infrastructure: build it on demand (gcc only, no GPU needed)
    lib = os.path.join(ROOT, "oracle", "libkosk_oracle.so")
    src = os.path.join(ROOT, "oracle", "kosk_oracle.c")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "--no-print-directory"])


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib


def run_gpu_child(code, timeout=900, env=None):
    """Run `code` in a fresh child Python process (never an exec of this one) from the repository root; a crash there -- abort,
    segfault, GPU fault -- is reported as this test's failure with the head and tail of the child's output."""
    e = dict(os.environ)
    e.update(env or {})
    e["PYTHONFAULTHANDLER"] = "1"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=timeout)
    if r.returncode != 0:
        out = r.stdout
        if len(out) > 6000:
            out = out[:3000] + "\n...\n" + out[-3000:]
        pytest.fail("child process exit code %d\n%s" % (r.returncode, out))
    return r.stdout


@pytest.fixture(scope="session")
def gpu_child():
    return run_gpu_child


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """A suite that skipped the reference pins must say so where nobody can miss it."""
    skipped = [r for r in terminalreporter.stats.get("skipped", []) if "test_oracle_vs_ref" in r.nodeid]
    if skipped:
        terminalreporter.write_line("WARNING: %d test(s) of tests/test_oracle_vs_ref.py were SKIPPED: oracle/_ref (the reference compiled in "
                                    "place) is absent, so the oracle is NOT pinned against the reference in this run" % len(skipped), yellow=True)
    elif any("test_oracle_vs_ref" in r.nodeid for r in terminalreporter.stats.get("passed", [])):
        terminalreporter.write_line("oracle pinned against oracle/_ref (the reference's own sources compiled in place): no test skipped")
