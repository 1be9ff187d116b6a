"""Sanitizer builds (race / sanitizer row): AddressSanitizer + UBSan on everything that is host code --
the CPU oracle, and the host side of libp25fe.so (argument checks, shard resolve, the streaming entry points' staging
and state bookkeeping).  GPU AddressSanitizer is not available on this pool; the device side is covered by the guard-band
tests (tests/test_gpu_guards.py)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def run_clean(exe):
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    return r.stdout


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    assert "oracle sanitizer driver ok" in run_clean(os.path.join(ROOT, "build", "oracle_asan"))


def test_library_host_side_under_asan_ubsan():
    """No GPU here: every entry point's argument checks and the pure host logic (p25fe_shard_resolve in both clock modes)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "p25rx_amd", "csrc"), "asan"],
                          env=dict(os.environ, HIPCC=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")))
    out = run_clean(os.path.join(ROOT, "build", "abi_host_asan"))
    assert "abi host driver ok" in out


@pytest.mark.gpu
def test_library_streaming_entry_points_under_host_asan():
    """On the GPU box (the sanitizer build travels prebuilt in build/): the host-buffer entry points -- pinned staging,
    history and tail bookkeeping, capacity / format errors, state export / import -- with the host side instrumented."""
    exe = os.path.join(ROOT, "build", "abi_host_asan")
    if not os.path.exists(exe):
        pytest.skip("sanitizer build not present")
    out = run_clean(exe)
    assert "streaming entry points exercised" in out
