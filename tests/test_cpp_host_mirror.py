"""The C++ host mirror (twenty-first_amd/host/twenty_first.hpp) restates the reference's own KATs
against the C ABI; this runs its self-test binary (built by __graft_entry__.build())."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "twenty-first_amd", "host")
BIN = os.path.join(HOST, "selftest")


def _build():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)


def test_selftest_without_gpu_reports_no_device(tf):
    if tf.lib().tf_device_count() > 0:
        pytest.skip("a GPU is present")
    _build()
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 77, r.stderr  # ntt() raised TF_ERR_NO_DEVICE: no CPU fallback behind the C ABI


@pytest.mark.gpu
def test_selftest_on_gpu():
    if not os.path.exists(BIN):
        _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all reference KATs pass" in r.stdout
    assert "host threads" in r.stdout and "same words as one device alone" in r.stdout  # the multi-device section ran
    assert "tf_*_multi slice" in r.stdout and "same words as the single-device calls" in r.stdout  # ... and the one-call split
    assert "FAIL" not in r.stdout
