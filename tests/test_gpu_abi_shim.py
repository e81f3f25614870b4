"""The cgo overlay's call sequence, compiled and run in C (tests/abi_shim_test.c): the overlay itself cannot be
compiled here (no Go toolchain), the ABI contract it relies on can."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / "abi_shim_test")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_shim_test.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "raisin_amd"), "-lrsn", "-Wl,-rpath," + os.path.join(ROOT, "raisin_amd")])
    return exe


def test_abi_shim_compiles_and_host_only_part_runs(tmp_path):
    """No GPU needed: rsn.h compiles as C, every entry point the shim binds links, the host-only entry point answers,
    device entry points fail cleanly when there is no device."""
    exe = _build(tmp_path)
    out = subprocess.run([exe, "nodev"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "ok" in out.stdout


@pytest.mark.gpu
def test_abi_shim_call_sequence(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "abi shim: ok" in out.stdout
