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
                           "-L", os.path.join(ROOT, "raisin_amd"), "-lrsn", "-ldl", "-Wl,-rpath," + os.path.join(ROOT, "raisin_amd")])
    return exe


def _build_thread_shim(tmp_path):
    so = str(tmp_path / "pthread_fail_shim.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", os.path.join(ROOT, "tests", "pthread_fail_shim.c"), "-o", so, "-ldl"])
    return so


def test_helper_pool_when_the_system_refuses_threads(tmp_path):
    """No GPU needed: rsn_helpers.h -- the pool every helper thread of the library comes from, and the guard
    every entry point runs in -- under a pthread_create that fails with EAGAIN (`ulimit -u` is not enforced for root and counts a
    whole user's threads): "no helper" is an answer, side jobs run on the caller, existing helpers serve, exceptions become
    RSN_ERR_NOMEM / RSN_ERR_DEVICE.  Never std::terminate (engine.go:315-328 recovers a panic, not a dead process)."""
    exe = str(tmp_path / "thread_fail_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-I", os.path.join(ROOT, "raisin_amd", "csrc"),
                           os.path.join(ROOT, "tests", "thread_fail_test.cpp"), "-o", exe, "-ldl"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, LD_PRELOAD=_build_thread_shim(tmp_path)))
    assert out.returncode == 0 and "thread fail test ok" in out.stdout, (out.stdout, out.stderr[-2000:])


@pytest.mark.gpu
def test_abi_shim_with_threads_refused(tmp_path):
    """The library's own calls that use helper threads (pipelined Huffman decode and LZSS encode, the batch, the sharded stream,
    the side jobs of 20 000-rune alphabets) while pthread_create fails: bytes by the serial form or a negative code + message."""
    exe = _build(tmp_path)
    out = subprocess.run([exe, "threadfail"], capture_output=True, text=True, timeout=900, env=dict(os.environ, LD_PRELOAD=_build_thread_shim(tmp_path)))
    assert out.returncode == 0, (out.stdout, out.stderr[-3000:])
    assert "threads refused" in out.stdout


def test_abi_shim_compiles_and_host_only_part_runs(tmp_path):
    """No GPU needed: rsn.h compiles as C, every entry point the shim binds links, the host-only entry point answers,
    device entry points fail cleanly when there is no device."""
    exe = _build(tmp_path)
    out = subprocess.run([exe, "nodev"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"RSN_BATCH_WORKERS": "3"}, {"RSN_BATCH_WORKERS": "2"},
                                 {"RSN_DEVICE": "rr", "RSN_MAX_PARKED": "1"}, {"RSN_BATCH_KEEP_MIB": "0", "RSN_BATCH_WORKERS": "8"}],
                         ids=["default", "3 workers", "2 workers", "round-robin devices, 1 parked", "8 workers, ring released"])
def test_abi_shim_call_sequence(tmp_path, env):
    """The shim's call sequence, also with the batch split over several per-device workers (on a one-GPU box they share the
    device: the dealing-out, the per-worker rings and the error path are the same code a multi-GPU node runs), with the default
    device handed out round-robin and with the parked-context pool capped at one."""
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "abi shim: ok" in out.stdout


@pytest.mark.gpu
def test_batch_split_over_workers_equals_single_calls(monkeypatch):
    """rsn_huffman_compress_batch with chunk k -> worker k mod G (engine.go:150-154): every segment equals the single
    call's, whatever the number of workers/devices, ragged chunk sizes, byte and rune alphabets mixed."""
    import numpy as np

    from raisin_amd import _lib, huffman
    rng = np.random.default_rng(5)
    chunks = [rng.integers(0, 128, size=200000 + 33333 * i, dtype=np.uint8).tobytes() for i in range(7)]
    chunks[3] = ("\u0416\u0443\u043a " * 40000).encode()          # a rune-path chunk between byte chunks
    chunks.append(b"z")                                          # single-symbol chunk
    single = [huffman.Compress(c) for c in chunks]
    for workers in ("1", "2", "3", "8", "64"):
        monkeypatch.setenv("RSN_BATCH_WORKERS", workers)
        assert huffman.CompressBatch(chunks) == single, workers
    monkeypatch.delenv("RSN_BATCH_WORKERS")
    monkeypatch.setenv("RSN_BATCH_DEVICES", "1")
    assert huffman.CompressBatch(chunks) == single
    assert _lib.lib().rsn_device_count() >= 1
    # an error in one worker's chunk undoes the whole batch
    monkeypatch.setenv("RSN_BATCH_WORKERS", "3")
    with pytest.raises(_lib.RsnError):
        huffman.CompressBatch(chunks[:4] + [b""] + chunks[4:])
