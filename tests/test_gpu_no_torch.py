"""The parity suites of the host-buffer API once more on the HIP runtime a C / Go host gets.

Every other -m gpu test runs in a process that has imported torch, and such a process resolves libamdhip64.so.7 to the copy torch
bundles (ROCm 7.0.2) -- librsn then runs on THAT runtime, not on the system's 7.2 it is linked against and that the cgo shim's process
will have (the four calls the shim binds: huffman.go:299,327, lzss.go:109,323).  The two behave differently (DESIGN 0 row 2,
INTEGRATION.md), so the suites that need nothing but bytes in / bytes out run again here, each as a child pytest process under
RSN_NO_TORCH=1 (raisin_amd/_lib.py never imports torch then); a sentinel inside the child checks that torch really is absent and that
the libamdhip64 mapped is the system's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["test_gpu_fuzz.py", "test_gpu_huffman_small.py", "test_gpu_host_pipeline.py", "test_gpu_shapes.py", "test_gpu_lzss_small.py", "test_gpu_literal.py"]


@pytest.mark.gpu
def test_this_process_runs_librsn_on_torchs_runtime_and_says_so():
    """What the rest of the suite runs on, pinned so that the claim in _lib.py / INTEGRATION.md stays true or gets corrected."""
    from raisin_amd import _lib
    if _lib.NO_TORCH:
        pytest.skip("the torch-free leg: see test_no_torch_sentinel")
    _lib.lib()                                                          # (imports torch first, in this mode)
    assert "torch" in sys.modules
    ver, paths = _lib.runtime_info()
    assert any("/torch/" in p for p in paths), paths          # torch's bundled copy is what serves the SONAME here
    assert ver > 0


@pytest.mark.gpu
def test_no_torch_sentinel():
    """Runs INSIDE the torch-free child (selected by name there); in the ordinary suite it starts that child for itself."""
    from raisin_amd import _lib
    if not _lib.NO_TORCH:
        out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_no_torch.py"), "-q", "-m", "gpu", "-k", "sentinel",
                              "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=600, env=dict(os.environ, RSN_NO_TORCH="1"), cwd=ROOT)
        assert out.returncode == 0 and "1 passed" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
        return
    from raisin_amd import huffman, lz
    data = b"the runtime a Go host gets " * 5000
    assert huffman.Decompress(huffman.Compress(data)) == data and lz.Decompress(lz.CompressAsync(data)) == data
    assert "torch" not in sys.modules
    ver, paths = _lib.runtime_info()
    assert paths and all("/torch/" not in p for p in paths), paths
    assert any(p.startswith("/opt/rocm") for p in paths), paths
    assert ver >= 70200000, ver                                     # ROCm 7.2's runtime (torch's bundled copy: 70051831)
    assert _lib.dev_codec(_lib.lib().rsn_huffman_compress_dev, data, len(data) + (1 << 16)) == huffman.Compress(data)      # device memory without torch


@pytest.mark.gpu
@pytest.mark.parametrize("suite", SUITES)
def test_suite_without_torch(suite):
    from raisin_amd import _lib
    if _lib.NO_TORCH:
        pytest.skip("already the torch-free leg")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", suite), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=1500, env=dict(os.environ, RSN_NO_TORCH="1"), cwd=ROOT)
    tail = [ln for ln in out.stdout.strip().splitlines() if " passed" in ln or " failed" in ln or " error" in ln]
    tail = tail[-1] if tail else ""
    # (exit status 3: conftest's check at the session's end -- librsn did NOT run on the system's runtime)
    assert out.returncode == 0 and " passed" in tail and "failed" not in tail, (out.returncode, out.stdout[-3000:], out.stderr[-1500:])
    print(suite, "without torch:", tail)


def test_no_torch_mode_does_not_import_torch():
    """No GPU needed: under RSN_NO_TORCH=1 loading the library (and the host-only entry points) leaves torch unimported."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from raisin_amd import _lib, huffman\n"
            "_lib.lib(); huffman.plan({97: 3, 98: 1})\n"
            "assert 'torch' not in sys.modules and _lib.NO_TORCH\n"
            "print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, RSN_NO_TORCH="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1000:]
