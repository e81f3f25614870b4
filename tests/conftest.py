import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The torch-free leg (tests/test_gpu_no_torch.py, RSN_NO_TORCH=1): librsn is loaded HERE, before any test can import torch -- some
    # suites build their inputs with workloads.py, which does -- so that the process's libamdhip64.so.7 is the system's (what librsn is
    # linked against and a Go host gets), whatever is imported later: a library that is already loaded serves every later request for
    # its SONAME, torch's included.  pytest_sessionfinish checks that this held.
    if os.environ.get("RSN_NO_TORCH") == "1":
        from raisin_amd import _lib
        if os.path.exists(_lib.LIB_PATH):
            _lib.lib()


def pytest_sessionfinish(session, exitstatus):
    if os.environ.get("RSN_NO_TORCH") != "1":
        return
    from raisin_amd import _lib
    if _lib._lib is None:
        return
    # (torch, imported later by workloads.py, maps its own copy as well: two runtimes, each serving who bound to it -- librsn was bound
    #  first, to the one the SONAME resolves to: the system's, ROCm 7.2 = 702xxxxx; torch's bundled copy answers 700xxxxx)
    ver, paths = _lib.runtime_info()
    if ver < 70200000:
        session.exitstatus = 3
        sys.stderr.write("RSN_NO_TORCH=1 but librsn runs on HIP runtime %d (%s)\n" % (ver, paths))


@pytest.fixture(scope="session")
def samiam():
    return open(os.path.join(GOLDEN, "samiam.txt"), "rb").read()


@pytest.fixture(scope="session")
def known():
    import json
    return json.load(open(os.path.join(GOLDEN, "known_answers.json")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
