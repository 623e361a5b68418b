"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs known answers, host logic, ABI symbol
checks.  `-m gpu` runs on an MI355X: HIP path vs oracle, through the C ABI.
"""
import os
import sys

import pytest

# the oracle's OpenMP team: sleep when idle (a 256-processor box otherwise spins between calls) and stay small —
# the parity tests need the oracle's answers, not its peak throughput (bench.py sets its own thread counts)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("IPCFP_ORACLE_MAX_THREADS", "48")
# the oracle's read_storage_slot runs on a non-throwing reader; under test every call is also answered by the
# exception-based form it replaced and the two must agree (oracle/verify.cpp)
os.environ.setdefault("IPCFP_ORACLE_CHECK_TRY", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def fuzz_seed(base: int) -> int:
    """The differential fuzzers' seeds are fixed (a failure must reproduce); IPCFP_FUZZ_SEED=k moves every one of them
    by k, so that spare minutes on a GPU box can run the same tests over other corpora (tools/gpu_fuzz_seeds.sh)."""
    return base + int(os.environ.get("IPCFP_FUZZ_SEED", "0"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def engine():
    """One engine context on cuda:0.  Fails loudly (no CPU fallback) if the HIP
    library or the device is missing."""
    # torch (its bundled HIP runtime) must come up before libipcfp.so creates its context when both are
    # used in one process — the same order bench.py uses
    import torch

    if torch.cuda.is_available():
        torch.cuda.init()
    import ipc_filecoin_proofs_amd as ipcfp

    eng = ipcfp.Engine(0)
    yield eng
    eng.close()
