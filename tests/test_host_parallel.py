"""CPU: csrc/host/parallel.h run_parts — the one way the host side of libipcfp.so puts parts of a job on threads of their
own (the bounds check of ipcfp_witness_create, the staged-ring upload, both claim lowerings).  Behind a C ABI nothing may be
thrown: every part runs exactly once whatever the number of parts (more than the pool holds: the rest on the caller's
thread), and a part that throws turns into a `false`, not into std::terminate on a thread nobody joins."""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "parallel_harness.cpp")
LIB = os.path.join(HERE, "native", "libparallel_harness.so")
HDR = os.path.join(HERE, "..", "ipc-filecoin-proofs_amd", "csrc", "host", "parallel.h")


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", LIB, SRC], check=True)
    lib = C.CDLL(LIB)
    lib.parts_run.restype = C.c_int
    lib.parts_run.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.parts_max.restype = C.c_uint
    return lib


def run(lib, parts, throws_at=0xFFFFFFFF):
    ran, mask = C.c_uint64(0), C.c_uint64(0)
    ok = lib.parts_run(parts, throws_at, C.byref(ran), C.byref(mask))
    return bool(ok), ran.value, mask.value


def test_every_part_runs_exactly_once(harness):
    cap = harness.parts_max()
    assert cap >= 32  # (the claim lowerings ask for up to 32 parts)
    for parts in (0, 1, 2, 4, cap - 1, cap, cap + 1, 2 * cap + 3):
        ok, ran, mask = run(harness, parts)
        assert ok and ran == parts, parts
        assert mask == (1 << min(parts, 64)) - 1, parts


def test_a_part_that_throws_is_reported_and_the_others_still_run(harness):
    cap = harness.parts_max()
    for parts, bad in ((1, 0), (4, 0), (4, 3), (cap, cap // 2), (cap + 5, cap + 2)):
        ok, ran, mask = run(harness, parts, bad)
        assert not ok and ran == parts, (parts, bad)
