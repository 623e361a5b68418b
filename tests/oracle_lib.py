"""ctypes loader for the CPU oracle (oracle/libipcfp_oracle.so) — the CHECKER.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libipcfp_oracle.so")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, u64 = C.c_void_p, C.c_uint64
        for name in ("orc_blake2b256", "orc_keccak256", "orc_sha256"):
            f = getattr(lib, name)
            f.restype = None
            f.argtypes = [C.c_char_p, u64, vp]
        lib.orc_hash_batch.restype = None
        lib.orc_hash_batch.argtypes = [C.c_int, vp, vp, vp, u64, vp]
        lib.orc_blake2b256_verify.restype = u64
        lib.orc_blake2b256_verify.argtypes = [vp, vp, vp, vp, u64, vp]

    def _one(self, fn, data: bytes) -> bytes:
        out = np.zeros(32, dtype=np.uint8)
        fn(data, len(data), _p(out))
        return out.tobytes()

    def blake2b256(self, data: bytes) -> bytes:
        return self._one(self.lib.orc_blake2b256, data)

    def keccak256(self, data: bytes) -> bytes:
        return self._one(self.lib.orc_keccak256, data)

    def sha256(self, data: bytes) -> bytes:
        return self._one(self.lib.orc_sha256, data)

    def hash_batch(self, kind: str, data, off, lens):
        k = {"blake2b256": 0, "keccak256": 1, "sha256": 2}[kind]
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out = np.zeros((len(off), 32), dtype=np.uint8)
        self.lib.orc_hash_batch(k, _p(data), _p(off), _p(lens), len(off), _p(out))
        return out

    def blake2b256_verify(self, data, off, lens, expect32):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        expect32 = np.ascontiguousarray(expect32, dtype=np.uint8)
        ok = np.zeros(len(off), dtype=np.uint8)
        good = self.lib.orc_blake2b256_verify(_p(data), _p(off), _p(lens), _p(expect32), len(off), _p(ok))
        return ok, int(good)


_cached = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load() -> Oracle:
    global _cached
    if _cached is None:
        if not os.path.exists(LIB):
            build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached
