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
        lib.orc_cid_to_string.restype = C.c_int
        lib.orc_cid_to_string.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]
        lib.orc_cid_from_string.restype = C.c_int
        lib.orc_cid_from_string.argtypes = [C.c_char_p, vp]
        lib.orc_cid_for_block.restype = None
        lib.orc_cid_for_block.argtypes = [C.c_char_p, u64, vp]
        lib.orc_store_create.restype = vp
        lib.orc_store_create.argtypes = [vp, vp, vp, vp, u64]
        lib.orc_store_create_var.restype = vp
        lib.orc_store_create_var.argtypes = [vp, vp, vp, vp, vp, vp, u64]
        lib.orc_store_create_mt.restype = vp
        lib.orc_store_create_mt.argtypes = [vp, vp, vp, vp, u64, C.c_int]
        lib.orc_store_size.restype = u64
        lib.orc_store_size.argtypes = [vp]
        lib.orc_num_procs.restype = C.c_int
        lib.orc_num_procs.argtypes = []
        lib.orc_use_threads.restype = C.c_int
        lib.orc_use_threads.argtypes = [C.c_int]
        lib.orc_blake2b256_verify_mt.restype = u64
        lib.orc_blake2b256_verify_mt.argtypes = [vp, vp, vp, vp, u64, vp, C.c_int]
        lib.orc_scan_events_mt.restype = C.c_uint8
        lib.orc_scan_events_mt.argtypes = [vp, vp, vp, C.c_int, u64, vp, u64, C.POINTER(u64), vp, u64, C.POINTER(u64),
                                           vp, u64, C.POINTER(u64), C.c_int]
        lib.orc_store_destroy.restype = None
        lib.orc_store_destroy.argtypes = [vp]
        lib.orc_amt_get.restype = None
        lib.orc_amt_get.argtypes = [vp, vp, C.c_int, C.c_int, vp, u64, vp, vp, C.c_uint32, vp]
        lib.orc_hamt_get.restype = None
        lib.orc_hamt_get.argtypes = [vp, vp, C.c_uint32, C.c_int, vp, vp, vp, u64, vp, vp, C.c_uint32, vp]
        lib.orc_hamt_get_t.restype = None
        lib.orc_hamt_get_t.argtypes = [vp, vp, C.c_uint32, C.c_int, vp, vp, vp, u64, vp, vp, C.c_uint32, vp, C.c_int]
        lib.orc_exec_order.restype = C.c_uint8
        lib.orc_exec_order.argtypes = [vp, vp, C.c_uint32, vp, u64, C.POINTER(u64)]
        lib.orc_scan_events.restype = C.c_uint8
        lib.orc_scan_events.argtypes = [vp, vp, vp, C.c_int, u64, vp, u64, C.POINTER(u64), vp, u64, C.POINTER(u64),
                                        vp, u64, C.POINTER(u64)]
        lib.orc_generate_event_proof.restype = C.c_uint8
        lib.orc_generate_event_proof.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_int, u64, vp, vp, u64, C.POINTER(u64),
                                                 vp, u64, C.POINTER(u64)]
        lib.orc_generate_storage_proof.restype = C.c_uint8
        lib.orc_generate_storage_proof.argtypes = [vp, vp, u64, vp, vp, vp, vp, u64, C.POINTER(u64)]
        lib.orc_verify_event_proofs.restype = None
        lib.orc_verify_event_proofs.argtypes = [vp, vp, u64, vp, vp, vp, C.c_int, C.c_int]
        lib.orc_verify_storage_claims_packed.restype = None
        lib.orc_verify_storage_claims_packed.argtypes = [vp, vp, u64, vp, vp, C.c_int]
        lib.orc_verify_event_claims_packed.restype = None
        lib.orc_verify_event_claims_packed.argtypes = [vp, vp, C.c_uint32, vp, u64, vp, vp, vp, vp, C.c_int]
        lib.orc_verify_storage_proofs.restype = None
        lib.orc_verify_storage_proofs.argtypes = [vp, vp, u64, vp, vp, C.c_int, C.c_int]

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

    # ---- CIDs ----
    def cid_to_string(self, cid: bytes) -> str:
        buf = C.create_string_buffer(256)
        n = self.lib.orc_cid_to_string(cid, len(cid), buf, 256)
        assert n >= 0
        return buf.value.decode()

    def cid_from_string(self, s: str):
        out = np.zeros(40, dtype=np.uint8)
        n = self.lib.orc_cid_from_string(s.encode(), _p(out))
        return None if n < 0 else out.tobytes()[:n]

    def cid_for_block(self, data: bytes) -> bytes:
        out = np.zeros(40, dtype=np.uint8)
        self.lib.orc_cid_for_block(data, len(data), _p(out))
        return out.tobytes()[:38]

    def store(self, data, off, lens, cids40, threads=1):
        """threads=1: the reference's sequential load_witness_store; 0 = every processor (baseline B2 all-cores)"""
        return OracleStore(self, data, off, lens, cids40, threads)

    def store_var(self, data, off, lens, cids):
        """The store keyed by CIDs of ANY length (`cids`: list of bytes) — a CID with a 64-byte digest does not fit the
        40-byte slots of `store`."""
        return OracleStore(self, data, off, lens, None, var_cids=cids)

    def num_procs(self) -> int:
        return int(self.lib.orc_num_procs())

    def use_threads(self, threads: int) -> int:
        """The OpenMP thread count `threads` resolves to (0 = every processor, capped by IPCFP_ORACLE_MAX_THREADS);
        also grows the worker threads' malloc arenas ahead of the first parallel phase."""
        return int(self.lib.orc_use_threads(threads))

    def hash_batch(self, kind: str, data, off, lens):
        k = {"blake2b256": 0, "keccak256": 1, "sha256": 2}[kind]
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out = np.zeros((len(off), 32), dtype=np.uint8)
        self.lib.orc_hash_batch(k, _p(data), _p(off), _p(lens), len(off), _p(out))
        return out

    def blake2b256_verify(self, data, off, lens, expect32, threads=1):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        expect32 = np.ascontiguousarray(expect32, dtype=np.uint8)
        ok = np.zeros(len(off), dtype=np.uint8)
        if threads == 1:
            good = self.lib.orc_blake2b256_verify(_p(data), _p(off), _p(lens), _p(expect32), len(off), _p(ok))
        else:
            good = self.lib.orc_blake2b256_verify_mt(_p(data), _p(off), _p(lens), _p(expect32), len(off), _p(ok), threads)
        return ok, int(good)


VALUE_KINDS = {"cid": 0, "receipt": 1, "stamped_event": 2, "actor_state": 3, "vec_u8": 4, "any": 5}


class OracleStore:
    """MemoryBlockstore over a witness table (keeps the arrays alive)."""

    def __init__(self, orc: Oracle, data, off, lens, cids40, threads=1, var_cids=None):
        self.orc = orc
        self.lib = orc.lib
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.off = np.ascontiguousarray(off, dtype=np.uint64)
        self.lens = np.ascontiguousarray(lens, dtype=np.uint32)
        if var_cids is not None:
            self.cid_len = np.array([len(c) for c in var_cids], dtype=np.uint32)
            self.cid_off = np.zeros(len(var_cids), dtype=np.uint64)
            if len(var_cids):
                self.cid_off[1:] = np.cumsum(self.cid_len[:-1], dtype=np.uint64)
            self.cid_bytes = np.frombuffer(b"".join(bytes(c) for c in var_cids) + b"\0", dtype=np.uint8).copy()
            self.h = self.lib.orc_store_create_var(_p(self.data), _p(self.off), _p(self.lens), _p(self.cid_bytes), _p(self.cid_off),
                                                   _p(self.cid_len), len(self.off))
            return
        self.cids = np.ascontiguousarray(cids40, dtype=np.uint8).reshape(-1, 40)
        if threads == 1:
            self.h = self.lib.orc_store_create(_p(self.data), _p(self.off), _p(self.lens), _p(self.cids), len(self.off))
        else:
            self.h = self.lib.orc_store_create_mt(_p(self.data), _p(self.off), _p(self.lens), _p(self.cids),
                                                  len(self.off), threads)

    def size(self) -> int:
        return int(self.lib.orc_store_size(self.h))

    def close(self):
        if self.h:
            self.lib.orc_store_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def amt_get(self, root40: bytes, version: int, kind: str, indices, cap=1024):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        n = len(idx)
        st = np.zeros(n, dtype=np.uint8)
        out = np.zeros((n, cap), dtype=np.uint8)
        ol = np.zeros(n, dtype=np.uint32)
        root = np.frombuffer(root40.ljust(40, b"\0"), dtype=np.uint8).copy()
        self.lib.orc_amt_get(self.h, _p(root), version, VALUE_KINDS[kind], _p(idx), n, _p(st), _p(out), cap, _p(ol))
        return st, [out[i, : ol[i]].tobytes() for i in range(n)]

    def hamt_get(self, root40: bytes, bit_width: int, kind: str, keys, cap=1024, want_values=True, threads=0):
        n = len(keys)
        kl = np.array([len(k) for k in keys], dtype=np.uint32)
        ko = np.zeros(n, dtype=np.uint32)
        if n:
            ko[1:] = np.cumsum(kl[:-1])
        kb = np.frombuffer(b"".join(keys), dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
        st = np.zeros(n, dtype=np.uint8)
        out = np.zeros((n, cap), dtype=np.uint8)
        ol = np.zeros(n, dtype=np.uint32)
        root = np.frombuffer(root40.ljust(40, b"\0"), dtype=np.uint8).copy()
        import time as _time

        t0 = _time.perf_counter()
        self.lib.orc_hamt_get_t(self.h, _p(root), bit_width, VALUE_KINDS[kind], _p(kb), _p(ko), _p(kl), n, _p(st),
                                _p(out), cap, _p(ol), int(threads))
        self.last_call_seconds = _time.perf_counter() - t0  # the C call alone (bench.py: no Python marshalling in a baseline)
        if not want_values:
            return st, None
        return st, [out[i, : ol[i]].tobytes() for i in range(n)]

    def verify_event_proofs(self, claims, trust=None, filt=None, mode=0, threads=0):
        st = np.zeros(claims.n, dtype=np.uint8)
        self.lib.orc_verify_event_proofs(self.h, C.cast(claims.arr, C.c_void_p), claims.n,
                                         C.cast(C.pointer(trust), C.c_void_p) if trust is not None else None,
                                         C.cast(C.pointer(filt), C.c_void_p) if filt is not None else None,
                                         _p(st), mode, threads)
        return st

    def verify_storage_proofs(self, claims, trust=None, mode=0, threads=0):
        st = np.zeros(claims.n, dtype=np.uint8)
        self.lib.orc_verify_storage_proofs(self.h, C.cast(claims.arr, C.c_void_p), claims.n,
                                           C.cast(C.pointer(trust), C.c_void_p) if trust is not None else None,
                                           _p(st), mode, threads)
        return st

    def verify_storage_claims_packed(self, claims: np.ndarray, trust=None, threads=0):
        """Packed ipcfp_storage_claim_t[n] (all flags set) → status bytes, through the string verifier."""
        claims = np.ascontiguousarray(claims)
        st = np.zeros(len(claims), dtype=np.uint8)
        self.lib.orc_verify_storage_claims_packed(self.h, _p(claims), len(claims),
                                                  C.cast(C.pointer(trust), C.c_void_p) if trust is not None else None,
                                                  _p(st), threads)
        return st

    def verify_event_claims_packed(self, tipsets: np.ndarray, claims: np.ndarray, blob: np.ndarray, trust=None,
                                   filt=None, threads=0):
        """Packed ipcfp_event_claim_t[n] (all flags set) → status bytes, through the string verifier (B2)."""
        tipsets = np.ascontiguousarray(tipsets)
        claims = np.ascontiguousarray(claims)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        st = np.zeros(len(claims), dtype=np.uint8)
        self.lib.orc_verify_event_claims_packed(self.h, _p(tipsets), len(tipsets), _p(claims), len(claims), _p(blob),
                                                C.cast(C.pointer(trust), C.c_void_p) if trust is not None else None,
                                                C.cast(C.pointer(filt), C.c_void_p) if filt is not None else None,
                                                _p(st), threads)
        return st

    def exec_order(self, parent_cids, cap=1 << 21):
        pc = np.zeros((len(parent_cids), 40), dtype=np.uint8)
        for i, c in enumerate(parent_cids):
            pc[i, : len(c)] = np.frombuffer(c, dtype=np.uint8)
        out = np.zeros((cap, 40), dtype=np.uint8)
        cnt = C.c_uint64()
        st = self.lib.orc_exec_order(self.h, _p(pc), len(parent_cids), _p(out), cap, C.byref(cnt))
        return int(st), out[: min(cnt.value, cap)]

    def scan_events(self, receipts_root: bytes, topic0: bytes, topic1: bytes, actor=None, cap_receipts=1 << 21,
                    cap_matches=1 << 20, cap_touched=1 << 21, want_touched=True, threads=1):
        filt = np.frombuffer(topic0 + topic1, dtype=np.uint8).copy()
        root = np.frombuffer(receipts_root.ljust(40, b"\0"), dtype=np.uint8).copy()
        has = np.zeros(cap_receipts, dtype=np.uint8)
        trip = np.zeros((cap_matches, 3), dtype=np.uint64)
        touched = np.zeros((cap_touched, 40), dtype=np.uint8) if want_touched else None
        nr, nm, nt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        st = self.lib.orc_scan_events_mt(self.h, _p(root), _p(filt), 0 if actor is None else 1,
                                         0 if actor is None else actor, _p(has), cap_receipts, C.byref(nr), _p(trip),
                                         cap_matches, C.byref(nm), _p(touched), cap_touched, C.byref(nt), threads)
        return int(st), has[: nr.value], trip[: nm.value], (touched[: nt.value] if want_touched else None)

    def generate_event_proof(self, parent_cids, child_cid: bytes, topic0: bytes, topic1: bytes, actor=None,
                             cap_proofs=1 << 20, cap_witness=1 << 21):
        """(status, triples u64[n,3], message cids u8[n,40], witness cids u8[m,40] in BTreeSet order)"""
        pc = np.zeros((len(parent_cids), 40), dtype=np.uint8)
        for i, c in enumerate(parent_cids):
            pc[i, : len(c)] = np.frombuffer(c, dtype=np.uint8)
        child = np.frombuffer(child_cid.ljust(40, b"\0"), dtype=np.uint8).copy()
        filt = np.frombuffer(topic0 + topic1, dtype=np.uint8).copy()
        trip = np.zeros((cap_proofs, 3), dtype=np.uint64)
        msg = np.zeros((cap_proofs, 40), dtype=np.uint8)
        wit = np.zeros((cap_witness, 40), dtype=np.uint8)
        n_p, n_w = C.c_uint64(), C.c_uint64()
        st = self.lib.orc_generate_event_proof(self.h, _p(pc), len(parent_cids), _p(child), _p(filt),
                                               0 if actor is None else 1, 0 if actor is None else actor, _p(trip),
                                               _p(msg), cap_proofs, C.byref(n_p), _p(wit), cap_witness, C.byref(n_w))
        return int(st), trip[: n_p.value], msg[: n_p.value], wit[: n_w.value]

    def generate_storage_proof(self, child_cid: bytes, actor_id: int, slot32: bytes, cap_witness=1 << 16):
        """(status, [parent_state_root, actor_state_cid, storage_root] u8[3,40], value u8[32], witness cids)"""
        child = np.frombuffer(child_cid.ljust(40, b"\0"), dtype=np.uint8).copy()
        slot = np.frombuffer(slot32, dtype=np.uint8).copy()
        out3 = np.zeros((3, 40), dtype=np.uint8)
        val = np.zeros(32, dtype=np.uint8)
        wit = np.zeros((cap_witness, 40), dtype=np.uint8)
        n_w = C.c_uint64()
        st = self.lib.orc_generate_storage_proof(self.h, _p(child), int(actor_id), _p(slot), _p(out3), _p(val), _p(wit),
                                                 cap_witness, C.byref(n_w))
        return int(st), out3, val, wit[: n_w.value]


_cached = None
_cached_native = None


def build():
    subprocess.run(["make", "-s", "-j8", "-C", ORACLE_DIR], check=True)


def load() -> Oracle:
    global _cached
    if _cached is None:
        override = os.environ.get("IPCFP_ORACLE_LIB")  # e.g. oracle/_asan/libipcfp_oracle.so (tests/test_sanitizers.py)
        if override:
            _cached = Oracle(C.CDLL(override))
            return _cached
        if not os.path.exists(LIB):
            build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached


def load_native():
    """The oracle compiled ON THIS MACHINE with -march=native (oracle/_native/, `make native`) — what bench.py's
    cpu_baseline leg times.  Returns (oracle, "native"), or (portable oracle, "x86-64-v3") when
    the build is not possible here."""
    global _cached_native
    if _cached_native is None:
        lib = os.path.join(ORACLE_DIR, "_native", "libipcfp_oracle.so")
        try:
            subprocess.run(["make", "-s", "-j16", "-C", ORACLE_DIR, "native"], check=True, timeout=600,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _cached_native = (Oracle(C.CDLL(lib)), "native")
        except (OSError, subprocess.SubprocessError):
            _cached_native = (load(), "x86-64-v3")
    return _cached_native
