"""ctypes loader of the synthetic tipset generator (tools/synth/synth.cpp).

Input tooling for tests and bench.py: builds a seeded chain fragment with real CIDs
(SURVEY.md §8d).  Not part of the product; does not use oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libipcfp_synth.so")
SEED_BASE = 0x1BC0F11EC0150000


class Params(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("parent_epoch", C.c_int64),
        ("n_parents", C.c_uint32),
        ("dup_permille", C.c_uint32),
        ("n_receipts", C.c_uint64),
        ("max_events", C.c_uint32),
        ("no_events_permille", C.c_uint32),
        ("n_emitters", C.c_uint32),
        ("emitter_base", C.c_uint32),
        ("n_sigs", C.c_uint32),
        ("n_subnets", C.c_uint32),
        ("n_planted", C.c_uint64),
        ("filter_actor", C.c_uint64),
        ("variety", C.c_uint32),
        ("events_bit_width", C.c_uint32),
        ("n_actors", C.c_uint64),
        ("actor_base", C.c_uint64),
        ("n_contracts", C.c_uint32),
        ("slots_per_contract", C.c_uint32),
        ("storage_layout_mix", C.c_uint32),
        ("keep_full_state", C.c_uint32),
        ("n_actor_queries", C.c_uint64),
    ]


DEFAULTS = dict(
    seed=SEED_BASE + 3, parent_epoch=2992953, n_parents=2, dup_permille=50, n_receipts=1000, max_events=4,
    no_events_permille=100, n_emitters=1024, emitter_base=1000, n_sigs=16, n_subnets=64, n_planted=10,
    filter_actor=1001, variety=0, events_bit_width=5, n_actors=0, actor_base=1000, n_contracts=0,
    slots_per_contract=0, storage_layout_mix=0, keep_full_state=1, n_actor_queries=0,
)

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        lib = C.CDLL(_LIB)
        vp, u64 = C.c_void_p, C.c_uint64
        lib.synth_build.restype = vp
        lib.synth_build.argtypes = [C.POINTER(Params)]
        lib.synth_free.argtypes = [vp]
        for f in ("synth_block_count", "synth_byte_count"):
            getattr(lib, f).restype = u64
            getattr(lib, f).argtypes = [vp]
        for f in ("synth_bytes", "synth_off", "synth_len", "synth_cids"):
            getattr(lib, f).restype = vp
            getattr(lib, f).argtypes = [vp]
        lib.synth_count.restype = u64
        lib.synth_count.argtypes = [vp, C.c_int]
        lib.synth_cid.argtypes = [vp, C.c_int, vp]
        lib.synth_filter.argtypes = [vp, vp, vp]
        lib.synth_exec_order.argtypes = [vp, vp]
        lib.synth_planted.argtypes = [vp, vp]
        lib.synth_event_claims.argtypes = [vp] + [vp] * 7
        lib.synth_actor_queries.argtypes = [vp, vp, vp]
        lib.synth_storage_claims.argtypes = [vp] + [vp] * 6
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    size = n * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * size).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class Tipset:
    """A generated chain fragment.  Arrays are copies (the native object is freed in __init__)."""

    def __init__(self, **kw):
        lib = _load()
        args = dict(DEFAULTS)
        args.update(kw)
        self.params = args
        p = Params(**args)
        h = lib.synth_build(C.byref(p))
        try:
            n = lib.synth_block_count(h)
            nb = lib.synth_byte_count(h)
            self.n_blocks = int(n)
            self.data = _view(lib.synth_bytes(h), nb, np.uint8).copy()
            self.off = _view(lib.synth_off(h), n, np.uint64).copy()
            self.lens = _view(lib.synth_len(h), n, np.uint32).copy()
            self.cids = _view(lib.synth_cids(h), n * 40, np.uint8).copy().reshape(n, 40)
            cnt = lambda w: int(lib.synth_count(h, w))  # noqa: E731

            def cid(which):
                out = np.zeros(40, dtype=np.uint8)
                lib.synth_cid(h, which, _p(out))
                return out.tobytes()[:38]

            self.child_cid = cid(0)
            self.receipts_root = cid(1)
            self.state_root = cid(2)
            self.actors_root = cid(3)
            self.parent_cids = [cid(100 + k) for k in range(cnt(0))]
            self.parent_epoch = args["parent_epoch"]
            self.child_epoch = args["parent_epoch"] + 1
            t0 = np.zeros(32, np.uint8)
            t1 = np.zeros(32, np.uint8)
            lib.synth_filter(h, _p(t0), _p(t1))
            self.topic0, self.topic1 = t0.tobytes(), t1.tobytes()
            self.filter_actor = args["filter_actor"]
            ne = cnt(1)
            self.exec_order = np.zeros((ne, 40), dtype=np.uint8)
            lib.synth_exec_order(h, _p(self.exec_order))
            self.planted = np.zeros(cnt(3), dtype=np.uint64)
            lib.synth_planted(h, _p(self.planted))
            nc = cnt(2)
            self.claim_exec = np.zeros(nc, np.uint64)
            self.claim_event = np.zeros(nc, np.uint64)
            self.claim_emitter = np.zeros(nc, np.uint64)
            self.claim_ntopics = np.zeros(nc, np.uint32)
            self.claim_topics = np.zeros((nc, 4, 32), np.uint8)
            self.claim_datalen = np.zeros(nc, np.uint32)
            self.claim_data = np.zeros((nc, 64), np.uint8)
            lib.synth_event_claims(h, _p(self.claim_exec), _p(self.claim_event), _p(self.claim_emitter),
                                   _p(self.claim_ntopics), _p(self.claim_topics), _p(self.claim_datalen),
                                   _p(self.claim_data))
            nq = cnt(4)
            self.query_ids = np.zeros(nq, np.uint64)
            self.query_present = np.zeros(nq, np.uint8)
            lib.synth_actor_queries(h, _p(self.query_ids), _p(self.query_present))
            ns = cnt(5)
            self.sc_actor = np.zeros(ns, np.uint64)
            self.sc_actor_state = np.zeros((ns, 40), np.uint8)
            self.sc_storage_root = np.zeros((ns, 40), np.uint8)
            self.sc_slot = np.zeros((ns, 32), np.uint8)
            self.sc_value = np.zeros((ns, 32), np.uint8)
            self.sc_present = np.zeros(ns, np.uint8)
            lib.synth_storage_claims(h, _p(self.sc_actor), _p(self.sc_actor_state), _p(self.sc_storage_root),
                                     _p(self.sc_slot), _p(self.sc_value), _p(self.sc_present))
            self.stats = {
                "receipts_amt_bytes": cnt(6), "events_amt_bytes": cnt(7), "message_amt_bytes": cnt(8),
                "state_bytes": cnt(9), "payload_bytes": cnt(10), "blocks": self.n_blocks,
            }
        finally:
            lib.synth_free(h)

    def block(self, i: int) -> bytes:
        o = int(self.off[i])
        return self.data[o: o + int(self.lens[i])].tobytes()

    def find_block(self, cid: bytes) -> int:
        key = np.frombuffer(cid.ljust(40, b"\0"), dtype=np.uint8)
        hits = np.nonzero((self.cids == key).all(axis=1))[0]
        return int(hits[-1]) if len(hits) else -1
