"""Build ctypes claim arrays (ipcfp_event_proof_t / ipcfp_storage_proof_t, include/ipcfp.h)
from a synthetic Tipset, the way generate_event_proof / generate_storage_proof fill the
reference's structs (src/proofs/events/generator.rs:274-293, storage/generator.rs:158-178).
Shared by the oracle tests (CPU) and the GPU parity tests: both sides get the SAME structs."""
import base64
import ctypes as C

import numpy as np


class EventProof(C.Structure):
    _fields_ = [
        ("parent_epoch", C.c_int64),
        ("child_epoch", C.c_int64),
        ("parent_tipset_cids", C.POINTER(C.c_char_p)),
        ("n_parent_tipset_cids", C.c_uint32),
        ("child_block_cid", C.c_char_p),
        ("message_cid", C.c_char_p),
        ("exec_index", C.c_uint64),
        ("event_index", C.c_uint64),
        ("emitter", C.c_uint64),
        ("topics", C.POINTER(C.c_char_p)),
        ("n_topics", C.c_uint32),
        ("data", C.c_char_p),
    ]


class StorageProof(C.Structure):
    _fields_ = [
        ("child_epoch", C.c_int64),
        ("child_block_cid", C.c_char_p),
        ("parent_state_root", C.c_char_p),
        ("actor_id", C.c_uint64),
        ("actor_state_cid", C.c_char_p),
        ("storage_root", C.c_char_p),
        ("slot", C.c_char_p),
        ("value", C.c_char_p),
    ]


class EventFilter(C.Structure):
    _fields_ = [("topic0", C.c_uint8 * 32), ("topic1", C.c_uint8 * 32)]


class TrustPolicy(C.Structure):
    _fields_ = [("kind", C.c_int), ("ec_chain_empty", C.c_int), ("min_epoch", C.c_int64), ("max_epoch", C.c_int64)]


def cid_str(cid: bytes) -> str:
    """CIDv1 → multibase base32-lower string (Cid::to_string)."""
    cid = bytes(cid).rstrip(b"\0") if len(cid) == 40 else bytes(cid)
    return "b" + base64.b32encode(cid).decode().lower().rstrip("=")


def cid40_str(slot: np.ndarray) -> str:
    return cid_str(bytes(slot[:38]))


def hex0x(b: bytes) -> str:
    return "0x" + bytes(b).hex()


def make_filter(topic0: bytes, topic1: bytes) -> EventFilter:
    f = EventFilter()
    C.memmove(f.topic0, topic0, 32)
    C.memmove(f.topic1, topic1, 32)
    return f


class EventClaims:
    """Owns the Python strings behind an array of EventProof structs."""

    def __init__(self, T, indices=None):
        idx = np.arange(len(T.claim_exec)) if indices is None else np.asarray(indices)
        self.n = len(idx)
        self.arr = (EventProof * self.n)()
        self._keep = []
        parents = [cid_str(c).encode() for c in T.parent_cids]
        self.parent_arr = (C.c_char_p * len(parents))(*parents)
        child = cid_str(T.child_cid).encode()
        self._keep += [parents, child]
        for k, i in enumerate(idx):
            p = self.arr[k]
            p.parent_epoch = T.parent_epoch
            p.child_epoch = T.child_epoch
            p.parent_tipset_cids = self.parent_arr
            p.n_parent_tipset_cids = len(parents)
            p.child_block_cid = child
            e = int(T.claim_exec[i])
            msg = cid40_str(T.exec_order[e]).encode()
            nt = int(T.claim_ntopics[i])
            topics = [hex0x(T.claim_topics[i, t].tobytes()).encode() for t in range(nt)]
            tarr = (C.c_char_p * max(nt, 1))(*topics)
            data = hex0x(T.claim_data[i, : int(T.claim_datalen[i])].tobytes()).encode()
            self._keep += [msg, topics, tarr, data]
            p.message_cid = msg
            p.exec_index = e
            p.event_index = int(T.claim_event[i])
            p.emitter = int(T.claim_emitter[i])
            p.topics = tarr
            p.n_topics = nt
            p.data = data

    def set_str(self, k, field, value):
        b = value.encode() if isinstance(value, str) else value
        self._keep.append(b)
        setattr(self.arr[k], field, b)

    def set_topics(self, k, topics):
        enc = [t.encode() for t in topics]
        tarr = (C.c_char_p * max(len(enc), 1))(*enc)
        self._keep += [enc, tarr]
        self.arr[k].topics = tarr
        self.arr[k].n_topics = len(enc)

    def set_parents(self, k, parents):
        enc = [t.encode() for t in parents]
        parr = (C.c_char_p * max(len(enc), 1))(*enc)
        self._keep += [enc, parr]
        self.arr[k].parent_tipset_cids = parr
        self.arr[k].n_parent_tipset_cids = len(enc)


class StorageClaims:
    def __init__(self, T, indices=None):
        idx = np.arange(len(T.sc_actor)) if indices is None else np.asarray(indices)
        self.n = len(idx)
        self.arr = (StorageProof * self.n)()
        self._keep = []
        child = cid_str(T.child_cid).encode()
        sroot = cid_str(T.state_root).encode()
        self._keep += [child, sroot]
        for k, i in enumerate(idx):
            p = self.arr[k]
            p.child_epoch = T.child_epoch
            p.child_block_cid = child
            p.parent_state_root = sroot
            p.actor_id = int(T.sc_actor[i])
            a = cid40_str(T.sc_actor_state[i]).encode()
            s = cid40_str(T.sc_storage_root[i]).encode()
            slot = hex0x(T.sc_slot[i].tobytes()).encode()
            val = hex0x(T.sc_value[i].tobytes()).encode()
            self._keep += [a, s, slot, val]
            p.actor_state_cid = a
            p.storage_root = s
            p.slot = slot
            p.value = val

    def set_str(self, k, field, value):
        b = value.encode() if isinstance(value, str) else value
        self._keep.append(b)
        setattr(self.arr[k], field, b)
