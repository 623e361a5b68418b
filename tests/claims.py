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


def _cbor_item(b: bytes, pos: int):
    """Minimal DAG-CBOR reader for test fixtures: returns (value, next_pos); uints, bytes, text, arrays only."""
    ib = b[pos]
    major, info = ib >> 5, ib & 31
    pos += 1
    if info < 24:
        arg = info
    else:
        nb = {24: 1, 25: 2, 26: 4, 27: 8}[info]
        arg = int.from_bytes(b[pos:pos + nb], "big")
        pos += nb
    if major == 0:
        return arg, pos
    if major == 2:
        return bytes(b[pos:pos + arg]), pos + arg
    if major == 3:
        return bytes(b[pos:pos + arg]).decode(), pos + arg
    if major == 4:
        out = []
        for _ in range(arg):
            v, pos = _cbor_item(b, pos)
            out.append(v)
        return out, pos
    raise ValueError(f"unexpected CBOR major type {major}")


def extract_evm_log(stamped_event: bytes):
    """`extract_evm_log` (src/proofs/common/evm.rs:13-59) on a StampedEvent's bytes → (emitter, topics, data) or None."""
    (emitter, entries), _ = _cbor_item(stamped_event, 0)
    kv = {}
    for _flags, key, _codec, value in entries:
        kv[key] = value  # a repeated key keeps the last value
    if "topics" in kv:
        t = kv["topics"]
        if len(t) % 32:
            return None
        return emitter, [t[i:i + 32] for i in range(0, len(t), 32)], kv.get("data", b"")
    topics = []
    for k in ("t1", "t2", "t3", "t4"):
        if k not in kv:
            break
        if len(kv[k]) != 32:
            return None
        topics.append(kv[k])
    if not topics:
        return None
    return emitter, topics, kv.get("d", b"")


class EventClaims:
    """Owns the Python strings behind an array of EventProof structs."""

    def __init__(self, T, indices=None, generated=None):
        """From the synthetic tipset's own claim table (default), or — `generated=(matches, message_cids)` —
        from what ipcfp_generate_event_proofs returned, the way generate_event_proof fills EventProof
        (src/proofs/events/generator.rs:274-293): topics/data are read back from the located event."""
        if generated is not None:
            matches, msg_cids = generated
            rows = []
            for m, mc in zip(matches, msg_cids):
                o = int(T.off[m["block"]]) + int(m["off"])
                log = extract_evm_log(T.data[o:o + int(m["len"])].tobytes())
                assert log is not None and log[0] == int(m["emitter"])
                rows.append((int(m["exec_index"]), int(m["event_index"]), int(m["emitter"]), cid40_str(mc), log[1], log[2]))
        else:
            idx = np.arange(len(T.claim_exec)) if indices is None else np.asarray(indices)
            rows = []
            for i in idx:
                e = int(T.claim_exec[i])
                nt = int(T.claim_ntopics[i])
                rows.append((e, int(T.claim_event[i]), int(T.claim_emitter[i]), cid40_str(T.exec_order[e]),
                             [T.claim_topics[i, t].tobytes() for t in range(nt)],
                             T.claim_data[i, : int(T.claim_datalen[i])].tobytes()))
        self.n = len(rows)
        self.arr = (EventProof * max(self.n, 1))()
        self._keep = []
        parents = [cid_str(c).encode() for c in T.parent_cids]
        self.parent_arr = (C.c_char_p * len(parents))(*parents)
        child = cid_str(T.child_cid).encode()
        self._keep += [parents, child]
        for k, (e, ev, emitter, msg_s, topic_list, data_b) in enumerate(rows):
            p = self.arr[k]
            p.parent_epoch = T.parent_epoch
            p.child_epoch = T.child_epoch
            p.parent_tipset_cids = self.parent_arr
            p.n_parent_tipset_cids = len(parents)
            p.child_block_cid = child
            msg = msg_s.encode()
            nt = len(topic_list)
            topics = [hex0x(t).encode() for t in topic_list]
            tarr = (C.c_char_p * max(nt, 1))(*topics)
            data = hex0x(data_b).encode()
            self._keep += [msg, topics, tarr, data]
            p.message_cid = msg
            p.exec_index = e
            p.event_index = ev
            p.emitter = emitter
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
