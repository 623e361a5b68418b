"""Test-side restatement of the bundle wire format: what
`serde_json::from_str::<UnifiedProofBundle>` accepts and produces, in plain Python.

Follows src/proofs/common/bundle.rs:10-45 (ProofBlock with base64 `data`, UnifiedProofBundle),
src/proofs/events/bundle.rs:5-22 (EventData, EventProof), src/proofs/storage/bundle.rs:4-14 (StorageProof)
and the derive(Deserialize) rules: unknown fields ignored, missing / duplicate fields and wrong types are
errors; u64 / i64 fields take integer literals only ("-0", fractions, exponents, out-of-range are errors);
base64 0.21 STANDARD (canonical padding, zero trailing bits).  `Cid` ⚠: serde_json has no bytes type, so
cid 0.11's newtype-around-bytes form is an array of numbers (recollection, un-vendored crate).

Test infrastructure only: the engine's parser is independent C++ (csrc/host/json_min.h, bundle.cpp)."""
import base64
import binascii
import json
import re

import numpy as np


class BundleError(Exception):
    pass


class _Obj(list):
    """JSON object as an ordered pair list (duplicates visible)."""


_FLOATY = object()


def _int_hook(s):
    if s == "-0":
        return _FLOATY  # serde_json parses -0 as the float -0.0
    return int(s)


def _const_hook(s):
    raise BundleError(f"{s} is not JSON")


def loads(text):
    if isinstance(text, bytes):
        try:
            text = text.decode("utf-8")
        except UnicodeDecodeError as e:
            raise BundleError(str(e))
    try:
        v = json.loads(text, object_pairs_hook=_Obj, parse_int=_int_hook, parse_constant=_const_hook)
    except (json.JSONDecodeError, RecursionError) as e:
        raise BundleError(str(e))
    return v


def _has_lone_surrogate(s):
    return any(0xD800 <= ord(c) <= 0xDFFF for c in s)


def _fields(obj, names, what, depth):
    """`depth` = containers enclosing the VALUES of this object (serde_json allows 127 nested containers)."""
    if not isinstance(obj, _Obj):
        raise BundleError(f"invalid type for {what}")
    seen = {}
    for k, v in obj:
        if _has_lone_surrogate(k):
            raise BundleError("lone surrogate")
        if k in names:
            if k in seen:
                raise BundleError(f"duplicate field `{k}`")
            seen[k] = v
        else:
            _check_ignored(v, depth)
    for k in names:
        if k not in seen:
            raise BundleError(f"missing field `{k}` in {what}")
    return seen


def _check_ignored(v, depth):
    if isinstance(v, str):
        if _has_lone_surrogate(v):
            raise BundleError("lone surrogate")
    elif isinstance(v, list):  # _Obj is a list too
        if depth + 1 >= 128:
            raise BundleError("recursion limit exceeded")
        for x in v:
            if isinstance(v, _Obj):
                if _has_lone_surrogate(x[0]):
                    raise BundleError("lone surrogate")
                _check_ignored(x[1], depth + 1)
            else:
                _check_ignored(x, depth + 1)


def _u64(v):
    if isinstance(v, bool) or not isinstance(v, int) or not (0 <= v < 1 << 64):
        raise BundleError("invalid type: expected u64")
    return v


def _i64(v):
    if isinstance(v, bool) or not isinstance(v, int) or not (-(1 << 63) <= v < 1 << 63):
        raise BundleError("invalid type: expected i64")
    return v


def _str(v):
    if not isinstance(v, str) or _has_lone_surrogate(v):
        raise BundleError("invalid type: expected a string")
    return v


def _str_list(v):
    if isinstance(v, _Obj) or not isinstance(v, list):
        raise BundleError("invalid type: expected a sequence")
    return [_str(x) for x in v]


def _seq(v):
    if isinstance(v, _Obj) or not isinstance(v, list):
        raise BundleError("invalid type: expected a sequence")
    return v


_B64 = re.compile(r"^(?:[A-Za-z0-9+/]{4})*(?:[A-Za-z0-9+/]{2}==|[A-Za-z0-9+/]{3}=)?$")


def b64decode_strict(s: str) -> bytes:
    if not _B64.match(s) or "\n" in s:
        raise BundleError("invalid base64")
    raw = base64.b64decode(s, validate=True)
    if base64.b64encode(raw).decode() != s:  # non-zero trailing bits
        raise BundleError("invalid base64 (last symbol)")
    return raw


def _uvarint(b, pos):
    v = 0
    for i in range(9):
        if pos >= len(b):
            raise BundleError("varint")
        c = b[pos]
        pos += 1
        v |= (c & 0x7F) << (7 * i)
        if not c & 0x80:
            if c == 0 and i > 0:
                raise BundleError("non-minimal varint")  # unsigned-varint rejects it (as the oracle's read_varint)
            return v, pos
    raise BundleError("varint")


def cid_check(b: bytes):
    if len(b) == 34 and b[0] == 0x12 and b[1] == 0x20:
        return
    ver, pos = _uvarint(b, 0)
    if ver != 1:
        raise BundleError("cid version")
    _, pos = _uvarint(b, pos)
    _, pos = _uvarint(b, pos)
    size, pos = _uvarint(b, pos)
    if size > 64 or len(b) - pos != size:
        raise BundleError("cid multihash size")


def _cid(v):
    lst = _seq(v)
    for x in lst:
        if isinstance(x, bool) or not isinstance(x, int) or not 0 <= x <= 255:
            raise BundleError("invalid CID byte")
    b = bytes(lst)
    cid_check(b)
    return b


def parse_bundle(text, check_content=True):
    """→ dict(storage_proofs=[…], event_proofs=[…], blocks=[(cid bytes, data bytes)…]) or BundleError.
    check_content=False stops where the engine's HOST half stops: `cid` must be an array (its numbers are the
    device's business) and `data` a string whose length is a multiple of 4."""
    top = _fields(loads(text), ("storage_proofs", "event_proofs", "blocks"), "UnifiedProofBundle", 1)
    out = {"storage_proofs": [], "event_proofs": [], "blocks": []}
    for sp in _seq(top["storage_proofs"]):
        f = _fields(sp, ("child_epoch", "child_block_cid", "parent_state_root", "actor_id", "actor_state_cid",
                         "storage_root", "slot", "value"), "StorageProof", 3)
        out["storage_proofs"].append(dict(
            child_epoch=_i64(f["child_epoch"]), child_block_cid=_str(f["child_block_cid"]),
            parent_state_root=_str(f["parent_state_root"]), actor_id=_u64(f["actor_id"]),
            actor_state_cid=_str(f["actor_state_cid"]), storage_root=_str(f["storage_root"]), slot=_str(f["slot"]),
            value=_str(f["value"])))
    for ep in _seq(top["event_proofs"]):
        f = _fields(ep, ("parent_epoch", "child_epoch", "parent_tipset_cids", "child_block_cid", "message_cid",
                         "exec_index", "event_index", "event_data"), "EventProof", 3)
        d = _fields(f["event_data"], ("emitter", "topics", "data"), "EventData", 4)
        out["event_proofs"].append(dict(
            parent_epoch=_i64(f["parent_epoch"]), child_epoch=_i64(f["child_epoch"]),
            parent_tipset_cids=_str_list(f["parent_tipset_cids"]), child_block_cid=_str(f["child_block_cid"]),
            message_cid=_str(f["message_cid"]), exec_index=_u64(f["exec_index"]), event_index=_u64(f["event_index"]),
            emitter=_u64(d["emitter"]), topics=_str_list(d["topics"]), data=_str(d["data"])))
    for blk in _seq(top["blocks"]):
        f = _fields(blk, ("cid", "data"), "ProofBlock", 3)
        if check_content:
            out["blocks"].append((_cid(f["cid"]), b64decode_strict(_str(f["data"]))))
        else:
            # the host locates `[ … ]` by its first closing bracket: anything nested ends the array early and
            # what follows is a syntax error there; flat contents of any type pass
            cid = _seq(f["cid"])
            if any(isinstance(x, list) for x in cid):
                raise BundleError("nested value in cid array")
            if any(isinstance(x, str) and "]" in x for x in cid):
                raise BundleError("']' inside the cid array")
            data = _str(f["data"])
            if len(data) % 4:
                raise BundleError("base64 length")
            out["blocks"].append((None, data))
    return out


def event_dicts(T, indices=None, generated=None):
    """EventProof dicts from a synthetic tipset's claim table, or from generate_event_proofs output."""
    import claims

    c = claims.EventClaims(T, indices=indices, generated=generated)
    out = []
    for k in range(c.n):
        a = c.arr[k]
        out.append(dict(parent_epoch=a.parent_epoch, child_epoch=a.child_epoch,
                        parent_tipset_cids=[a.parent_tipset_cids[i].decode() for i in range(a.n_parent_tipset_cids)],
                        child_block_cid=a.child_block_cid.decode(), message_cid=a.message_cid.decode(),
                        exec_index=a.exec_index, event_index=a.event_index, emitter=a.emitter,
                        topics=[a.topics[i].decode() for i in range(a.n_topics)], data=a.data.decode()))
    return out


def storage_dicts(T, indices=None):
    import claims

    c = claims.StorageClaims(T, indices=indices)
    out = []
    for k in range(c.n):
        a = c.arr[k]
        out.append(dict(child_epoch=a.child_epoch, child_block_cid=a.child_block_cid.decode(),
                        parent_state_root=a.parent_state_root.decode(), actor_id=a.actor_id,
                        actor_state_cid=a.actor_state_cid.decode(), storage_root=a.storage_root.decode(),
                        slot=a.slot.decode(), value=a.value.decode()))
    return out


# ---------------------------------------------------------------------------------------------
# writer: what serde_json::to_string(&bundle) produces (field order = declaration order)
# ---------------------------------------------------------------------------------------------
def block_json(cid: bytes, data: bytes) -> str:
    return '{"cid":[%s],"data":"%s"}' % (",".join(str(x) for x in cid), base64.b64encode(data).decode())


def event_json(p) -> str:
    return json.dumps({
        "parent_epoch": p["parent_epoch"], "child_epoch": p["child_epoch"],
        "parent_tipset_cids": p["parent_tipset_cids"], "child_block_cid": p["child_block_cid"],
        "message_cid": p["message_cid"], "exec_index": p["exec_index"], "event_index": p["event_index"],
        "event_data": {"emitter": p["emitter"], "topics": p["topics"], "data": p["data"]}}, separators=(",", ":"))


def storage_json(p) -> str:
    return json.dumps({k: p[k] for k in ("child_epoch", "child_block_cid", "parent_state_root", "actor_id",
                                         "actor_state_cid", "storage_root", "slot", "value")}, separators=(",", ":"))


def bundle_json(storage, events, blocks) -> str:
    return '{"storage_proofs":[%s],"event_proofs":[%s],"blocks":[%s]}' % (
        ",".join(storage_json(p) for p in storage), ",".join(event_json(p) for p in events),
        ",".join(block_json(c, d) for c, d in blocks))


def claims_from_parsed(parsed):
    """parsed bundle → (EventClaims-like, StorageClaims-like) ctypes arrays for the oracle / engine.
    A C string cannot carry the NUL a JSON string may hold ("\\u0000"): it is replaced by 0x01, which — like NUL —
    is in no multibase / hex alphabet and equals no character of a canonical form, so every parse and every
    compare of the reference keeps its outcome (the engine's bundle lowering does the same)."""
    import ctypes as C

    import claims

    parsed = {
        "event_proofs": [{k: ([t.replace("\0", "\x01") for t in v] if isinstance(v, list) else
                              v.replace("\0", "\x01") if isinstance(v, str) else v) for k, v in p.items()}
                         for p in parsed["event_proofs"]],
        "storage_proofs": [{k: v.replace("\0", "\x01") if isinstance(v, str) else v for k, v in p.items()}
                           for p in parsed["storage_proofs"]],
    }

    class Holder:
        pass

    ev = Holder()
    ev.n = len(parsed["event_proofs"])
    ev.arr = (claims.EventProof * max(ev.n, 1))()
    ev._keep = []
    for k, p in enumerate(parsed["event_proofs"]):
        parents = [s.encode() for s in p["parent_tipset_cids"]]
        parr = (C.c_char_p * max(len(parents), 1))(*parents)
        topics = [s.encode() for s in p["topics"]]
        tarr = (C.c_char_p * max(len(topics), 1))(*topics)
        strs = [p["child_block_cid"].encode(), p["message_cid"].encode(), p["data"].encode()]
        ev._keep += [parents, parr, topics, tarr, strs]
        a = ev.arr[k]
        a.parent_epoch, a.child_epoch = p["parent_epoch"], p["child_epoch"]
        a.parent_tipset_cids, a.n_parent_tipset_cids = parr, len(parents)
        a.child_block_cid, a.message_cid, a.data = strs
        a.exec_index, a.event_index, a.emitter = p["exec_index"], p["event_index"], p["emitter"]
        a.topics, a.n_topics = tarr, len(topics)
    st = Holder()
    st.n = len(parsed["storage_proofs"])
    st.arr = (claims.StorageProof * max(st.n, 1))()
    st._keep = []
    for k, p in enumerate(parsed["storage_proofs"]):
        strs = [p[f].encode() for f in ("child_block_cid", "parent_state_root", "actor_state_cid", "storage_root", "slot",
                                        "value")]
        st._keep.append(strs)
        a = st.arr[k]
        a.child_epoch, a.actor_id = p["child_epoch"], p["actor_id"]
        (a.child_block_cid, a.parent_state_root, a.actor_state_cid, a.storage_root, a.slot, a.value) = strs
    return ev, st


def tables_from_blocks(blocks):
    """[(cid, data)…] → (data u8[], off u64[], len u32[], cids u8[n,40])"""
    lens = np.array([len(d) for _, d in blocks], dtype=np.uint32)
    off = np.zeros(len(blocks), dtype=np.uint64)
    if len(blocks):
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    data = np.frombuffer(b"".join(d for _, d in blocks), dtype=np.uint8).copy() if blocks else np.zeros(0, np.uint8)
    cids = np.zeros((len(blocks), 40), dtype=np.uint8)
    for i, (c, _) in enumerate(blocks):
        cids[i, : len(c)] = np.frombuffer(c, dtype=np.uint8)
    return data, off, lens, cids
