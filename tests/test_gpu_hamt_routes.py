"""GPU: every ROUTE of the HAMT entry points answers what the per-query walker answers, and the oracle.

`ipcfp_hamt_get*` has three routes (ipcfp_ctx_set_tuning): the per-query walker (`hamt_levels` = 0), LEVEL BY LEVEL —
every visited node decoded once (kernels/hamt_levels.hip; default for large batches, forced here for every batch size
and also with too FEW levels, so that deep queries are handed to the walker mid-tree) — and the whole-witness node
table (`hamt_table` = 1).  `ipcfp_verify_storage_proofs` has the tabled route (default for large batches) and the
one-lane route (`hamt_table` = 0).  All are driven over the same corpora: well-formed trees of three bit widths, the
synthetic state tree with wrong value types / bit widths, and witnesses with random byte flips in their blocks
(src/proofs/common/decode.rs:29-39, src/proofs/storage/decode.rs:36-97 are the semantics at stake)."""
import numpy as np
import pytest

from conftest import fuzz_seed

import claims
import pyamt
import pyhamt
from test_gpu_fuzz import idaddr, mutate
from tools.synth import Tipset

pytestmark = pytest.mark.gpu

ROUTES = [("walker", {"hamt_levels": 0, "hamt_table": -1}), ("levels", {"hamt_levels": 14, "hamt_table": -1}),
          ("levels-one-lane-parse", {"hamt_levels": 14, "hamt_table": -1, "hamt_coop": 0}),
          ("levels-2-then-walker", {"hamt_levels": 2, "hamt_table": -1}), ("levels-1", {"hamt_levels": 1, "hamt_table": -1}),
          ("table", {"hamt_levels": 0, "hamt_table": 1})]


@pytest.fixture()
def routed(engine):
    def use(cfg):
        engine.set_tuning("hamt_coop", -1)
        for k, v in cfg.items():
            engine.set_tuning(k, v)
    yield use
    engine.set_tuning("hamt_levels", -1)
    engine.set_tuning("hamt_table", -1)
    engine.set_tuning("hamt_coop", -1)


def loc_bytes(data, off, loc):
    out = []
    for l in loc:
        if l["block"] == 0xFFFFFFFF:
            out.append(b"")
        else:
            o = int(off[l["block"]]) + int(l["off"])
            out.append(data[o: o + int(l["len"])].tobytes())
    return out


@pytest.mark.parametrize("bw", [3, 5, 8])
@pytest.mark.parametrize("n", [0, 1, 40, 2000, 30000])
def test_python_written_trees_every_route(engine, oracle, routed, bw, n):
    store = pyamt.Store()
    keys = [b"\x00" + pyamt.uint(1000 + i)[0:9] + bytes([i & 0xFF, (i >> 8) & 0xFF, i >> 16]) for i in range(n)]
    items = {k: pyamt.array([pyamt.uint(i), pyamt.bstr(bytes([i & 0xFF]) * (i % 7))]) for i, k in enumerate(keys)}
    root = pyhamt.build_hamt(store, items, bit_width=bw)
    data, off, lens, cids = store.tables()
    rng = np.random.default_rng(n + bw)
    probe = [keys[int(i)] for i in rng.integers(0, max(n, 1), min(n, 3000))] + [b"nope", b"", b"\x00\xff\xff"]
    st = oracle.store(data, off, lens, cids)
    os_, ov = st.hamt_get(root, bw, "any", probe)
    st.close()
    with engine.witness(data, off, lens, cids) as w:
        for name, cfg in ROUTES:
            routed(cfg)
            gs, gl = w.hamt_get(root, bw, "any", probe)
            assert np.array_equal(gs, os_), (name, np.nonzero(gs != os_)[0][:5])
            assert loc_bytes(data, off, gl) == ov, name
    for k, s in zip(probe[:-3], os_):
        assert s == 1


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=300, n_parents=2, n_planted=3, variety=1, n_actors=60000, n_contracts=24, slots_per_contract=64,
                  storage_layout_mix=1, n_actor_queries=4000, keep_full_state=0, seed=fuzz_seed(4242))


def test_state_tree_gets_every_route_and_wrong_types(tip, engine, oracle, routed):
    keys = [idaddr(int(i)) for i in tip.query_ids] + [b"", b"\x00", bytes(40)]
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        for bw, kind in ((5, "actor_state"), (5, "vec_u8"), (5, "any"), (4, "actor_state"), (8, "actor_state"), (9, "actor_state"),
                         (0, "actor_state")):
            os_, ov = st.hamt_get(tip.actors_root, bw, kind, keys)
            for name, cfg in ROUTES:
                routed(cfg)
                gs, gl = w.hamt_get(tip.actors_root, bw, kind, keys)
                assert np.array_equal(gs, os_), (name, bw, kind, np.nonzero(gs != os_)[0][:5], gs[gs != os_][:5], os_[gs != os_][:5])
                assert loc_bytes(tip.data, tip.off, gl) == ov, (name, bw, kind)
        # a root that is not in the witness
        missing = oracle.cid_for_block(b"no such root")
        for name, cfg in ROUTES:
            routed(cfg)
            gs, _ = w.hamt_get(missing, 5, "actor_state", keys[:70])
            assert (gs == 65).all(), name
    st.close()
    present = tip.query_present.astype(bool)
    assert present.sum() > 3000 and (~present).sum() > 10


def test_mutated_witnesses_every_route(tip, engine, oracle, routed):
    """Random byte flips inside blocks (CIDs left alone: the reference's MemoryBlockstore never re-hashes): decode errors,
    bitfield / count mismatches, broken links and type errors must come out as the same status on every route."""
    rng = np.random.default_rng(fuzz_seed(77))
    keys = [idaddr(int(i)) for i in tip.query_ids]
    sc = claims.StorageClaims(tip)
    n_diff_from_clean = 0
    for it in range(24):
        data, touched = mutate(tip, rng, n_flips=3 + 5 * (it % 5))
        st = oracle.store(data, tip.off, tip.lens, tip.cids)
        os_, ov = st.hamt_get(tip.actors_root, 5, "actor_state", keys)
        want_sp = st.verify_storage_proofs(sc, mode=1)
        st.close()
        n_diff_from_clean += int((os_ >= 64).any()) + int((want_sp != 1).any())
        with engine.witness(data, tip.off, tip.lens, tip.cids) as w:
            for name, cfg in ROUTES:
                routed(cfg)
                gs, gl = w.hamt_get(tip.actors_root, 5, "actor_state", keys)
                assert np.array_equal(gs, os_), (name, it, touched, np.nonzero(gs != os_)[0][:5], gs[gs != os_][:5], os_[gs != os_][:5])
                assert loc_bytes(data, tip.off, gl) == ov, (name, it)
            for table in (0, 1, -1):  # storage proofs: one-lane route, tabled route, the default choice
                engine.set_tuning("hamt_table", table)
                got = w.verify_storage_proofs(sc.arr, sc.n)
                assert np.array_equal(got, want_sp), ("storage", table, it, touched, np.nonzero(got != want_sp)[0][:5])
    assert n_diff_from_clean > 4  # the flips really reach error paths


def actor_state(i, rng, flaw=None):
    """One ActorState `[code, state, sequence, balance, delegated_address]` in the spellings encoders write — every
    width of the sequence, TokenAmounts from empty to 20 bytes, with and without an f4 / f1 / f3 address — or with ONE
    flaw the typed decode rejects (fvm_shared ActorState)."""
    code = pyamt.link(pyamt.cid_of(b"code%d" % (i % 5)))
    state = pyamt.link(pyamt.cid_of(b"state%d" % i))
    seq = pyamt.uint([0, 7, 23, 24, 255, 256, 70000, 1 << 33][i % 8])
    mag = bytes(rng.integers(1, 256, int(rng.integers(0, 20)), dtype=np.uint8))
    bal = pyamt.bstr((bytes([i & 1]) + mag) if (len(mag) or i % 3) else b"")
    kind = i % 4
    if kind == 0:
        adr = pyamt.NULL
    elif kind == 1:
        adr = pyamt.bstr(b"\x04\x0a" + bytes(rng.integers(0, 256, 20, dtype=np.uint8)))     # f4: namespace 10 + 20 bytes
    elif kind == 2:
        adr = pyamt.bstr(b"\x01" + bytes(rng.integers(0, 256, 20, dtype=np.uint8)))         # f1
    else:
        adr = pyamt.bstr(b"\x03" + bytes(rng.integers(0, 256, 48, dtype=np.uint8)))         # f3 (49 bytes: a 0x58 header)
    if flaw == "sign":
        bal = pyamt.bstr(b"\x02\x05")
    elif flaw == "balance_len":
        bal = pyamt.bstr(b"\x00" + b"\x01" * 129)
    elif flaw == "address_proto":
        adr = pyamt.bstr(b"\x05\x01\x02")
    elif flaw == "address_len":
        adr = pyamt.bstr(b"\x01" + b"\x07" * 19)
    elif flaw == "arity":
        return pyamt.array([code, state, seq, bal])
    elif flaw == "link":
        state = pyamt.uint(5)
    return pyamt.array([code, state, seq, bal, adr])


@pytest.mark.parametrize("flaw", [None, "sign", "balance_len", "address_proto", "address_len", "arity", "link"])
def test_actor_state_spellings_and_flaws_every_route(engine, oracle, routed, flaw):
    """The sixteen-lane node decode (k_hamt_lv_parse_actor) judges ActorStates from item headers; every spelling it
    accepts, and every flaw it must leave to the item-by-item decode, comes out like the walker's and the oracle's
    verdict: one flawed value poisons its whole node (serde decodes all of it), and only the paths through that node."""
    rng = np.random.default_rng(5)
    n = 6000
    keys = [idaddr(1000 + 37 * i) for i in range(n)]
    bad_at = {} if flaw is None else {int(i): flaw for i in rng.integers(0, n, 12)}
    items = {k: actor_state(i, rng, bad_at.get(i)) for i, k in enumerate(keys)}
    store = pyamt.Store()
    root = pyhamt.build_hamt(store, items, bit_width=5)
    data, off, lens, cids = store.tables()
    probe = keys + [idaddr(5), b"", idaddr(1 << 40)]
    st = oracle.store(data, off, lens, cids)
    os_, ov = st.hamt_get(root, 5, "actor_state", probe)
    st.close()
    assert (os_[:n] == 1).sum() > n // 2 and ((os_ == 66).any() == (flaw is not None))
    with engine.witness(data, off, lens, cids) as w:
        for name, cfg in ROUTES:
            routed(cfg)
            gs, gl = w.hamt_get(root, 5, "actor_state", probe)
            assert np.array_equal(gs, os_), (name, flaw, np.nonzero(gs != os_)[0][:5], gs[gs != os_][:5], os_[gs != os_][:5])
            assert loc_bytes(data, off, gl) == ov, (name, flaw)


# ---- storage values and keys in every spelling: the plain-read paths of round 6 against the item-by-item reader -----------

def _head(b, p):
    """CBOR item head at p → (major, argument, next position)"""
    m, ai = b[p] >> 5, b[p] & 31
    if ai < 24:
        return m, ai, p + 1
    nb = 1 << (ai - 24)
    return m, int.from_bytes(b[p + 1: p + 1 + nb], "big"), p + 1 + nb


def parse_storage_node(b):
    """`[bitfield, [pointer…]]` with pointers = links or buckets of `[32-byte key, [u8…]]` → (bitfield bytes, pointers) with a
    link as bytes and a bucket as [(key, [elements])]; None when the block is anything else."""
    try:
        m, a, p = _head(b, 0)
        if (m, a) != (4, 2):
            return None
        m, a, p = _head(b, p)
        if m != 2 or a > 8:
            return None
        bf, p = b[p: p + a], p + a
        m, npt, p = _head(b, p)
        if m != 4 or npt > 32:
            return None
        ptrs = []
        for _ in range(npt):
            if b[p] == 0xD8:
                if b[p: p + 5] != b"\xd8\x2a\x58\x27\x00":
                    return None
                ptrs.append(bytes(b[p: p + 43]))
                p += 43
                continue
            m, nkv, p = _head(b, p)
            if m != 4:
                return None
            bucket = []
            for _ in range(nkv):
                m, a, p = _head(b, p)
                if (m, a) != (4, 2):
                    return None
                m, kl, p = _head(b, p)
                if m != 2 or kl != 32:
                    return None
                key, p = bytes(b[p: p + 32]), p + 32
                m, n, p = _head(b, p)
                if m != 4:
                    return None
                el = []
                for _ in range(n):
                    m, v, p = _head(b, p)
                    if m != 0 or v > 255:
                        return None
                    el.append(v)
                bucket.append((key, el))
            ptrs.append(bucket)
        return (bytes(bf), ptrs) if p == len(b) else None
    except IndexError:
        return None


def wide_head(major, n, nbytes):
    return bytes([(major << 5) | {1: 24, 2: 25, 4: 26}[nbytes]]) + int(n).to_bytes(nbytes, "big")


def respell_entry(key, el, mode, rng):
    """One bucket entry, its key and value spelled by `mode`; → (bytes, the elements a decoder reads or None if it must fail)"""
    el = list(el)
    khead, vhead, spell = None, None, {}
    if mode == 1:
        vhead = wide_head(4, len(el), 2)                       # 99 00 nn
    elif mode == 2 and el:
        spell[int(rng.integers(0, len(el)))] = 2                # one element as 19 00 xx
    elif mode == 3 and el:
        k = int(rng.integers(0, len(el)))
        el[k] = int(rng.integers(0, 24))
        spell[k] = 1                                            # a small element in two bytes (18 0x)
    elif mode == 4:
        el = el[: [0, 1, 2, 3, 5, 23, 24, 31][int(rng.integers(0, 8))]]   # shorter than a word: left-padded
    elif mode == 5:
        el = [int(x) for x in rng.integers(0, 256, int(rng.integers(1, 9)))] + el   # longer: the last 32 count
    elif mode == 6:
        khead = wide_head(2, 32, 2)                             # 59 00 20
    elif mode == 7:
        el = [24] * len(el)                                     # 18 18 18 18 …: headers and payloads look alike
    elif mode == 8:
        el[-1:] = [300]                                         # not a u8: the typed decode fails
        spell[len(el) - 1] = 2
    elif mode == 9 and el:
        el = el[:-1] + [0x17, 0x18, 0x19][: 1 + int(rng.integers(0, 3))]  # the three kinds of first byte side by side
    out = (khead or pyamt.head(2, 32)) + key + (vhead or pyamt.head(4, len(el)))
    for k, v in enumerate(el):
        out += wide_head(0, v, spell[k]) if k in spell else pyamt.uint(v)
    return out, (None if mode == 8 else el)


def left_pad_32(el):
    b = bytes(el)
    return b[-32:] if len(b) >= 32 else bytes(32 - len(b)) + b


@pytest.mark.parametrize("slots", [40, 300])
def test_storage_values_in_every_spelling_both_routes(engine, oracle, slots):
    """Storage nodes rewritten in place (their CIDs kept: the store never re-hashes) so that keys and values come in every
    spelling a decoder accepts — wide heads, wide and non-minimal elements, values shorter and longer than 32 elements, runs of
    0x18 — and one it rejects.  The tabled route reads the usual spelling with plain 8-byte loads (walk_dev.h
    bucket_find32_raw, verify_storage.hip left_pad_32_raw, cbor_dev.h vec_u8_end) and everything else item by item; the
    one-lane route reads everything item by item: both must say what the oracle says, claim by claim."""
    rng = np.random.default_rng(fuzz_seed(4242) + slots)
    tip = Tipset(n_receipts=8, n_parents=1, n_actors=300, n_contracts=12, slots_per_contract=slots, storage_layout_mix=0,
                 keep_full_state=1, seed=fuzz_seed(31) + slots)
    blocks = [bytes(tip.data[int(o): int(o) + int(l)]) for o, l in zip(tip.off, tip.lens)]
    modes, twice = {}, set()
    new_value = {}   # slot → the elements its entry now holds (None: undecodable)
    n_nodes = 0
    for bi, b in enumerate(blocks):
        node = parse_storage_node(b)
        if node is None or not any(isinstance(p, list) and p for p in node[1]):
            continue
        n_nodes += 1
        if n_nodes % 4 == 0:
            continue  # every fourth node stays as written
        ptrs = []
        for p in node[1]:
            if not isinstance(p, list):
                ptrs.append(p)
                continue
            ents = []
            for key, el in p:
                mode = int(rng.integers(0, 10)) if rng.integers(0, 3) else 0
                if mode == 8 and rng.integers(0, 16):
                    mode = 0  # (a flawed value takes its whole node down: keep them rare)
                if n_nodes == 2 and not ents and not any(isinstance(q, bytes) and q[:1] != b"\xd8" for q in ptrs):
                    mode = 8  # … but have one
                e, now = respell_entry(key, el, mode, rng)
                ents.append(b"\x82" + e)
                ident = (key, left_pad_32(el))  # (contracts share slot numbers: an entry is its slot and what it held)
                if ident in new_value or ident in twice:  # … and two contracts may hold the same value in the same slot
                    twice.add(ident)
                    new_value.pop(ident, None)
                    modes.pop(ident, None)
                else:
                    new_value[ident] = now
                    modes[ident] = mode
            bucket_head = wide_head(4, len(ents), 1) if rng.integers(0, 8) == 0 else pyamt.head(4, len(ents))
            ptrs.append(bucket_head + b"".join(ents))
        blocks[bi] = pyamt.head(4, 2) + pyamt.bstr(node[0]) + pyamt.head(4, len(ptrs)) + b"".join(ptrs)
    assert n_nodes >= 12
    lens = np.array([len(b) for b in blocks], dtype=tip.lens.dtype)
    off = np.zeros(len(blocks), dtype=tip.off.dtype)
    off[1:] = np.cumsum(lens)[:-1]
    data = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()
    sc = claims.StorageClaims(tip)
    n_set = 0
    for k in range(sc.n):  # two claims of three follow their entry's new value: TRUE where the value still decodes
        now = new_value.get((tip.sc_slot[k].tobytes(), tip.sc_value[k].tobytes()))
        if now is not None and k % 3:
            sc.set_str(k, "value", "0x" + left_pad_32(now).hex())
            n_set += 1
    st = oracle.store(data, off, lens, tip.cids)
    want = st.verify_storage_proofs(sc, mode=1)
    st.close()
    for m in range(10):  # every decodable spelling verifies against the value it now spells
        ks = [k for k in range(sc.n) if k % 3 and modes.get((tip.sc_slot[k].tobytes(), tip.sc_value[k].tobytes())) == m and want[k] < 64]
        assert m == 8 or (len(ks) > 0 and (want[ks] == 1).all()), (m, len(ks), want[ks][:10])
    assert (want == 1).sum() > sc.n // 4 and (want == 21).any() and (want >= 64).any(), np.unique(want, return_counts=True)
    with engine.witness(data, off, lens, tip.cids) as w:
        for table in (1, 0, -1):
            engine.set_tuning("hamt_table", table)
            got = w.verify_storage_proofs(sc.arr, sc.n)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (table, bad[:8], got[bad[:8]], want[bad[:8]])
    engine.set_tuning("hamt_table", -1)
