"""CPU: the PARALLEL outline of a state-tree HAMT node (csrc/kernels/hamt_outline.h — anchors on `85 d8 2a`, entries parsed
forward, gaps walked, the pieces must tile the node) against the sequential reader of the same header, both compiled for
the host (tests/native/outline_harness.cpp simulates the 32 lanes of k_hamt_lv_parse_actor in lock step).
What must hold: whatever the parallel outline accepts, the sequential one accepts WITH THE SAME RECORD (pointer offsets,
entry offsets, link mask, bitfield) — the kernel falls back to the sequential reader for everything else, so this is what
makes the record independent of the route.  Honest nodes must take the parallel route (that is the speed-up), and the three
anchor bytes planted inside a digest or a balance must push a node off it, not through it.
Reference: `Hamt::get` decodes every node on the path completely, src/proofs/common/decode.rs:29-39."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tools.synth import Tipset

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "outline_harness.cpp")
LIB = os.path.join(HERE, "native", "liboutline_harness.so")
HDR = os.path.join(HERE, "..", "ipc-filecoin-proofs_amd", "csrc", "kernels", "hamt_outline.h")


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC], check=True)
    lib = C.CDLL(LIB)
    vp = C.c_void_p
    lib.outline_seq.restype = C.c_int
    lib.outline_seq.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.outline_par.restype = C.c_int
    lib.outline_par.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    return lib


def run(lib, node: np.ndarray, seed=1):
    def bufs():
        return (np.zeros(6, np.uint32), np.zeros(32, np.uint16), np.zeros(100, np.uint16), np.zeros(100, np.uint16),
                np.zeros(100, np.uint16), np.zeros(100, np.uint16), np.zeros(100, np.uint8), np.full(32, 255, np.uint8))
    node = np.ascontiguousarray(node, dtype=np.uint8)
    so, sp, sv, s2, sa, se, sk, sf = bufs()
    po, pp, pv, p2, pa, pe, pk, pf = bufs()
    na = C.c_uint32()
    s_ok = lib.outline_seq(node.ctypes.data, len(node), seed, so.ctypes.data, sp.ctypes.data, sv.ctypes.data, s2.ctypes.data, sa.ctypes.data,
                           se.ctypes.data, sk.ctypes.data, sf.ctypes.data)
    p_ok = lib.outline_par(node.ctypes.data, len(node), seed, po.ctypes.data, pp.ctypes.data, pv.ctypes.data, p2.ctypes.data, pa.ctypes.data,
                           C.byref(na), pe.ctypes.data, pk.ctypes.data, pf.ctypes.data)

    def rec(o, p, v, l2, a, en, kl, fo):
        # (np, ne, links, bitfield, pointer offsets, entry: ActorState / second link / address offsets; then the entry table's
        # extras: entry ends, key lengths, each bucket pointer's first entry — 255 where the pointer is no bucket with entries)
        np_, ne = int(o[0]), int(o[1])
        return (np_, ne, int(o[2]), int(o[3]), int(o[4]), p[:np_].tolist(), v[:ne].tolist(), l2[:ne].tolist(), a[:ne].tolist(),
                en[:ne].tolist(), kl[:ne].tolist(), fo[:np_].tolist())
    return bool(s_ok), rec(so, sp, sv, s2, sa, se, sk, sf), bool(p_ok), rec(po, pp, pv, p2, pa, pe, pk, pf), int(na.value)


@pytest.fixture(scope="module")
def state_nodes():
    T = Tipset(n_receipts=8, n_planted=0, n_actors=40_000, n_contracts=6, slots_per_contract=8, storage_layout_mix=1,
               keep_full_state=1, n_actor_queries=200, seed=515)
    return [T.data[int(o): int(o) + int(l)] for o, l in zip(T.off, T.lens) if 3 <= l <= 6800]


def test_honest_nodes_take_the_parallel_route_with_the_sequential_record(harness, state_nodes):
    seq_ok = par_ok = entries = 0
    for i, node in enumerate(state_nodes):
        s_ok, s_rec, p_ok, p_rec, na = run(harness, node, seed=i)
        assert not p_ok or (s_ok and p_rec == s_rec), i
        seq_ok += s_ok
        par_ok += p_ok
        entries += s_rec[1] if s_ok else 0
    assert seq_ok > 1000 and entries > 30_000          # the state tree's nodes are there, buckets and all
    assert par_ok >= seq_ok - max(2, seq_ok // 500)     # … and (all but a stray anchor in a digest) take the parallel route


def test_mutated_nodes_never_get_another_record(harness, state_nodes):
    rng = np.random.default_rng(77)
    big = [n for n in state_nodes if len(n) > 600][:400]
    checked = accepted = 0
    for k, node in enumerate(big):
        for _ in range(25):
            m = node.copy()
            how = rng.integers(0, 4)
            if how == 0:
                m[rng.integers(0, len(m))] ^= 1 << rng.integers(0, 8)
            elif how == 1:
                m[rng.integers(0, len(m))] = rng.choice([0x82, 0x85, 0xd8, 0x2a, 0x80, 0x81, 0x58, 0x40, 0xf6, 0x98])
            elif how == 2:
                m = m[: rng.integers(3, len(m))]                        # cut short
            else:
                at = rng.integers(0, len(m) - 3)
                m[at: at + 3] = (0x85, 0xd8, 0x2a)                      # a stray anchor
            s_ok, s_rec, p_ok, p_rec, _ = run(harness, m, seed=k)
            assert not p_ok or (s_ok and p_rec == s_rec)
            checked += 1
            accepted += p_ok
    assert checked == 10_000 and accepted > 100


def test_stray_anchor_bytes_are_dropped_or_decline_never_another_record(harness, state_nodes):
    hit = 0
    for k, node in enumerate(n for n in state_nodes if len(n) > 600):
        s_ok, s_rec, p_ok, p_rec, na = run(harness, node, seed=k)
        if not (s_ok and p_ok and s_rec[1] >= 3):
            continue
        # the three bytes inside the digest of the second entry's `state` link: an anchor that does not parse as an entry
        # (a digest is opaque to the decoder) — dropped, the node keeps the parallel route and its record
        l2 = s_rec[7][1]
        m = node.copy()
        m[l2 + 20: l2 + 23] = (0x85, 0xd8, 0x2a)
        s2_ok, s2_rec, p2_ok, p2_rec, na2 = run(harness, m, seed=k)
        assert s2_ok and s2_rec == s_rec and na2 == na + 1
        assert p2_ok and p2_rec == s_rec
        hit += 1
        if hit >= 50:
            break
    assert hit >= 20


def test_a_whole_entry_hidden_in_a_balance_declines(harness):
    """An anchor that DOES parse — the bytes of a complete ActorState inside a 100-byte balance — is kept, and then the
    pieces no longer tile the node: the parallel route declines (the kernel reads the node front to back), the sequential
    reader sees an ordinary entry with a long balance."""
    link = bytes.fromhex("d82a58270001 71a0e40220") + bytes(range(32))
    code = bytes.fromhex("d82a52000155000d") + b"fil/12/actor3"
    fake = b"\x85" + code + link + b"\x05" + b"\x41\x00" + b"\xf6"
    bal = b"\x00" + fake + bytes(100 - 1 - len(fake))
    assert len(bal) == 100

    def entry(key, balance):
        bh = bytes([0x40 + len(balance)]) if len(balance) < 24 else bytes([0x58, len(balance)])
        return b"\x82" + bytes([0x40 + len(key)]) + key + b"\x85" + code + link + b"\x05" + bh + balance + b"\xf6"

    node = b"\x82\x41\x03\x82" + b"\x82" + entry(b"\x00\x01", b"\x00\x07") + entry(b"\x00\x02", bal) + link
    n = np.frombuffer(node, dtype=np.uint8)
    s_ok, s_rec, p_ok, p_rec, na = run(harness, n)
    assert s_ok and s_rec[0] == 2 and s_rec[1] == 2 and na == 3 and not p_ok
    honest = b"\x82\x41\x03\x82" + b"\x82" + entry(b"\x00\x01", b"\x00\x07") + entry(b"\x00\x02", b"\x00" + bytes(99)) + link
    s_ok, s_rec, p_ok, p_rec, na = run(harness, np.frombuffer(honest, dtype=np.uint8))
    assert s_ok and p_ok and s_rec == p_rec and na == 2


def test_hand_made_shapes(harness):
    link = bytes.fromhex("d82a58270001 71a0e40220") + bytes(range(32))
    code = bytes.fromhex("d82a52000155000d") + b"fil/12/actor3"
    assert len(code) == 3 + 0x12

    def entry(key: bytes, seq=5, bal=b"\x00\x01\x02", addr=None):
        a = b"\xf6" if addr is None else bytes([0x40 + len(addr)]) + addr
        kh = bytes([0x40 + len(key)]) if len(key) < 24 else bytes([0x58, len(key)])
        return b"\x82" + kh + key + b"\x85" + code + link + bytes([seq]) + bytes([0x40 + len(bal)]) + bal + a

    def node(pointers, bf=b"\x0f"):
        return b"\x82" + bytes([0x40 + len(bf)]) + bf + bytes([0x80 + len(pointers)]) + b"".join(pointers)

    def bucket(entries):
        return bytes([0x80 + len(entries)]) + b"".join(entries)

    shapes = {
        "links only": node([link, link, link]),
        "one bucket": node([bucket([entry(b"\x00\x05"), entry(b"\x00\x06", addr=b"\x00\x07")])]),
        "link, bucket, empty bucket, link": node([link, bucket([entry(b"\x00\x05")]), b"\x80", link]),
        "empty buckets first and last": node([b"\x80", bucket([entry(b"\x00\x01"), entry(b"\x00\x02"), entry(b"\x00\x03")]), b"\x80"]),
        "two buckets back to back": node([bucket([entry(b"\x00\x01")]), bucket([entry(b"\x00\x02"), entry(b"\x00\x03")])]),
        "long key (58 form)": node([bucket([entry(bytes(30))])]),
        "no pointers": node([]),
    }
    for name, b in shapes.items():
        s_ok, s_rec, p_ok, p_rec, _ = run(harness, np.frombuffer(b, dtype=np.uint8))
        assert s_ok and p_ok and s_rec == p_rec, name
    wrong = {
        "count one too many": node([b"\x83" + entry(b"\x00\x01") + entry(b"\x00\x02")]),
        "count one too few": node([b"\x81" + entry(b"\x00\x01") + entry(b"\x00\x02")]),
        "entry behind a link without a header": node([link, entry(b"\x00\x01")])[:],
        "pointer count too small": b"\x82\x41\x0f\x81" + link + link,
        "bytes behind the node": node([link]) + b"\x00",
        "entry first without a header": node([entry(b"\x00\x01")]),
    }
    for name, b in wrong.items():
        s_ok, _, p_ok, _, _ = run(harness, np.frombuffer(b, dtype=np.uint8))
        assert not s_ok and not p_ok, name
