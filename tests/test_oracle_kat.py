"""CPU: pin the oracle's wire-format layer with the known answers of SURVEY.md Appendix B
(well-known Filecoin CIDs, the demo's own constants from /root/reference/src/main.rs:38,61-62,
and HAMT hash-bit sequences).  The reference itself holds no tests (parity unpinned by it)."""
import hashlib

import numpy as np
import pytest

import pymini

EMPTY_AMT_V0 = bytes.fromhex("8300008341008080")
EMPTY_AMT_V0_CID = "bafy2bzacedswlcz5ddgqnyo3sak3jmhmkxashisnlpq6ujgyhe4mlobzpnhs6"

CID_KATS = [
    (bytes.fromhex("80"), "bafy2bzacebc3bt6cedhoyw34drrmjvazhu4oj25er2ebk4u445pzycvq4ta4a"),  # EMPTY_ARR_CID
    (EMPTY_AMT_V0, EMPTY_AMT_V0_CID),
    (bytes.fromhex("840300008341008080"), "bafy2bzacedijw74yui7otvo63nfl3hdq2vdzuy7wx2tnptwed6zml4vvz7wee"),
    (bytes.fromhex("8405000083440000000080" "80"), "bafy2bzaceacu3yonapahihxmnhzuvk76yupwjmyeymd2l5n6xfs5st52shm6a"),
    (bytes.fromhex("824080"), "bafy2bzaceamp42wmmgr2g2ymg46euououzfyck7szknvfacqscohrvaikwfay"),  # empty HAMT
]


@pytest.mark.parametrize("block,cid_str", CID_KATS)
def test_wellknown_cids(oracle, block, cid_str):
    cid = oracle.cid_for_block(block)
    assert cid[:6] == bytes.fromhex("0171a0e40220")
    assert cid[6:] == hashlib.blake2b(block, digest_size=32).digest()
    assert oracle.cid_to_string(cid) == cid_str
    assert oracle.cid_from_string(cid_str) == cid


def test_empty_txmeta_cid_exercises_the_43_byte_link(oracle):
    amt = oracle.cid_for_block(EMPTY_AMT_V0)
    link = bytes.fromhex("d82a582700") + amt
    assert len(link) == 43
    txmeta = b"\x82" + link + link
    assert len(txmeta) == 87
    assert oracle.cid_to_string(oracle.cid_for_block(txmeta)) == \
        "bafy2bzacecmda75ovposbdateg7eyhwij65zklgyijgcjwynlklmqazpwlhba"


def test_cid_string_forms(oracle):
    cid = oracle.cid_for_block(b"\x80")
    s = oracle.cid_to_string(cid)
    assert oracle.cid_from_string(s.upper().replace("B", "B", 1)) == cid  # base32 upper multibase 'B'
    assert oracle.cid_from_string("f" + cid.hex()) == cid                  # base16 multibase
    assert oracle.cid_from_string("bafy2bzace!!!") is None
    assert oracle.cid_from_string("") is None
    assert oracle.cid_from_string(s[:-1]) is None                          # truncated digest
    # CIDv0 (sha2-256 dag-pb) round trip: "Qm…" 46 chars
    v0 = bytes([0x12, 0x20]) + hashlib.sha256(b"x").digest()
    qs = oracle.cid_to_string(v0)
    assert qs.startswith("Qm") and len(qs) == 46 and oracle.cid_from_string(qs) == v0


def ascii_to_bytes32(s: str) -> bytes:
    b = s.encode()[:32]
    return b + b"\0" * (32 - len(b))


def mapping_slot(keccak, key32: bytes, index: int) -> bytes:
    return keccak(key32 + index.to_bytes(32, "big"))


def test_demo_constants(oracle):
    """main.rs:61 event signature, main.rs:62 topic1, main.rs:38 storage slot."""
    for k in (oracle.keccak256, pymini.keccak256):
        assert k(b"NewTopDownMessage(bytes32,uint256)").hex() == \
            "43a056172fcece714f70dd1570a9c9152899b8f299e7924d5e817217ee7783bf"
        key = ascii_to_bytes32("calib-subnet-1")
        assert key.hex() == "63616c69622d7375626e65742d31" + "00" * 18
        assert mapping_slot(k, key, 0).hex() == "085f9ee70231a68f18fd95b4fbe6ab1d2e6617d2d5454895bab0176f50037272"
        assert mapping_slot(k, key, 7).hex() == "576cef9dda367d57a31c0385b176da3b8dcdddaba86ed56dccd98bb9b488ad3e"


def hamt_indices(digest: bytes, bw=5, n=8):
    bits = "".join(f"{b:08b}" for b in digest)
    return [int(bits[i * bw:(i + 1) * bw], 2) for i in range(n)]


@pytest.mark.parametrize("key,prefix,suffix,idx", [
    (bytes.fromhex("0000"), "96a296d2", "cfc7", [18, 26, 17, 9, 13, 20, 17, 4]),
    (bytes.fromhex("0001"), "b413f47d", "c8d2", [22, 16, 9, 31, 8, 31, 8, 19]),
    (bytes.fromhex("00d209"), "2a96470f", "9140", [5, 10, 11, 4, 14, 3, 28, 28]),
    (bytes.fromhex("008092f401"), "6658e32a", "ea52", [12, 25, 12, 14, 6, 10, 20, 15]),
])
def test_hamt_key_hash_bits(oracle, key, prefix, suffix, idx):
    d = oracle.sha256(key)
    assert d.hex().startswith(prefix) and d.hex().endswith(suffix)
    assert hamt_indices(d) == idx


def test_empty_structures_walk(oracle):
    """get on the empty AMT v0 / v3 / HAMT blocks → None; missing root → Err(missing block)."""
    blocks = [EMPTY_AMT_V0, bytes.fromhex("840300008341008080"), bytes.fromhex("824080")]
    cids = [oracle.cid_for_block(b) for b in blocks]
    lens = np.array([len(b) for b in blocks], dtype=np.uint32)
    off = np.zeros(3, dtype=np.uint64)
    off[1:] = np.cumsum(lens[:-1])
    data = np.frombuffer(b"".join(blocks), dtype=np.uint8)
    c40 = np.zeros((3, 40), dtype=np.uint8)
    for i, c in enumerate(cids):
        c40[i, :38] = np.frombuffer(c, dtype=np.uint8)
    st = oracle.store(data, off, lens, c40)
    s, _ = st.amt_get(cids[0], 0, "cid", [0, 7, 8, 2 ** 63])
    assert s.tolist() == [32, 32, 32, 32]
    s, _ = st.amt_get(cids[1], 3, "stamped_event", [0, 5])
    assert s.tolist() == [32, 32]
    s, _ = st.amt_get(cids[1], 0, "cid", [0])           # v3 root read as v0 → arity error
    assert s.tolist() == [66]
    s, _ = st.hamt_get(cids[2], 5, "actor_state", [bytes.fromhex("0001")])
    assert s.tolist() == [32]
    missing = oracle.cid_for_block(b"nope")
    s, _ = st.amt_get(missing, 0, "cid", [0])
    assert s.tolist() == [65]
    s, _ = st.hamt_get(missing, 5, "actor_state", [b"\x00\x01"])
    assert s.tolist() == [65]
