"""CPU: the transport form of EventProof claims (include/ipcfp.h ipcfp_event_claim_compact_t) — the host-side converter
`ipcfp_compact_event_claims` against a pure-Python expansion: compact → plain must give back the plain lowering byte for
byte (claims with their blob segments in claim order), and everything the compact record cannot hold must be refused, not
truncated.  The device expansion is held to the same bytes in tests/test_gpu_transport.py.
Reference fields: src/proofs/events/bundle.rs:5-23."""
import numpy as np
import pytest

import ipc_filecoin_proofs_amd as ipcfp

STD = np.frombuffer(bytes.fromhex("0171a0e40220"), dtype=np.uint8)


def random_claims(rng, n, n_groups=3):
    cl = np.zeros(n, dtype=ipcfp.CLAIM_DTYPE)
    g = rng.integers(0, n_groups, n)
    cl["parent_epoch"] = 1000 + g
    cl["child_epoch"] = 1001 + g
    cl["tipset"] = g % 2
    cl["exec_index"] = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    cl["event_index"] = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    cl["emitter"] = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    cl["flags"] = rng.integers(0, 4, n)
    parsed = (cl["flags"] & 1) != 0
    cl["message_cid"][parsed, :6] = STD
    cl["message_cid"][parsed, 6:38] = rng.integers(0, 256, (int(parsed.sum()), 32), dtype=np.uint8)
    nt = rng.integers(0, 9, n)
    dl = rng.integers(0, 300, n)
    dl[rng.integers(0, n, max(1, n // 10))] = 0
    sizes = nt * 33 + dl
    starts = np.zeros(n, dtype=np.int64)
    starts[1:] = np.cumsum(sizes[:-1])
    total = int(sizes.sum())
    blob = rng.integers(0, 256, total + 64, dtype=np.uint8)
    for i in range(n):
        for t in range(nt[i]):
            blob[starts[i] + 33 * t] = rng.integers(0, 2)
    cl["n_topics"], cl["topics_off"], cl["data_off"], cl["data_len"] = nt, starts, starts + nt * 33, dl
    return cl, blob, total


def expand_ref(groups, cc, cblob):
    """compact → plain, the way kernels/claims_compact.hip does it"""
    n = len(cc)
    cl = np.zeros(n, dtype=ipcfp.CLAIM_DTYPE)
    out = bytearray()
    at = 0
    for i in range(n):
        c = cc[i]
        nt, dl = int(c["n_topics"]), int(c["data_len"])
        gi = groups[int(c["group"])]
        cl[i]["parent_epoch"], cl[i]["child_epoch"], cl[i]["tipset"] = gi["parent_epoch"], gi["child_epoch"], gi["tipset"]
        cl[i]["exec_index"], cl[i]["event_index"], cl[i]["emitter"] = c["exec_index"], c["event_index"], c["emitter"]
        if c["flags"] & 1:
            cl[i]["message_cid"][:6] = STD
            cl[i]["message_cid"][6:38] = c["message_digest"]
        cl[i]["flags"], cl[i]["n_topics"], cl[i]["data_len"] = c["flags"], nt, dl
        cl[i]["topics_off"], cl[i]["data_off"] = len(out), len(out) + 33 * nt
        for t in range(nt):
            out.append((int(c["topic_flags"]) >> t) & 1)
            out += cblob[at + 32 * t: at + 32 * t + 32].tobytes()
        out += cblob[at + 32 * nt: at + 32 * nt + dl].tobytes()
        at += 32 * nt + dl
    return cl, np.frombuffer(bytes(out), dtype=np.uint8), at


@pytest.mark.parametrize("seed,n", [(1, 1), (2, 17), (3, 2000)])
def test_round_trip_is_byte_exact(seed, n):
    rng = np.random.default_rng(seed)
    cl, blob, blob_len = random_claims(rng, n)
    # an unparsed message CID is not carried: its bytes are not part of the claim's meaning
    cl["message_cid"][(cl["flags"] & 1) == 0] = 0
    groups, cc, cblob, cblob_len = ipcfp.compact_event_claims(cl, blob, blob_len)
    assert cblob_len == int((cl["n_topics"].astype(np.int64) * 32 + cl["data_len"]).sum())
    assert cc.nbytes + cblob_len < cl.nbytes + blob_len
    back, bblob, used = expand_ref(groups, cc, cblob)
    assert used == cblob_len
    assert back.tobytes() == cl.tobytes()
    assert bblob.tobytes() == blob[:blob_len].tobytes()


def test_blob_segments_in_any_order_come_back_in_claim_order():
    rng = np.random.default_rng(9)
    cl, blob, blob_len = random_claims(rng, 50)
    cl["message_cid"][(cl["flags"] & 1) == 0] = 0
    # the plain form may place a claim's segments anywhere in the blob (here: data in front of the topics, reversed claims)
    sizes = cl["n_topics"].astype(np.int64) * 33 + cl["data_len"]
    new = np.zeros(blob_len + 64, dtype=np.uint8)
    at = 0
    moved = cl.copy()
    for i in reversed(range(len(cl))):
        nt, dl = int(cl["n_topics"][i]), int(cl["data_len"][i])
        new[at: at + dl] = blob[cl["data_off"][i]: cl["data_off"][i] + dl]
        new[at + dl: at + dl + 33 * nt] = blob[cl["topics_off"][i]: cl["topics_off"][i] + 33 * nt]
        moved["data_off"][i], moved["topics_off"][i] = at, at + dl
        at += int(sizes[i])
    groups, cc, cblob, cblob_len = ipcfp.compact_event_claims(moved, new, blob_len)
    back, bblob, _ = expand_ref(groups, cc, cblob)
    assert back.tobytes() == cl.tobytes() and bblob.tobytes() == blob[:blob_len].tobytes()


def test_what_does_not_fit_is_refused():
    rng = np.random.default_rng(4)
    base, blob, blob_len = random_claims(rng, 8)
    base["flags"] = 3
    base["message_cid"][:, :6] = STD

    def refused(mut):
        cl = base.copy()
        mut(cl)
        with pytest.raises(ipcfp.EngineError):
            ipcfp.compact_event_claims(cl, blob, blob_len)

    ipcfp.compact_event_claims(base, blob, blob_len)  # (the base batch itself is fine)
    refused(lambda c: c["exec_index"].__setitem__(3, 1 << 32))
    refused(lambda c: c["event_index"].__setitem__(0, (1 << 63) + 5))
    refused(lambda c: c["n_topics"].__setitem__(2, 9))
    refused(lambda c: c["data_len"].__setitem__(1, 65536))
    refused(lambda c: c["message_cid"].__setitem__((4, 1), 0x55))   # raw codec: not the standard prefix
    refused(lambda c: c["message_cid"].__setitem__((4, 39), 1))     # a 40-byte CID
    refused(lambda c: c["flags"].__setitem__(5, 4))                 # an unknown flag
    refused(lambda c: c["data_off"].__setitem__(6, blob_len))       # outside the blob
    # more than 256 (epoch, epoch, tipset) groups
    many, mblob, mlen = random_claims(rng, 300)
    many["parent_epoch"] = np.arange(300)
    with pytest.raises(ipcfp.EngineError):
        ipcfp.compact_event_claims(many, mblob, mlen)
    many["parent_epoch"] = np.arange(300) % 256
    many["child_epoch"], many["tipset"] = 7, 0
    groups, _, _, _ = ipcfp.compact_event_claims(many, mblob, mlen)
    assert len(groups) == 256


def test_packed_witness_tables_split_standard_and_other_cids():
    rng = np.random.default_rng(11)
    n = 40
    lens = rng.integers(0, 90, n).astype(np.uint32)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(lens[:-1])
    data = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    cids = np.zeros((n, 40), dtype=np.uint8)
    cids[:, :6] = STD
    cids[:, 6:38] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    cids[7, :4] = (0x01, 0x71, 0x12, 0x20)   # sha2-256: 36 bytes
    cids[7, 36:] = 0
    cids[20, 38] = 9                          # a 39-byte CID under the standard prefix: not the standard form
    pk = ipcfp.PackedWitnessTables(data, off, lens, cids)
    assert pk.esc_index.tolist() == [7, 20] and np.array_equal(pk.esc_cids, cids[[7, 20]])
    assert np.array_equal(pk.digests[[0, 39]], cids[[0, 39], 6:38]) and pk.data is not None
    assert pk.h2d_bytes == data.size + 4 * n + 32 * n + 2 * 4 + 2 * 40
    # blocks that are not back to back (gaps, another order) are laid out so
    off2 = off[::-1].copy() + np.uint64(5)
    data2 = np.zeros(data.size + 5, dtype=np.uint8)
    data2[5:] = data
    pk2 = ipcfp.PackedWitnessTables(data2, off2, lens[::-1].copy(), cids[::-1].copy())
    want = b"".join(data[int(off[i]): int(off[i]) + int(lens[i])].tobytes() for i in reversed(range(n)))
    assert pk2.data.tobytes() == want
