"""GPU: BASELINE.json's configs at their stated sizes, EVERY output compared with the oracle (run on all host
cores: oracle_lib's `threads=0` paths, tests/test_oracle_mt.py holds them to the sequential restatement).

  cfg 2  100 000 x 1 KiB blocks, digest bit flipped where i % 1024 == 7           
  cfg 3  the 1M-receipt tipset: scan has-match map, match list, recorded set, and the status byte of all
         1 000 000 claims, a tenth of them adversarial
  cfg 4  4M-actor state tree, 65 536 present ids + 1 % absent: every status and value
  cfg 5  10 000 contracts x 256 slots: all 2.57 M storage proofs, 0.1 % with a wrong value
"""
import numpy as np
import pytest

import ipc_filecoin_proofs_amd as ipcfp
from tools.synth import SEED_BASE, Tipset

pytestmark = pytest.mark.gpu


def idaddr(i: int) -> bytes:
    b = bytearray([0])
    while True:
        c = i & 0x7F
        i >>= 7
        if i:
            b.append(c | 0x80)
        else:
            b.append(c)
            return bytes(b)


def test_cfg2_100k_cid_checks_every_status(engine, oracle):
    from bench import make_cfg2

    n = 100_000
    data, off, lens = make_cfg2(n, SEED_BASE + 2)
    assert data[0] == 0x59 and data[1] == 0x03 and data[2] == 0xFD and len(data) == n * 1024
    dig = oracle.hash_batch("blake2b256", data, off, lens)
    assert np.array_equal(engine.blake2b256(data, off, lens), dig)
    cids = np.zeros((n, 40), dtype=np.uint8)
    cids[:, :6] = np.frombuffer(bytes.fromhex("0171a0e40220"), dtype=np.uint8)
    cids[:, 6:38] = dig
    flipped = np.arange(7, n, 1024)
    cids[flipped, 6 + (flipped % 32)] ^= (1 << (flipped % 8)).astype(np.uint8)
    ok, good = oracle.blake2b256_verify(data, off, lens, np.ascontiguousarray(cids[:, 6:38]), threads=0)
    with engine.witness(data, off, lens, cids) as w:
        st, nbad = w.verify_cids()
    assert np.array_equal(st, ok) and nbad == n - good == len(flipped)
    assert (st[flipped] == 0).all() and st.sum() == n - len(flipped)


@pytest.fixture(scope="module")
def big():
    return Tipset(seed=SEED_BASE + 3, n_receipts=1_000_000, n_parents=5, dup_permille=20, n_planted=1000, max_events=4,
                  no_events_permille=0, variety=0)


def test_cfg3_1m_receipts_scan_and_every_claim_vs_oracle(big, engine, oracle):
    ost = oracle.store(big.data, big.off, big.lens, big.cids, threads=0)
    n = len(big.claim_exec)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        big.parent_cids, big.child_cid, big.parent_epoch, big.child_epoch, big.claim_exec, big.claim_event,
        big.claim_emitter, big.exec_order[big.claim_exec.astype(np.int64)], big.claim_ntopics, big.claim_topics,
        big.claim_datalen, big.claim_data)
    # a tenth of the claims lie, each in its own way (index i lies iff i % 10 == 3; the way is i // 10 % 6)
    liars = np.arange(3, n, 10)
    way = (liars // 10) % 6
    cl["exec_index"][liars[way == 0]] += 1                                # → FALSE_EXEC_INDEX
    cl["event_index"][liars[way == 1]] += 40                              # → FALSE_NO_EVENT
    cl["emitter"][liars[way == 2]] ^= 1                                   # → FALSE_EMITTER
    blob[cl["data_off"][liars[way == 3]]] ^= 0x80                         # → FALSE_DATA
    blob[cl["topics_off"][liars[way == 4]] + 5] ^= 0x01                   # → FALSE_TOPIC
    cl["message_cid"][liars[way == 5], 20] ^= 0x55                        # → FALSE_MSG_NOT_IN_EXEC
    with engine.witness(big.data, big.off, big.lens, big.cids) as w:
        gs, ghas, gm, gtouched = w.scan_events(big.receipts_root, big.topic0, big.topic1, actor=big.filter_actor)
        got = w.verify_event_claims(ts, cl, blob, blob_len)
    os_, ohas, otrip, otouched = ost.scan_events(big.receipts_root, big.topic0, big.topic1, actor=big.filter_actor,
                                                 threads=0)
    assert gs == os_ == 1 and np.array_equal(ghas, ohas) and len(gm) == len(otrip) >= 1000
    assert np.array_equal(gm["exec_index"], otrip[:, 0]) and np.array_equal(gm["event_index"], otrip[:, 1])
    assert np.array_equal(gm["emitter"], otrip[:, 2])
    # recorded set: the oracle's take_seen() CIDs == the CIDs of the blocks the device marked
    assert {bytes(c) for c in big.cids[gtouched]} == {bytes(c) for c in otouched}
    want = ost.verify_event_claims_packed(ts, cl, blob, threads=0)
    ost.close()
    assert (want != 255).all()
    assert np.array_equal(got, want)
    assert (got[np.setdiff1d(np.arange(n), liars)] == 1).all() and (got[liars] != 1).all()
    assert set(np.unique(got[liars]).tolist()) == {8, 11, 12, 16, 15, 7}


@pytest.fixture(scope="module")
def state():
    return Tipset(seed=SEED_BASE + 4, n_receipts=8, n_planted=0, n_actors=4_000_000, n_contracts=10_000,
                  slots_per_contract=256, keep_full_state=0, n_actor_queries=int(65536 * 1.01))


@pytest.fixture(scope="module")
def state_oracle(state, oracle):
    st = oracle.store(state.data, state.off, state.lens, state.cids, threads=0)
    yield st
    st.close()


@pytest.fixture(scope="module")
def state_witness(state, engine):
    w = engine.witness(state.data, state.off, state.lens, state.cids)
    yield w
    w.close()


def test_cfg4_actor_gets_every_status_and_value(state, state_witness, state_oracle):
    T = state
    assert len(T.query_ids) >= 65536 and (T.query_present == 0).sum() >= 600
    keys = [idaddr(int(i)) for i in T.query_ids]
    gs, gl = state_witness.hamt_get(T.actors_root, 5, "actor_state", keys)
    os_, ov = state_oracle.hamt_get(T.actors_root, 5, "actor_state", keys)
    assert np.array_equal(gs, os_)
    present = T.query_present.astype(bool)
    assert (gs[present] == 1).all() and (gs[~present] == 32).all()
    starts = T.off[gl["block"][present]].astype(np.int64) + gl["off"][present].astype(np.int64)
    lens = gl["len"][present].astype(np.int64)
    k = 0
    for q in np.nonzero(present)[0]:
        assert T.data[starts[k]:starts[k] + lens[k]].tobytes() == ov[q]
        k += 1


def test_cfg5_every_storage_proof(state, state_witness, state_oracle):
    import torch

    T = state
    n = len(T.sc_actor)
    assert n >= 2_560_000
    cl = ipcfp.pack_storage_claims(T.child_cid, T.state_root, T.child_epoch, T.sc_actor, T.sc_actor_state,
                                   T.sc_storage_root, T.sc_slot, T.sc_value)
    wrong = np.arange(500, n, 1000)
    cl["value"][wrong, 31] ^= 1
    d_cl = torch.from_numpy(cl.view(np.uint8).reshape(-1)).cuda()
    d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    state_witness.verify_storage_claims_device(d_cl.data_ptr(), n, d_st.data_ptr())
    got = d_st.cpu().numpy()
    want = state_oracle.verify_storage_claims_packed(cl, threads=0)
    assert (want != 255).all() and np.array_equal(got, want)
    assert (got[wrong] == 21).all() and (got == 1).sum() == n - len(wrong)


def test_duplicate_cid_two_payloads_last_wins(engine, oracle):
    """load_witness_store: a second block under the same CID REPLACES the first (`HashMap::insert`,
    events/verifier.rs:79-89; K4's atomicMax).  The CID is the one of the valid payload; whichever payload comes
    last is the one every later lookup sees — on the device and in the oracle alike."""
    good = bytes.fromhex("83000083410080" "80")      # Amtv0 root [0, 0, [h'00', [], []]]: empty
    other = bytes.fromhex("830001834101808118" "2a")  # Amtv0 root [0, 1, [h'01', [], [42]]]: one value at index 0
    cid = oracle.cid_for_block(good)
    for order, expect_found in (((other, good), False), ((good, other), True)):
        data = np.frombuffer(b"".join(order), dtype=np.uint8).copy()
        lens = np.array([len(b) for b in order], dtype=np.uint32)
        off = np.array([0, len(order[0])], dtype=np.uint64)
        cids = np.zeros((2, 40), dtype=np.uint8)
        cids[:, :38] = np.frombuffer(cid, dtype=np.uint8)
        ost = oracle.store(data, off, lens, cids)
        o_st, o_val = ost.amt_get(cid, 0, "any", [0])
        ost.close()
        with engine.witness(data, off, lens, cids) as w:
            g_st, g_loc = w.amt_get(cid, 0, "any", [0])
            cst, nbad = w.verify_cids()
        assert g_st.tolist() == o_st.tolist() == ([1] if expect_found else [32])
        if expect_found:
            assert int(g_loc["block"][0]) == 1 and o_val[0] == b"\x18\x2a"
        # K1 reports per block, not per CID: exactly the block whose bytes hash to the CID is OK
        assert cst.tolist() == [1 if b is good else 0 for b in order] and nbad == 1
