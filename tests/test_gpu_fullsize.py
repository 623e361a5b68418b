"""GPU: BASELINE.json's full-size tipset (1,000,000 receipts, ≈1.29 M blocks, 0.44 GB) checked through
size-independent properties — the oracle is too slow to re-verify everything here, the generator's own
records (writer-side truth) and algebraic properties are not."""
import numpy as np
import pytest

from tools.synth import SEED_BASE, Tipset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    return Tipset(seed=SEED_BASE + 3, n_receipts=1_000_000, n_parents=5, dup_permille=20, n_planted=1000, max_events=4,
                  no_events_permille=0, variety=0)


@pytest.fixture(scope="module")
def wbig(big, engine):
    w = engine.witness(big.data, big.off, big.lens, big.cids)
    yield w
    w.close()


def test_every_cid_verifies_and_exactly_the_tampered_ones_fail(big, wbig, engine):
    st, nbad = wbig.verify_cids()
    assert nbad == 0 and (st == 1).all()
    st2, nbad2 = wbig.verify_cids()
    assert nbad2 == 0 and np.array_equal(st, st2)  # idempotent
    rng = np.random.default_rng(1)
    victims = np.unique(rng.integers(0, big.n_blocks, 64))
    data = big.data.copy()
    for b in victims:
        data[int(big.off[b]) + int(rng.integers(0, big.lens[b]))] ^= 0x20
    w2 = engine.witness(data, big.off, big.lens, big.cids)
    st, nbad = w2.verify_cids()
    want = np.ones(big.n_blocks, dtype=np.uint8)
    want[victims] = 0
    assert nbad == len(victims) and np.array_equal(st, want)
    w2.close()


def test_exec_order_equals_the_writers_record(big, wbig):
    s, order = wbig.exec_order(big.parent_cids)
    assert s == 1 and np.array_equal(order, big.exec_order)


def test_scan_finds_exactly_the_emitter_filtered_matches(big, wbig):
    s, has, m, _ = wbig.scan_events(big.receipts_root, big.topic0, big.topic1, actor=big.filter_actor,
                                    want_touched=False)
    assert s == 1 and len(has) == 1_000_000
    assert set(big.planted.tolist()) <= set(m["exec_index"].tolist())
    assert has.sum() == len(set(m["exec_index"].tolist()))
    assert np.array_equal(np.nonzero(has)[0], np.unique(m["exec_index"]))
    # match list is in (exec_index, event_index) order and every match is a StampedEvent from the filter actor
    key = m["exec_index"].astype(np.int64) * 64 + m["event_index"].astype(np.int64)
    assert (np.diff(key) > 0).all()
    assert (m["emitter"] == big.filter_actor).all()
    # without the emitter filter the match set can only grow
    s2, has2, m2, _ = wbig.scan_events(big.receipts_root, big.topic0, big.topic1, actor=None, want_touched=False)
    assert s2 == 1 and (has2 >= has).all() and len(m2) >= len(m)


def test_all_claims_verify_and_shifted_claims_fail(big, wbig):
    import torch

    import ipc_filecoin_proofs_amd as ipcfp

    n = len(big.claim_exec)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        big.parent_cids, big.child_cid, big.parent_epoch, big.child_epoch, big.claim_exec, big.claim_event,
        big.claim_emitter, big.exec_order[big.claim_exec.astype(np.int64)], big.claim_ntopics, big.claim_topics,
        big.claim_datalen, big.claim_data)
    d_blob = torch.from_numpy(blob).cuda()
    d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")

    def run(c):
        d_cl = torch.from_numpy(c.view(np.uint8).reshape(-1)).cuda()
        torch.cuda.synchronize()
        wbig.verify_event_claims_device(ts, d_cl.data_ptr(), n, d_blob.data_ptr(), blob_len, d_st.data_ptr())
        return d_st.cpu().numpy()

    assert (run(cl) == 1).all()
    shifted = cl.copy()
    shifted["exec_index"] += 1
    assert (run(shifted) == 8).all()            # FALSE_EXEC_INDEX everywhere
    wrong_event = cl.copy()
    wrong_event["event_index"] += 40
    assert (run(wrong_event) == 11).all()       # FALSE_NO_EVENT everywhere
    wrong_emitter = cl.copy()
    wrong_emitter["emitter"] ^= 1
    assert (run(wrong_emitter) == 12).all()     # FALSE_EMITTER everywhere


def test_amt_get_sample_is_consistent_with_the_enumeration(big, wbig):
    rng = np.random.default_rng(3)
    idx = np.unique(rng.integers(0, 1_000_000, 50_000)).astype(np.uint64)
    st, loc = wbig.amt_get(big.receipts_root, 0, "receipt", idx)
    assert (st == 1).all()
    # every receipt value is a 4-tuple and distinct indices give distinct locations
    starts = big.off[loc["block"]].astype(np.int64) + loc["off"].astype(np.int64)
    assert (big.data[starts] == 0x84).all()
    assert len(np.unique(starts)) == len(idx)
    st, _ = wbig.amt_get(big.receipts_root, 0, "receipt", np.array([1_000_000, 8 ** 7 - 1, 8 ** 7], dtype=np.uint64))
    assert st.tolist() == [32, 32, 32]
