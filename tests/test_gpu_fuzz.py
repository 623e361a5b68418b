"""GPU: differential fuzzing of the error paths.  Random byte flips inside witness blocks (CIDs left
alone — the reference's MemoryBlockstore never re-hashes) must produce the SAME
Ok(true)/Ok(false)/Err status byte on the HIP path and on the CPU oracle for every primitive and both
verifiers: this is where the strict DAG-CBOR rules, the whole-node decode semantics of AMT/HAMT and
the first-error-wins ordering are exercised far beyond the hand-written adversarial cases."""
import numpy as np
import pytest

from conftest import fuzz_seed

import claims
from tools.synth import Tipset

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


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=260, n_parents=2, dup_permille=100, n_planted=6, variety=1, max_events=5,
                  n_actors=700, n_contracts=6, slots_per_contract=5, storage_layout_mix=1, n_actor_queries=40,
                  keep_full_state=1, seed=fuzz_seed(991))


def mutate(tip, rng, n_flips):
    data = tip.data.copy()
    touched = []
    for _ in range(n_flips):
        b = int(rng.integers(0, tip.n_blocks))
        L = int(tip.lens[b])
        if L == 0:
            continue
        pos = int(tip.off[b]) + int(rng.integers(0, L))
        mode = int(rng.integers(0, 4))
        if mode == 0:
            data[pos] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            data[pos] = int(rng.integers(0, 256))
        elif mode == 2:
            data[pos] = [0x80, 0x9F, 0xFF, 0xD8, 0x5F, 0xF6, 0x00, 0x1B][int(rng.integers(0, 8))]
        else:  # structural: bump a small array/bytes/int header
            data[pos] = (int(data[pos]) + 1) & 0xFF
        touched.append(b)
    return data, touched


def test_random_corruptions_agree(tip, engine, oracle):
    rng = np.random.default_rng(fuzz_seed(20260921))
    ec = claims.EventClaims(tip)
    sc = claims.StorageClaims(tip)
    filt = claims.make_filter(tip.topic0, tip.topic1)
    keys = [idaddr(int(i)) for i in tip.query_ids]
    idx = np.arange(0, tip.params["n_receipts"] + 3, dtype=np.uint64)
    n_err_seen = 0
    rounds = 300
    for it in range(rounds):
        data, touched = mutate(tip, rng, n_flips=1 + it % 4)
        w = engine.witness(data, tip.off, tip.lens, tip.cids)
        st = oracle.store(data, tip.off, tip.lens, tip.cids)
        ctx = f"round {it}, blocks {touched}"
        # primitives
        gs, _ = w.amt_get(tip.receipts_root, 0, "receipt", idx)
        os_, _ = st.amt_get(tip.receipts_root, 0, "receipt", idx)
        assert np.array_equal(gs, os_), ("amt_get", ctx, np.nonzero(gs != os_)[0][:5], gs[gs != os_][:5], os_[gs != os_][:5])
        gs, _ = w.hamt_get(tip.actors_root, 5, "actor_state", keys)
        os_, _ = st.hamt_get(tip.actors_root, 5, "actor_state", keys)
        assert np.array_equal(gs, os_), ("hamt_get", ctx)
        g, gc = w.exec_order(tip.parent_cids)
        o, oc = st.exec_order(tip.parent_cids)
        assert g == o and np.array_equal(gc, oc), ("exec_order", ctx, g, o)
        g, ghas, gm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                       want_touched=False)
        o, ohas, otrip, _ = st.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                           want_touched=False)
        assert g == o, ("scan status", ctx, g, o)
        if g == 1:
            assert np.array_equal(ghas, ohas) and len(gm) == len(otrip), ("scan result", ctx)
        # verifiers
        got = w.verify_event_proofs(ec.arr, ec.n, filt=filt)
        want = st.verify_event_proofs(ec, filt=filt, mode=1)
        assert np.array_equal(got, want), ("verify_event_proofs", ctx, np.nonzero(got != want)[0][:5],
                                            got[got != want][:5], want[got != want][:5])
        got = w.verify_storage_proofs(sc.arr, sc.n)
        want = st.verify_storage_proofs(sc, mode=1)
        assert np.array_equal(got, want), ("verify_storage_proofs", ctx, np.nonzero(got != want)[0][:5],
                                            got[got != want][:5], want[got != want][:5])
        n_err_seen += int((got >= 64).any()) + int(g >= 64)
        w.close()
        st.close()
    assert n_err_seen > 5  # the fuzz really reaches error paths
