"""CPU: the host-side claim lowering (csrc/host/pack_claims.cpp, the first half of ipcfp_verify_event_proofs —
no GPU involved): strings → ipcfp_tipset_ref_t / ipcfp_event_claim_t / blob.  Checked against the numpy packer
of the binding (written independently, from binary inputs), single- and multi-threaded, and on claims whose
strings do not parse."""
import numpy as np

import claims
import ipc_filecoin_proofs_amd as ipcfp
from tools.synth import Tipset


def numpy_pack(tip, idx):
    return ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec[idx], tip.claim_event[idx],
        tip.claim_emitter[idx], tip.exec_order[tip.claim_exec[idx].astype(np.int64)], tip.claim_ntopics[idx],
        tip.claim_topics[idx], tip.claim_datalen[idx], tip.claim_data[idx])


def test_small_batch_equals_numpy_packer():
    tip = Tipset(n_receipts=500, n_parents=3, n_planted=4, variety=1, max_events=4)
    idx = np.arange(len(tip.claim_exec))
    ec = claims.EventClaims(tip)
    ts, cl, blob = ipcfp.pack_event_proofs(ec.arr, ec.n)
    nts, ncl, nblob, total = numpy_pack(tip, idx)
    assert ts.tobytes() == nts.tobytes()
    assert cl.tobytes() == ncl.tobytes()
    assert blob.tobytes() == nblob[:total].tobytes()


def test_threaded_batch_equals_numpy_packer():
    """≥ 8192 claims: the lowering runs on several threads, each range into its own blob, offsets rebased."""
    tip = Tipset(n_receipts=40000, n_parents=2, n_planted=10, variety=1, max_events=3, no_events_permille=0)
    idx = np.arange(len(tip.claim_exec))
    assert len(idx) >= 3 * 8192
    ec = claims.EventClaims(tip)
    ts, cl, blob = ipcfp.pack_event_proofs(ec.arr, ec.n)
    nts, ncl, nblob, total = numpy_pack(tip, idx)
    assert ts.tobytes() == nts.tobytes()
    assert np.array_equal(cl, ncl)
    assert blob.tobytes() == nblob[:total].tobytes()
    # every claim's slices land where its offsets say
    for i in (0, 8191, 8192, 20000, len(idx) - 1):
        o = int(cl["topics_off"][i])
        nt = int(cl["n_topics"][i])
        for t in range(nt):
            assert blob[o + 33 * t] == 1 and blob[o + 33 * t + 1: o + 33 * t + 33].tobytes() == tip.claim_topics[i, t].tobytes()
        d = int(cl["data_off"][i])
        assert blob[d: d + int(cl["data_len"][i])].tobytes() == tip.claim_data[i, : int(tip.claim_datalen[i])].tobytes()


def test_unparsable_strings_and_several_tipsets():
    tip = Tipset(n_receipts=200, n_parents=2, n_planted=2, variety=1)
    ec = claims.EventClaims(tip, indices=np.arange(12))
    ec.set_str(1, "message_cid", "not a cid")
    ec.set_str(2, "data", "0xabc")            # odd number of digits
    ec.set_str(3, "data", "abcd")             # no 0x
    ec.set_str(4, "data", "0XABCD")           # upper-case prefix and digits compare equal ignoring case
    ec.set_topics(5, ["0x" + "ab" * 31, "0x" + "cd" * 32, "zz"])
    ec.set_parents(6, [claims.cid_str(tip.parent_cids[1])])                 # another tipset key
    ec.set_parents(7, [claims.cid_str(tip.parent_cids[1])])                 # the same one again (other array object)
    ec.set_parents(8, ["garbage", claims.cid_str(tip.parent_cids[0])])
    ec.set_str(9, "child_block_cid", "also garbage")
    ec.set_parents(10, [])
    ts, cl, blob = ipcfp.pack_event_proofs(ec.arr, ec.n)
    assert cl["tipset"].tolist() == [0, 0, 0, 0, 0, 0, 1, 1, 2, 3, 4, 0]
    assert ts["flags"].tolist() == [3, 3, 2, 1, 3] and ts["n_parents"].tolist() == [2, 1, 2, 2, 0]
    assert ts["parents"][1, 0, :38].tobytes() == tip.parent_cids[1]
    f = cl["flags"]
    assert f[0] == 3 and f[1] == 2 and f[2] == 1 and f[3] == 1 and f[4] == 3
    assert bytes(cl["message_cid"][1]) == b"\xff" * 40  # never a possible witness key
    assert cl["data_len"][4] == 2 and blob[cl["data_off"][4]: cl["data_off"][4] + 2].tobytes() == b"\xab\xcd"
    o = int(cl["topics_off"][5])
    assert cl["n_topics"][5] == 3 and [int(blob[o + 33 * t]) for t in range(3)] == [0, 1, 0]
    assert blob[o + 34: o + 66].tobytes() == b"\xcd" * 32


def storage_tip():
    return Tipset(n_receipts=50, n_planted=1, n_actors=30000, n_contracts=100, slots_per_contract=200,
                  storage_layout_mix=1, n_actor_queries=4, keep_full_state=0)


def test_storage_claims_equal_numpy_packer():
    """Strings → ipcfp_storage_claim_t, ≥ 8192 claims (several threads), byte for byte the numpy packer's output."""
    tip = storage_tip()
    n = len(tip.sc_actor)
    assert n >= 2 * 8192
    sc = claims.StorageClaims(tip)
    got = ipcfp.pack_storage_proofs(sc.arr, sc.n)
    want = ipcfp.pack_storage_claims(tip.child_cid, tip.state_root, tip.child_epoch, tip.sc_actor, tip.sc_actor_state,
                                     tip.sc_storage_root, tip.sc_slot, tip.sc_value)
    assert got.tobytes() == want.tobytes()


def test_storage_claims_with_unparsable_strings():
    tip = Tipset(n_receipts=50, n_planted=1, n_actors=500, n_contracts=3, slots_per_contract=6, storage_layout_mix=1)
    sc = claims.StorageClaims(tip, indices=np.arange(10))
    good = claims.cid_str(tip.state_root)
    sc.set_str(1, "child_block_cid", "nope")
    sc.set_str(2, "parent_state_root", good.upper())          # parses, but is not the canonical spelling
    sc.set_str(3, "actor_state_cid", "f" + tip.sc_actor_state[3, :38].tobytes().hex())  # base16 multibase: same
    sc.set_str(4, "storage_root", "")
    sc.set_str(5, "slot", "0x0x" + "11" * 32)                 # trim_start_matches strips every leading "0x"
    sc.set_str(6, "slot", "0x" + "11" * 31)
    sc.set_str(7, "value", "0X" + "AB" * 32)
    sc.set_str(8, "value", "0x" + "ab" * 31)
    sc.set_str(9, "value", "ab" * 33)
    f = ipcfp.pack_storage_proofs(sc.arr, sc.n)["flags"]
    assert f[0] == 63
    assert f[1] == 63 & ~1 and f[2] == 63 & ~2 and f[3] == 63 & ~4 and f[4] == 63 & ~8
    assert f[5] == 63 and f[6] == 63 & ~16
    assert f[7] == 63 and f[8] == 63 & ~32 and f[9] == 63 & ~32
    out = ipcfp.pack_storage_proofs(sc.arr, sc.n)
    assert out["slot"][5].tobytes() == b"\x11" * 32 and out["value"][7].tobytes() == b"\xab" * 32
