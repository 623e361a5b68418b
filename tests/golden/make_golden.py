#!/usr/bin/env python3
"""Regenerate the committed golden fixtures:  python tests/golden/make_golden.py

The reference ships no test vectors for this path, so these fixtures pin the
repo's own behaviour: inputs come from the seeded synthetic writer (tools/synth), expected outputs
from the CPU oracle (oracle/).  Both the oracle (CPU test) and the engine (GPU test) are then held to
the committed files, so the two cannot drift together unnoticed.

  bundle_small.json       a complete UnifiedProofBundle (wire format): 4 generated event proofs +
                          6 storage proofs + adversarial variants, with its pruned witness blocks
  bundle_small.expect.json   the statuses verify_proof_bundle must return for it
  tipset_small.expect.json   for Tipset(**params): exec-order digest, scan matches, recorded-set digest,
                          per-claim statuses of the honest and the adversarial claim tables,
                          generator output digests
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bundle_ref  # noqa: E402
import claims  # noqa: E402
import oracle_lib  # noqa: E402
from tools.synth import Tipset  # noqa: E402

PARAMS = dict(n_receipts=300, n_parents=2, dup_permille=80, n_planted=4, variety=1, max_events=4,
              no_events_permille=120, n_actors=900, n_contracts=4, slots_per_contract=6, storage_layout_mix=1,
              n_actor_queries=8)


def digest(arr) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def adversarial_event_claims(tip, ec):
    """In-place mutations of a copy of the honest table: one per Ok(false)/Err class a claim can reach."""
    n = ec.n
    muts = []
    k = 0

    def nxt():
        nonlocal k
        k += 1
        return (k * 7) % n

    i = nxt(); ec.arr[i].exec_index += 1; muts.append((i, "exec_index+1"))
    i = nxt(); ec.arr[i].event_index += 7; muts.append((i, "event_index+7"))
    i = nxt(); ec.arr[i].emitter += 1; muts.append((i, "emitter+1"))
    i = nxt(); ec.arr[i].parent_epoch += 1; muts.append((i, "parent_epoch+1"))
    i = nxt(); ec.arr[i].child_epoch -= 1; muts.append((i, "child_epoch-1"))
    i = nxt(); ec.set_str(i, "message_cid", claims.cid_str(tip.receipts_root)); muts.append((i, "message_cid=other"))
    i = nxt(); ec.set_str(i, "message_cid", "zzz"); muts.append((i, "message_cid=garbage"))
    i = nxt(); ec.set_str(i, "child_block_cid", claims.cid_str(tip.parent_cids[0])); muts.append((i, "child=parent0"))
    i = nxt(); ec.set_str(i, "data", "0x00ff"); muts.append((i, "data=0x00ff"))
    i = nxt(); ec.set_str(i, "data", "nothex"); muts.append((i, "data=nothex"))
    i = nxt(); ec.set_topics(i, ["0x" + "ab" * 32]); muts.append((i, "topics=[ab..]"))
    i = nxt(); ec.set_topics(i, []); muts.append((i, "topics=[]"))
    i = nxt(); ec.set_parents(i, [claims.cid_str(c) for c in reversed(tip.parent_cids)]); muts.append((i, "parents reversed"))
    i = nxt(); ec.set_parents(i, []); muts.append((i, "parents empty"))
    i = nxt(); ec.arr[i].exec_index = 10 ** 12; muts.append((i, "exec_index=1e12"))
    return muts


def adversarial_storage_claims(tip, sc):
    n = sc.n
    muts = []
    i = 1 % n; sc.set_str(i, "value", "0x" + "ee" * 32); muts.append((i, "value=ee.."))
    i = 2 % n; sc.set_str(i, "slot", "0x" + "01" * 32); muts.append((i, "slot=01.."))
    i = 3 % n; sc.arr[i].actor_id += 10 ** 6; muts.append((i, "actor_id+1e6"))
    i = 4 % n; sc.set_str(i, "storage_root", claims.cid_str(tip.receipts_root)); muts.append((i, "storage_root=other"))
    i = 5 % n; sc.set_str(i, "actor_state_cid", "bafyjunk"); muts.append((i, "actor_state=garbage"))
    i = 6 % n; sc.set_str(i, "parent_state_root", claims.cid_str(tip.child_cid)); muts.append((i, "state_root=child"))
    i = 7 % n; sc.set_str(i, "value", "0x01"); muts.append((i, "value=0x01"))
    i = 8 % n; sc.arr[i].child_epoch += 3; muts.append((i, "child_epoch+3"))
    return muts


def main():
    orc = oracle_lib.load()
    tip = Tipset(**PARAMS)
    st = orc.store(tip.data, tip.off, tip.lens, tip.cids)
    exp = {"params": PARAMS, "n_blocks": int(tip.n_blocks), "payload_bytes": int(tip.lens.sum()),
           "arena_sha256": digest(tip.data), "cids_sha256": digest(tip.cids)}
    s, order = st.exec_order(tip.parent_cids)
    exp["exec_order"] = {"status": s, "count": len(order), "sha256": digest(order)}
    for name, actor in (("filter_actor", tip.filter_actor), ("any_actor", None)):
        s, has, trip, touched = st.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=actor)
        exp["scan_" + name] = {"status": s, "n_receipts": len(has), "has_sha256": digest(has), "matches": trip.tolist(),
                               "touched_count": len(touched), "touched_sha256": digest(touched)}
        s, trip, msg, wit = st.generate_event_proof(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=actor)
        exp["generate_" + name] = {"status": s, "proofs": trip.tolist(), "message_cids_sha256": digest(msg),
                                   "witness_count": len(wit), "witness_sha256": digest(wit)}
    ec = claims.EventClaims(tip)
    exp["event_honest"] = st.verify_event_proofs(ec, mode=0).tolist()
    filt = claims.make_filter(tip.topic0, tip.topic1)
    exp["event_honest_filtered"] = st.verify_event_proofs(ec, filt=filt, mode=0).tolist()
    muts = adversarial_event_claims(tip, ec)
    got = st.verify_event_proofs(ec, mode=0)
    exp["event_adversarial"] = [{"claim": i, "mutation": m, "status": int(got[i])} for i, m in muts]
    for tp_name, tp in (("f3_inside", claims.TrustPolicy(1, 0, tip.parent_epoch, tip.child_epoch)),
                        ("f3_child_outside", claims.TrustPolicy(1, 0, tip.parent_epoch, tip.parent_epoch)),
                        ("f3_empty_chain", claims.TrustPolicy(1, 1, 0, 10 ** 9))):
        ech = claims.EventClaims(tip, indices=np.arange(8))
        exp["event_trust_" + tp_name] = st.verify_event_proofs(ech, trust=tp, mode=0).tolist()
    sc = claims.StorageClaims(tip)
    exp["storage_honest"] = st.verify_storage_proofs(sc, mode=0).tolist()
    muts = adversarial_storage_claims(tip, sc)
    got = st.verify_storage_proofs(sc, mode=0)
    exp["storage_adversarial"] = [{"claim": i, "mutation": m, "status": int(got[i])} for i, m in muts]
    gs = []
    for i in range(min(6, len(tip.sc_actor))):
        s, o3, val, wit = st.generate_storage_proof(tip.child_cid, int(tip.sc_actor[i]), tip.sc_slot[i].tobytes())
        gs.append({"status": s, "claim_sha256": digest(np.concatenate([o3.reshape(-1), val])), "witness_count": len(wit),
                   "witness_sha256": digest(wit)})
    exp["generate_storage"] = gs
    with open(os.path.join(HERE, "tipset_small.expect.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
        f.write("\n")

    # ---- the wire-format fixture: a complete bundle ----
    s, trip, msg, wit = st.generate_event_proof(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=tip.filter_actor)
    want = {(int(e), int(v)) for e, v, _ in trip}
    sel = [i for i in range(len(tip.claim_exec)) if (int(tip.claim_exec[i]), int(tip.claim_event[i])) in want]
    ids = {tip.find_block(bytes(c[:38])) for c in wit}
    sidx = list(range(min(6, len(tip.sc_actor))))
    for i in sidx:
        _, _, _, w2 = st.generate_storage_proof(tip.child_cid, int(tip.sc_actor[i]), tip.sc_slot[i].tobytes())
        ids |= {tip.find_block(bytes(c[:38])) for c in w2}
    blocks = sorted(((tip.cids[i, :38].tobytes(), tip.block(i)) for i in ids), key=lambda cb: cb[0][6:])
    events = bundle_ref.event_dicts(tip, sel)
    storage = bundle_ref.storage_dicts(tip, sidx)
    events += [dict(events[0], exec_index=events[0]["exec_index"] + 1), dict(events[0], data="0x"),
               dict(events[0], message_cid="bafkqaaa"), dict(events[0], parent_tipset_cids=[])]
    storage += [dict(storage[0], value="0x" + "00" * 32), dict(storage[0], slot="0x12"),
                dict(storage[0], actor_id=storage[0]["actor_id"] + 12345)]
    text = bundle_ref.bundle_json(storage, events, blocks)
    with open(os.path.join(HERE, "bundle_small.json"), "w") as f:
        f.write(text)
    parsed = bundle_ref.parse_bundle(text)
    ev, sg = bundle_ref.claims_from_parsed(parsed)
    pst = orc.store(*bundle_ref.tables_from_blocks(parsed["blocks"]))
    bexp = {"n_blocks": len(blocks), "sha256": hashlib.sha256(text.encode()).hexdigest(),
            "event_status": pst.verify_event_proofs(ev, mode=0).tolist(),
            "storage_status": pst.verify_storage_proofs(sg, mode=0).tolist(),
            "honest_events": len(sel), "honest_storage": len(sidx)}
    pst.close()
    st.close()
    with open(os.path.join(HERE, "bundle_small.expect.json"), "w") as f:
        json.dump(bexp, f, indent=1, sort_keys=True)
        f.write("\n")
    print("tipset blocks", tip.n_blocks, "bundle blocks", len(blocks), "bytes", len(text))
    print("event adversarial:", [(e["mutation"], e["status"]) for e in exp["event_adversarial"]])
    print("storage adversarial:", [(e["mutation"], e["status"]) for e in exp["storage_adversarial"]])
    print("bundle:", bexp["event_status"], bexp["storage_status"])


if __name__ == "__main__":
    main()
