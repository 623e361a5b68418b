"""Repro of IPCFP_FUZZ_SEED=1010, round 54 of tests/test_gpu_shard_pull_fuzz.py (a pulled shard answers ERR_MISSING_BLOCK where
the unsharded engine says TRUE)."""
import os, sys
os.environ.setdefault("IPCFP_FUZZ_SEED", "1010")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import ipc_filecoin_proofs_amd as ipcfp
from conftest import fuzz_seed
from test_gpu_fuzz import mutate
from tools.synth import Tipset

ROUND = int(os.environ.get("ROUND", "54"))
tip = Tipset(n_receipts=700, n_parents=3, dup_permille=80, n_planted=9, variety=1, max_events=5, no_events_permille=60, seed=fuzz_seed(5151))
engine = ipcfp.Engine(0)
rng = np.random.default_rng(fuzz_seed(50505))
ts, cl, blob, blob_len = ipcfp.pack_event_claims(
    tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
    tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
    tip.claim_datalen, tip.claim_data)
for it in range(ROUND + 1):
    data, touched = mutate(tip, rng, n_flips=1 + it % 3)
print("round", it, "touched", touched)
diff = np.nonzero(data != tip.data)[0]
for d in diff:
    b = int(np.searchsorted(tip.off, d, side="right") - 1)
    print("byte", int(d), "block", b, "offset in block", int(d - tip.off[b]), "len", int(tip.lens[b]), "was %02x now %02x" % (tip.data[d], data[d]))
    o = int(tip.off[b]); L = int(tip.lens[b])
    print(" block head:", bytes(tip.data[o:o + min(L, 96)]).hex())
    print(" around    :", bytes(tip.data[max(o, d - 24): d + 24]).hex(), "->", bytes(data[max(o, d - 24): d + 24]).hex())
    cid = bytes(tip.cids[b])
    print(" is parent header:", any(bytes(c) == cid for c in tip.parent_cids), " is child:", bytes(tip.child_cid) == cid)
G = 2 + it % 2
with engine.witness(data, tip.off, tip.lens, tip.cids) as w:
    want = w.verify_event_claims(ts, cl, blob, blob_len)
    plans = [w.shard_plan_tipset(tip.parent_cids, tip.child_cid, G, r) for r in range(G)]
print("want statuses:", dict(zip(*np.unique(want, return_counts=True))))
pk = ipcfp.PackedWitnessTables(data, tip.off, tip.lens, tip.cids)
ipcfp.host_register(pk.data)
allc = [bytes(c) for c in tip.cids]
for r in range(G):
    st, sw, lo, hi, nr, stats = engine.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, G, r)
    print("rank", r, "st", st, "range", lo, hi, nr, "blocks", sw.block_count if sw else None, "plan st", plans[r][0], "plan blocks", len(plans[r][4]) if plans[r][4] is not None else None)
    print(" stats", stats)
    if sw is None:
        continue
    present, _ = sw.has(allc)
    pos, c_r, b_r, bl_r = ipcfp.route_event_claims(cl, blob, blob_len, lo, hi, r == G - 1)
    got = sw.verify_event_claims(ts, c_r, b_r, bl_r)
    bad = np.nonzero(got != want[pos.astype(np.int64)])[0]
    print(" claims", len(pos), "mismatches", len(bad), "first", pos[bad[:5]], got[bad[:5]], want[pos[bad[:5]].astype(np.int64)])
    if plans[r][0] == 1:
        pids = plans[r][4]
        miss = [int(i) for i in pids if not present[i]]
        print(" planned but absent:", miss[:20], [int(tip.lens[i]) for i in miss[:20]])
        for i in miss[:3]:
            o = int(tip.off[i]); print("  absent block", i, bytes(data[o:o + min(int(tip.lens[i]), 80)]).hex())
    sw.close()
ipcfp.host_unregister(pk.data)
