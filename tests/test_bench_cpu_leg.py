"""CPU: bench.py's `cpu_baseline` leg (the oracle timed on the host, 1 thread and all cores, plus the as-written
variant on a reduced tipset) runs and reports a consistent record on a small tipset — the GPU results it
cross-checks are stood in for by the sequential oracle's own."""
import numpy as np

import claims
from tools.synth import Tipset


def test_cpu_baseline_record(oracle):
    import bench

    tip = Tipset(n_receipts=3000, n_parents=3, n_planted=5, max_events=4, no_events_permille=0, variety=0)
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    ec = claims.EventClaims(tip)
    status = st.verify_event_proofs(ec, mode=1, threads=1)
    s, has, trip, _ = st.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
    st.close()
    assert (status == 1).all() and s == 1
    gm = np.zeros(len(trip), dtype=[("exec_index", np.uint64), ("event_index", np.uint64), ("emitter", np.uint64)])
    gm["exec_index"], gm["event_index"], gm["emitter"] = trip[:, 0], trip[:, 1], trip[:, 2]
    cid_status = np.ones(tip.n_blocks, dtype=np.uint8)
    rec = bench.cpu_baseline(tip, status, 1000, 2000, cid_status, (s, has, gm))
    assert rec["unit"] == "proofs/s" and rec["kind"] == "port" and rec["cores"] >= 1
    assert rec["value"] > 0 and rec["value_1_thread"] > 0 and rec["b1_as_written"]["proofs_per_s"] > 0
    for leg, nsample in ((rec["seconds"], 2000), (rec["seconds_1_thread"], 1000)):
        assert all(leg[k] >= 0 for k in ("store_build", "cid_check", "event_scan", "exec_order", "verify_sample"))
        assert leg["verify_sample_claims"] == nsample
    n = len(tip.claim_exec)
    assert abs(rec["value"] - n / rec["seconds"]["step"]) < 1e-6 * rec["value"]
    assert "3000-receipt" in rec["sample"] and "first 2000 of" in rec["sample"] and "all-cores" in rec["sample"]
    assert str(rec["cores"]) in rec["thread_sweep_step_seconds"] and "1" in rec["thread_sweep_step_seconds"]
    # a wrong GPU verdict is caught by the leg
    bad = status.copy()
    bad[5] = 0
    try:
        bench.cpu_baseline(tip, bad, 1000, 1000, cid_status, (s, has, gm))
    except SystemExit as e:
        assert "differ" in str(e)
    else:
        raise AssertionError("the cpu_baseline leg accepted a wrong verdict")
