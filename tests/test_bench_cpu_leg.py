"""CPU: bench.py's `cpu_baseline` leg (the oracle timed on the host) runs and reports a consistent record on a
small tipset — the GPU verdicts it cross-checks are stood in for by the oracle's own."""
import numpy as np

import claims
from tools.synth import Tipset


def test_cpu_baseline_record(oracle):
    import bench

    tip = Tipset(n_receipts=3000, n_parents=3, n_planted=5, max_events=4, no_events_permille=0, variety=0)
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    ec = claims.EventClaims(tip)
    status = st.verify_event_proofs(ec, mode=1, threads=1)
    st.close()
    assert (status == 1).all()
    rec = bench.cpu_baseline(tip, status, sample=1000)
    assert rec["unit"] == "proofs/s" and rec["cores"] == 1 and rec["kind"] == "port"
    assert rec["value"] > 0 and rec["value_verify_all_host_threads"] > 0
    secs = rec["seconds"]
    assert all(secs[k] >= 0 for k in ("store_build", "cid_check", "event_scan", "exec_order", "verify_sample"))
    n = len(tip.claim_exec)
    assert abs(rec["value"] - n / secs["step_1_thread"]) < 1e-6 * rec["value"]
    assert "3000-receipt" in rec["sample"] and "first 1000 of" in rec["sample"]
