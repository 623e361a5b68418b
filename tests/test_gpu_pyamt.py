"""GPU: ipcfp_amt_get on AMTs written by the pure-Python writer (tests/pyamt.py): the located bytes are exactly
what was stored, absent indices are NOT_FOUND, for v0 and v3 roots at bit widths 1, 3, 5 and 8, dense and sparse —
and every status equals the oracle's."""
import numpy as np
import pytest

import pyamt

pytestmark = pytest.mark.gpu


def value(i):
    return pyamt.receipt(gas=1000 + i)


@pytest.mark.parametrize("version,bw", [(0, 3), (3, 1), (3, 3), (3, 5), (3, 8)])
@pytest.mark.parametrize("shape", ["dense", "sparse"])
def test_amt_get_locates_what_was_written(engine, oracle, version, bw, shape):
    idx = list(range(0, 70)) if shape == "dense" else [0, 1, 7, 8, 63, 64, 65, 511, 512, 4095, 4096, 70001]
    store = pyamt.Store()
    items = {i: value(i) for i in idx}
    root = pyamt.build_amt(store, items, version=version, bit_width=bw)
    data, off, lens, cids = store.tables()
    w = engine.witness(data, off, lens, cids)
    st = oracle.store(data, off, lens, cids)
    probe = sorted(set(idx) | {2, 9, 66, 513, 70000, 70002, 10 ** 9}) if shape == "sparse" else list(range(0, 75))
    gs, gl = w.amt_get(root, version, "receipt", probe)
    os_, ov = st.amt_get(root, version, "receipt", probe)
    assert np.array_equal(gs, os_)
    for i, s, l in zip(probe, gs, gl):
        if i in items:
            o = int(off[l["block"]]) + int(l["off"])
            assert s == 1 and data[o:o + int(l["len"])].tobytes() == items[i], i
        else:
            assert s == 32, i
    # wrong version: same error on both sides
    other = 3 if version == 0 else 0
    gs2, _ = w.amt_get(root, other, "receipt", probe[:3])
    os2, _ = st.amt_get(root, other, "receipt", probe[:3])
    assert np.array_equal(gs2, os2)
    w.close()
    st.close()
