"""CPU: the oracle's AMT reader against a SECOND independent writer (tests/pyamt.py, pure Python; the first is
tools/synth's C++ writer): get() returns exactly the bytes that were stored, None for absent indices, for v0 and
v3 roots, several bit widths, dense and sparse index sets, and the scan's receipt map agrees."""
import numpy as np
import pytest

import pyamt


def value(i):
    return pyamt.receipt(gas=1000 + i)


@pytest.mark.parametrize("version,bw", [(0, 3), (3, 1), (3, 3), (3, 5), (3, 8)])
@pytest.mark.parametrize("shape", ["dense", "sparse"])
def test_amt_get_returns_what_was_written(oracle, version, bw, shape):
    idx = list(range(0, 70)) if shape == "dense" else [0, 1, 7, 8, 63, 64, 65, 511, 512, 4095, 4096, 70001]
    store = pyamt.Store()
    items = {i: value(i) for i in idx}
    root = pyamt.build_amt(store, items, version=version, bit_width=bw)
    st = oracle.store(*store.tables())
    probe = sorted(set(idx) | {2, 9, 66, 513, 70000, 70002, 10 ** 9}) if shape == "sparse" else list(range(0, 75))
    status, vals = st.amt_get(root, version, "receipt", probe)
    for i, s, v in zip(probe, status, vals):
        if i in items:
            assert s == 1 and v == items[i], (i, s)
        else:
            assert s == 32, (i, s)  # NOT_FOUND
    st.close()


def test_scan_sees_every_written_receipt(oracle):
    store = pyamt.Store()
    idx = [0, 3, 64, 65, 1000]
    root = pyamt.build_amt(store, {i: value(i) for i in idx})
    st = oracle.store(*store.tables())
    s, has, trip, touched = st.scan_events(root, bytes(32), bytes(32))
    assert s == 1 and len(has) == 1001 and not has.any() and len(trip) == 0
    st.close()


def test_wrong_version_or_width_is_an_error(oracle):
    store = pyamt.Store()
    root0 = pyamt.build_amt(store, {i: value(i) for i in range(20)}, version=0)
    root3 = pyamt.build_amt(store, {i: value(i) for i in range(20)}, version=3, bit_width=5)
    st = oracle.store(*store.tables())
    assert st.amt_get(root0, 3, "receipt", [0])[0][0] >= 64   # a v0 root read as v3: arity error
    assert st.amt_get(root3, 0, "receipt", [0])[0][0] >= 64
    assert st.amt_get(root3, 3, "receipt", [5])[0][0] == 1
    st.close()
