"""GPU: ipcfp_hamt_get on HAMTs written by the pure-Python writer (tests/pyhamt.py): present keys locate exactly
the stored value, absent keys are NOT_FOUND, bit widths 3, 5 and 8, bucket-only and multi-level trees — and every
status equals the oracle's."""
import numpy as np
import pytest

import pyamt
import pyhamt

pytestmark = pytest.mark.gpu


def val(i):
    return pyamt.array([pyamt.uint(i), pyamt.bstr(bytes([i & 0xFF]) * (i % 7))])


@pytest.mark.parametrize("bw", [3, 5, 8])
@pytest.mark.parametrize("n", [0, 1, 4, 40, 2000])
def test_hamt_get_locates_what_was_written(engine, oracle, bw, n):
    store = pyamt.Store()
    keys = [b"\x00" + pyamt.uint(1000 + i)[0:9] + bytes([i & 0xFF, i >> 8]) for i in range(n)]
    items = {k: val(i) for i, k in enumerate(keys)}
    root = pyhamt.build_hamt(store, items, bit_width=bw)
    data, off, lens, cids = store.tables()
    w = engine.witness(data, off, lens, cids)
    st = oracle.store(data, off, lens, cids)
    probe = keys[:300] + [b"nope", b"", b"\x00\xff\xff", keys[0] + b"\x00" if n else b"x"]
    gs, gl = w.hamt_get(root, bw, "any", probe)
    os_, ov = st.hamt_get(root, bw, "any", probe)
    assert np.array_equal(gs, os_)
    for k, s, l in zip(probe, gs, gl):
        if k in items:
            o = int(off[l["block"]]) + int(l["off"])
            assert s == 1 and data[o:o + int(l["len"])].tobytes() == items[k], k
        else:
            assert s == 32, k
    w.close()
    st.close()
