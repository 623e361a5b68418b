"""CPU: the oracle's HAMT reader against a second independent writer (tests/pyhamt.py): get() returns exactly the
stored value for present keys and None for absent ones, for several bit widths and sizes (buckets only, one and
several levels of links); a wrong bit width changes the walk."""
import numpy as np
import pytest

import pyamt
import pyhamt


def val(i):
    return pyamt.array([pyamt.uint(i), pyamt.bstr(bytes([i & 0xFF]) * (i % 7))])


@pytest.mark.parametrize("bw", [3, 5, 8])
@pytest.mark.parametrize("n", [0, 1, 3, 4, 40, 2000])
def test_hamt_get_returns_what_was_written(oracle, bw, n):
    store = pyamt.Store()
    keys = [b"\x00" + pyamt.uint(1000 + i)[0:9] + bytes([i & 0xFF, i >> 8]) for i in range(n)]
    assert len(set(keys)) == n
    items = {k: val(i) for i, k in enumerate(keys)}
    root = pyhamt.build_hamt(store, items, bit_width=bw)
    st = oracle.store(*store.tables())
    absent = [b"nope", b"", b"\x00\xff\xff", keys[0] + b"\x00" if n else b"x"]
    probe = keys[:300] + absent
    status, vals = st.hamt_get(root, bw, "any", probe)
    for k, s, v in zip(probe, status, vals):
        if k in items:
            assert s == 1 and v == items[k], k
        else:
            assert s == 32, (k, s)
    if n >= 2000:
        assert len(store.blocks) > 1  # more keys than 2^bw buckets of 3 can hold: links exist
    st.close()


def test_wrong_bit_width_misses(oracle):
    store = pyamt.Store()
    items = {bytes([i, i ^ 0x5A, 7]): val(i) for i in range(200)}
    root = pyhamt.build_hamt(store, items, bit_width=5)
    st = oracle.store(*store.tables())
    keys = list(items)
    s5, _ = st.hamt_get(root, 5, "any", keys)
    s3, _ = st.hamt_get(root, 3, "any", keys)
    assert (s5 == 1).all() and not (s3 == 1).all()
    st.close()
