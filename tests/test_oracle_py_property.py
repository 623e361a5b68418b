"""CPU, property-based (hypothesis): random AMTs and HAMTs written by the pure-Python writers are read back by
the oracle exactly — every stored index/key returns its value, every other one None."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle_lib
import pyamt
import pyhamt

ORC = oracle_lib.load()


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(version=st.sampled_from([0, 3]), bw=st.integers(1, 8),
       idx=st.sets(st.one_of(st.integers(0, 300), st.integers(0, 1 << 20), st.integers(0, (1 << 40))), max_size=40),
       probes=st.lists(st.integers(0, (1 << 41)), max_size=10))
def test_random_amt(version, bw, idx, probes):
    store = pyamt.Store()
    items = {i: pyamt.array([pyamt.uint(i), pyamt.bstr(b"v" * (i % 5))]) for i in idx}
    root = pyamt.build_amt(store, items, version=version, bit_width=bw)
    s = ORC.store(*store.tables())
    probe = sorted(set(idx) | set(probes))
    if probe:
        status, vals = s.amt_get(root, version, "any", probe)
        for i, stt, v in zip(probe, status, vals):
            if i in items:
                assert stt == 1 and v == items[i], i
            else:
                assert stt == 32, (i, stt)
    s.close()


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(bw=st.sampled_from([1, 2, 3, 4, 5, 6, 8]), keys=st.sets(st.binary(min_size=0, max_size=12), max_size=120),
       probes=st.lists(st.binary(min_size=0, max_size=12), max_size=8))
def test_random_hamt(bw, keys, probes):
    store = pyamt.Store()
    items = {k: pyamt.array([pyamt.bstr(k), pyamt.uint(len(k))]) for k in keys}
    root = pyhamt.build_hamt(store, items, bit_width=bw)
    s = ORC.store(*store.tables())
    probe = list(keys) + [p for p in probes if p not in keys]
    if probe:
        status, vals = s.hamt_get(root, bw, "any", probe)
        for k, stt, v in zip(probe, status, vals):
            if k in items:
                assert stt == 1 and v == items[k], k
            else:
                assert stt == 32, (k, stt)
    s.close()
