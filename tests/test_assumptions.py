"""The ⚠ assumptions about the un-vendored crates, one named case each (tests/assumption_cases.py), held against the
oracle here and against the engine on the GPU.  A failing name says WHICH assumption the two disagree on — or, the day a
real-chain fixture or the crate sources are available, which one to flip."""
import numpy as np
import pytest

import assumption_cases as ac


@pytest.mark.parametrize("name", sorted(ac.CASES))
def test_oracle_assumption(oracle, name):
    store, root, version, kind, index, expect = ac.CASES[name]()
    st = oracle.store(*store.tables())
    status, _ = st.amt_get(root, version, kind, [index])
    st.close()
    assert int(status[0]) == expect, name


@pytest.mark.gpu
def test_engine_assumptions(engine):
    wrong = []
    for name in sorted(ac.CASES):
        store, root, version, kind, index, expect = ac.CASES[name]()
        data, off, lens, cids = store.tables()
        with engine.witness(data, off, lens, cids) as w:
            status, _ = w.amt_get(root, version, kind, np.array([index], dtype=np.uint64))
        if int(status[0]) != expect:
            wrong.append((name, int(status[0]), expect))
    assert not wrong, wrong
