"""GPU: AMT enumeration on shapes that leave the dense fast path (amt_enum.hip): sparse indices, a lying
`count`, a non-minimal height, an empty tree — the scan must fall back to the general level-synchronous
walk and still agree with the oracle bit for bit."""
import numpy as np
import pytest

import pyamt

pytestmark = pytest.mark.gpu

T0 = bytes(range(32))
T1 = bytes(range(32, 64))


def run_scan(engine, oracle, store, root):
    tabs = store.tables()
    w = engine.witness(*tabs)
    st = oracle.store(*tabs)
    gs, ghas, gm, gids = w.scan_events(root, T0, T1)
    os_, ohas, otrip, otouched = st.scan_events(root, T0, T1)
    w.close()
    st.close()
    assert gs == os_, (gs, os_)
    if gs == 1:
        assert np.array_equal(ghas, ohas) and len(gm) == len(otrip)
    return gs, ghas


CASES = {
    "dense 100": dict(idx=range(100)),
    "dense 8": dict(idx=range(8)),
    "dense 9": dict(idx=range(9)),
    "dense 512": dict(idx=range(512)),
    "dense 513": dict(idx=range(513)),
    "single 0": dict(idx=[0]),
    "single far": dict(idx=[70000]),
    "sparse mix": dict(idx=[0, 5, 64, 65, 4095, 100000]),
    "hole at 3": dict(idx=[i for i in range(40) if i != 3]),
    "missing tail node": dict(idx=list(range(16)) + list(range(24, 30))),
    "count too big": dict(idx=range(50), count=60),
    "count too small": dict(idx=range(50), count=40),
    "count zero but values": dict(idx=range(5), count=0),
    "tall root": dict(idx=range(20), height=3),
    # dense, but one interior link is a 36-byte sha2-256 CID: the 43-byte link stride of the fast path is wrong
    "odd link first child": dict(idx=range(600), odd_links={(1, 0)}),
    "odd link middle": dict(idx=range(600), odd_links={(0, 264), (1, 128)}),
    "odd link last": dict(idx=range(600), odd_links={(0, 592)}),
    "empty": dict(idx=[]),
    "empty tall": dict(idx=[], height=2),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_scan_over_receipt_amt_shapes(engine, oracle, name):
    c = CASES[name]
    store = pyamt.Store()
    items = {i: pyamt.receipt(gas=1000 + i) for i in c["idx"]}
    root = pyamt.build_amt(store, items, version=0, height=c.get("height"), count=c.get("count"),
                           odd_links=c.get("odd_links", ()))
    gs, has = run_scan(engine, oracle, store, root)
    assert gs == 1
    idx = list(c["idx"])
    assert len(has) == (max(idx) + 1 if idx else 0)
    assert not has.any()


def test_scan_missing_interior_block(engine, oracle):
    store = pyamt.Store()
    root = pyamt.build_amt(store, {i: pyamt.receipt(gas=i) for i in range(600)})
    victim = [c for c, b in store.blocks.items() if c != root][3]
    del store.blocks[victim]
    gs, _ = run_scan(engine, oracle, store, root)
    assert gs == 65  # ERR_MISSING_BLOCK on both sides


def test_scan_value_of_wrong_type(engine, oracle):
    store = pyamt.Store()
    items = {i: pyamt.receipt(gas=i) for i in range(30)}
    items[17] = pyamt.array([pyamt.uint(0), pyamt.bstr(b""), pyamt.uint(1)])  # 3-tuple: not a Receipt
    root = pyamt.build_amt(store, items)
    gs, _ = run_scan(engine, oracle, store, root)
    assert gs == 66  # ERR_DECODE
