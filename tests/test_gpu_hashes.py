"""GPU: K1/K2/K3 hash kernels vs the CPU oracle, bit-exact, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE_LENS = [0, 1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 55, 56, 57, 63, 64, 65, 111, 112, 119, 120, 121, 127, 128,
             129, 135, 136, 137, 255, 256, 257, 271, 272, 273, 383, 384, 385, 1023, 1024, 1025, 2047, 2048, 4097]


def make_table(lens, seed, pack_tight=True):
    rng = np.random.default_rng(seed)
    lens = np.asarray(lens, dtype=np.uint32)
    if pack_tight:
        off = np.zeros(len(lens), dtype=np.uint64)
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        total = int(lens.sum())
    else:  # ragged gaps
        gaps = rng.integers(0, 5, len(lens))
        off = np.zeros(len(lens), dtype=np.uint64)
        pos = 3
        for i, L in enumerate(lens):
            off[i] = pos
            pos += int(L) + int(gaps[i])
        total = pos
    data = rng.integers(0, 256, max(total, 1), dtype=np.uint8)[:total]
    return data, off, lens


@pytest.mark.parametrize("kind", ["blake2b256", "keccak256", "sha256"])
@pytest.mark.parametrize("tight", [True, False])
def test_hash_edge_lengths(engine, oracle, kind, tight):
    data, off, lens = make_table(EDGE_LENS * 3, seed=11, pack_tight=tight)
    got = getattr(engine, kind)(data, off, lens)
    want = oracle.hash_batch(kind, data, off, lens)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ["blake2b256", "keccak256", "sha256"])
def test_hash_random_batch(engine, oracle, kind):
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 700, 20000).astype(np.uint32)
    data, off, lens = make_table(lens, seed=6)
    got = getattr(engine, kind)(data, off, lens)
    want = oracle.hash_batch(kind, data, off, lens)
    assert np.array_equal(got, want)


def test_empty_batch(engine):
    z = np.zeros(0, dtype=np.uint8)
    assert engine.blake2b256(z, np.zeros(0, np.uint64), np.zeros(0, np.uint32)).shape == (0, 32)


def cids_for(digests, flip=()):
    n = len(digests)
    c = np.zeros((n, 40), dtype=np.uint8)
    c[:, :6] = np.frombuffer(bytes.fromhex("0171a0e40220"), dtype=np.uint8)
    c[:, 6:38] = digests
    for i in flip:
        c[i, 6 + (i % 32)] ^= 1 << (i % 8)
    return c


def test_cid_check_config2_shape(engine, oracle):
    """BASELINE config 2 in miniature: 1 KiB blocks `59 03 FD ‖ 1021 bytes`, every
    block i with i % 1024 == 7 carries a CID with one digest bit flipped."""
    n = 5000
    rng = np.random.default_rng(2)
    data = rng.integers(0, 256, n * 1024, dtype=np.uint8)
    data = data.reshape(n, 1024)
    data[:, 0:3] = [0x59, 0x03, 0xFD]
    data = data.reshape(-1)
    off = (np.arange(n, dtype=np.uint64) * 1024)
    lens = np.full(n, 1024, dtype=np.uint32)
    dig = oracle.hash_batch("blake2b256", data, off, lens)
    bad = [i for i in range(n) if i % 1024 == 7]
    cids = cids_for(dig, flip=bad)
    with engine.witness(data, off, lens, cids) as w:
        st, nbad = w.verify_cids()
    want = np.ones(n, dtype=np.uint8)
    want[bad] = 0
    assert np.array_equal(st, want) and nbad == len(bad)
    ok, good = oracle.blake2b256_verify(data, off, lens, cids[:, 6:38].copy())
    assert np.array_equal(ok, st) and good == n - len(bad)


def test_cid_check_ragged_and_unchecked(engine, oracle):
    """Variable lengths (incl. empty block), unaligned offsets, a duplicate CID, and a
    non-blake2b CID that must be reported UNCHECKED rather than hashed."""
    lens = [0, 1, 127, 128, 129, 300, 5000, 64, 64, 77, 1024, 2049]
    data, off, lens = make_table(lens, seed=9, pack_tight=False)
    dig = oracle.hash_batch("blake2b256", data, off, lens)
    cids = cids_for(dig, flip=[5])
    # block 9: sha2-256 CID (01 71 12 20 …) → not checked by K1
    cids[9] = 0
    cids[9, :4] = [0x01, 0x71, 0x12, 0x20]
    cids[9, 4:36] = oracle.hash_batch("sha256", data, off, lens)[9]
    with engine.witness(data, off, lens, cids) as w:
        st, nbad = w.verify_cids()
        st2, nbad2 = w.verify_cids()  # idempotent
    want = np.ones(len(lens), dtype=np.uint8)
    want[5] = 0
    want[9] = 2
    assert st.tolist() == want.tolist() and nbad == 1
    assert np.array_equal(st, st2) and nbad2 == 1
