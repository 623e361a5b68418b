"""CPU: pin the C++ oracle's three hashes against hashlib, an independent pure-Python
Keccak, and the published known answers of SURVEY.md Appendix B."""
import hashlib
import os
import random

import numpy as np
import pytest

import pymini

LENS = [0, 1, 2, 3, 31, 32, 33, 54, 55, 56, 57, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 135, 136, 137,
        255, 256, 257, 271, 272, 273, 1023, 1024, 1025, 4096, 10000]


def test_pymini_keccak_permutation_matches_hashlib_sha3():
    rng = random.Random(1)
    for n in [0, 1, 55, 135, 136, 137, 300, 1000]:
        b = bytes(rng.getrandbits(8) for _ in range(n))
        assert pymini.sha3_256_via_mini(b) == hashlib.sha3_256(b).digest()


KATS = [
    ("blake2b256", b"", "0e5751c026e543b2e8ab2eb06099daa1d1e5df47778f7787faab45cdf12fe3a8"),
    ("blake2b256", b"abc", "bddd813c634239723171ef3fee98579b94964e3bb1cb3e427262c8c068d52319"),
    ("keccak256", b"", "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"),
    ("keccak256", b"abc", "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"),
    ("keccak256", b"Transfer(address,address,uint256)",
     "ddf252ad1be2c89b69c2b068fc378daa952ba7f163c4a11628f55a4df523b3ef"),
    # the demo's own event signature, /root/reference/src/main.rs:61
    ("keccak256", b"NewTopDownMessage(bytes32,uint256)",
     "43a056172fcece714f70dd1570a9c9152899b8f299e7924d5e817217ee7783bf"),
    ("sha256", b"abc", "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"),
    ("sha256", b"", "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"),
]


@pytest.mark.parametrize("kind,msg,hexd", KATS)
def test_oracle_known_answers(oracle, kind, msg, hexd):
    assert getattr(oracle, kind)(msg).hex() == hexd
    assert getattr(pymini, kind)(msg).hex() == hexd


@pytest.mark.parametrize("kind", ["blake2b256", "keccak256", "sha256"])
def test_oracle_matches_python_all_lengths(oracle, kind):
    rng = random.Random(7)
    for n in LENS:
        b = bytes(rng.getrandbits(8) for _ in range(n))
        assert getattr(oracle, kind)(b) == getattr(pymini, kind)(b), (kind, n)


def test_oracle_batch_and_verify(oracle):
    rng = np.random.default_rng(3)
    lens = np.array([0, 1, 127, 128, 129, 1024, 77, 256], dtype=np.uint32)
    off = np.zeros(len(lens), dtype=np.uint64)
    off[1:] = np.cumsum(lens[:-1])
    data = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    for kind in ("blake2b256", "keccak256", "sha256"):
        out = oracle.hash_batch(kind, data, off, lens)
        for i in range(len(lens)):
            m = data[int(off[i]): int(off[i]) + int(lens[i])].tobytes()
            assert out[i].tobytes() == getattr(pymini, kind)(m)
    exp = oracle.hash_batch("blake2b256", data, off, lens)
    exp[3, 5] ^= 0x10
    ok, good = oracle.blake2b256_verify(data, off, lens, exp)
    assert ok.tolist() == [1, 1, 1, 0, 1, 1, 1, 1] and good == 7
