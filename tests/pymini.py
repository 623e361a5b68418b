"""Second, INDEPENDENT mini-oracle in pure Python (hashlib + a 30-line Keccak).

Purpose: pin the C++ oracle.  Blake2b-256 and SHA-256 come from hashlib; Keccak-256
is written here from the Keccak reference pseudo-code and is itself validated against
hashlib.sha3_256 (same permutation, different pad byte) in test_oracle_hashes.py.
"""
import hashlib

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_M = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def _f1600(a):
    # a[x][y], reference formulation
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        x, y = 1, 0
        b[0][0] = a[0][0]
        for t in range(24):
            r = ((t + 1) * (t + 2) // 2) % 64
            b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], r)
            x, y = y, (2 * x + 3 * y) % 5
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    return a


def _sponge(data: bytes, pad: int) -> bytes:
    rate = 136
    p = bytearray(data)
    p.append(pad)
    while len(p) % rate:
        p.append(0)
    p[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for o in range(0, len(p), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(p[o + 8 * i : o + 8 * i + 8], "little")
        a = _f1600(a)
    out = b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


def keccak256(data: bytes) -> bytes:
    return _sponge(data, 0x01)


def sha3_256_via_mini(data: bytes) -> bytes:
    return _sponge(data, 0x06)


def blake2b256(data: bytes) -> bytes:
    return hashlib.blake2b(data, digest_size=32).digest()


def sha256(data: bytes) -> bytes:
    return hashlib.sha256(data).digest()
