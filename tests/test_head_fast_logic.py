"""CPU: the arithmetic of the two head fast paths of the event parse — `decode_event_head_fast`
(csrc/kernels/event_log_dev.h: `82`, emitter, entries-array header of a StampedEvent) and `amt_leaf_root_fast`
(csrc/kernels/block_events.hip: the root of an events AMT whose node is a leaf) — transcribed line by line and held
against the item-by-item decode the kernels fall back to, on random and mutated inputs.  The claim the kernels rely on:
whenever a fast path accepts, the general decode accepts too and yields the same values and the same position; whenever
it declines, nothing was consumed.  (The device code itself is exercised by the GPU parity and fuzz suites; this test
pins the case analysis, like tests/test_entry_fast_logic.py does for the entry decode.)"""
import random

from test_entry_fast_logic import Bad, head

M64 = (1 << 64) - 1


def bytes_from(w0: int, w1: int, o: int) -> int:
    """transcription of bytes_from: the 8 bytes that start at byte o (o <= 8) of the 16 bytes (w0, w1)"""
    sh = 8 * (o & 7)
    lo, hi = (w0, w1) if o < 8 else (w1, 0)
    return ((lo >> sh) | (((hi << 1) & M64) << (63 - sh))) & M64


def bswap64(x: int) -> int:
    return int.from_bytes(x.to_bytes(8, "little"), "big")


# ---------------------------------------------------------------- the head of a StampedEvent
def event_head_fast(buf: bytes, p0: int, n: int):
    if p0 > n or n - p0 < 3:
        return None
    w0 = int.from_bytes(buf[p0:p0 + 8], "little")
    w1 = int.from_bytes(buf[p0 + 8:p0 + 16], "little")
    b0, b1 = w0 & 0xFF, (w0 >> 8) & 0xFF
    ai = b1 & 31
    ok = b0 == 0x82 and (b1 >> 5) == 0 and ai <= 27
    nb = 0 if ai < 24 else (1 << ((ai - 24) & 3))
    arg = bswap64(bytes_from(w0, w1, 2))
    em = ai if ai < 24 else (arg if nb == 8 else (arg >> ((64 - 8 * nb) & 63)))
    ho = 2 + nb
    hw = bytes_from(w0, w1, 8 if ho > 8 else ho) >> (8 * (ho - 8) if ho > 8 else 0)
    h = hw & 0xFF
    h_imm, h_1 = (h >> 5) == 4 and (h & 31) < 24, h == 0x98
    ok = ok and (h_imm or h_1)
    total = ho + (1 if h_imm else 2)
    ok = ok and total <= n - p0
    if not ok:
        return None
    ne = (h & 31) if h_imm else ((hw >> 8) & 0xFF)
    return em, ne, p0 + total


def event_head_general(buf: bytes, p0: int, n: int):
    """expect_array(2); read_uint; read_array"""
    try:
        m, a, pos = head(buf, p0, n)
        if m != 4 or a != 2:
            raise Bad
        m, em, pos = head(buf, pos, n)
        if m != 0:
            raise Bad
        m, ne, pos = head(buf, pos, n)
        if m != 4:
            raise Bad
        return em, ne, pos
    except Bad:
        return None


def uint(x: int) -> bytes:
    if x < 24:
        return bytes([x])
    for k, lead in ((1, 0x18), (2, 0x19), (4, 0x1A), (8, 0x1B)):
        if x < (1 << (8 * k)):
            return bytes([lead]) + x.to_bytes(k, "big")
    raise ValueError


def hdr(major: int, x: int) -> bytes:
    u = uint(x)
    return bytes([u[0] | (major << 5)]) + u[1:]


def test_event_head_fast_agrees_with_the_general_decode():
    rng = random.Random(20260922)
    accepted = declined = 0
    for it in range(120_000):
        emitter = rng.choice([0, 5, 23, 24, 255, 256, 1001, 65535, 65536, 2 ** 32 - 1, 2 ** 32, 2 ** 63, 2 ** 64 - 1,
                              rng.randrange(2 ** 64)])
        ne = rng.choice([0, 1, 3, 5, 23, 24, 100, 255, 256, 70000])
        e = bytearray(b"\x82" + uint(emitter) + hdr(4, ne))
        if it % 7 == 6:  # a non-minimal emitter: the general decode accepts it, so the fast path may as well
            e = bytearray(b"\x82" + bytes([0x1B]) + (emitter & 0xFFFF).to_bytes(8, "big") + hdr(4, ne))
        kind = it % 5
        if kind == 1:
            e[rng.randrange(len(e))] = rng.randrange(256)
        elif kind == 2:
            e = e[:rng.randrange(len(e) + 1)]
        elif kind == 3:
            e = bytearray(rng.randbytes(rng.randrange(0, 16)))
        pre = rng.randbytes(rng.randrange(0, 9))
        body = rng.randbytes(rng.randrange(0, 6))  # entries may or may not follow inside the item
        n = len(pre) + len(e) + len(body) - (rng.randrange(0, 3) if kind == 4 else 0)
        n = max(n, len(pre))
        buf = bytes(pre + e + body) + rng.randbytes(32)
        p0 = len(pre)
        f, g = event_head_fast(buf, p0, n), event_head_general(buf, p0, n)
        if f is not None:
            accepted += 1
            assert g == f, (buf[p0:p0 + 16].hex(), n - p0, f, g)
        else:
            declined += 1
    assert accepted > 20_000 and declined > 20_000, (accepted, declined)


# ---------------------------------------------------------------- the root of an events AMT whose node is a leaf
def leaf_root_fast(buf: bytes, n: int):
    """transcription of amt_leaf_root_fast (reader at position 0): → (nv, bits, pos') or None"""
    if n < 8:
        return None
    w0 = int.from_bytes(buf[0:8], "little")
    w1 = int.from_bytes(buf[8:16], "little")
    b0, bw, b2, b3 = w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, (w0 >> 24) & 0xFF
    ok = b0 == 0x84 and 1 <= bw <= 6 and b2 == 0
    c_imm, c_1 = b3 < 0x18, b3 == 0x18
    ok = ok and (c_imm or c_1)
    o = 4 if c_imm else 5
    bwq = bw if 1 <= bw <= 6 else 1
    width = 1 << bwq
    bl = (width + 7) // 8
    nw = bytes_from(w0, w1, o)
    ok = ok and (nw & 0xFFFF) == (0x83 | ((0x40 + bl) << 8))
    bo = o + 2
    bm = bytes_from(w0, w1, bo)
    to = bo + bl
    ok = ok and to + 2 <= 16
    tq = to if to + 2 <= 16 else 8
    tw = bytes_from(w0, w1, 8 if tq > 8 else tq) >> (8 * (tq - 8) if tq > 8 else 0)
    l, v, v2 = tw & 0xFF, (tw >> 8) & 0xFF, (tw >> 16) & 0xFF
    v_imm = (v >> 5) == 4 and (v & 31) < 24
    v_1 = v == 0x98 and tq + 3 <= 16
    ok = ok and l == 0x80 and (v_imm or v_1)
    if bl < 8:
        bm &= (1 << (8 * bl)) - 1
    if width < 64:
        bm &= (1 << width) - 1
    nvals = (v & 31) if v_imm else v2
    total = to + (2 if v_imm else 3)
    ok = ok and total <= n and nvals == bin(bm).count("1")
    if not ok:
        return None
    return nvals, bm, total


def leaf_root_general(buf: bytes, n: int):
    """block_events_parse's item-by-item form: → (nv, bits, pos') when the block is tabulated, else None"""
    try:
        m, a, pos = head(buf, 0, n)
        if m != 4 or a != 4:
            raise Bad
        m, bw, pos = head(buf, pos, n)
        if m != 0 or bw < 1 or bw > 6:
            raise Bad
        m, height, pos = head(buf, pos, n)
        if m != 0:
            raise Bad
        m, _count, pos = head(buf, pos, n)
        if m != 0:
            raise Bad
        if height != 0:
            return None
        width = 1 << bw
        m, a, pos = head(buf, pos, n)
        if m != 4 or a != 3:
            raise Bad
        m, bl, pos = head(buf, pos, n)
        if m != 2 or bl > n - pos:
            raise Bad
        bo = pos
        pos += bl
        if bl != (width + 7) // 8:
            return None
        bits = int.from_bytes(buf[bo:bo + 8], "little")
        if bl < 8:
            bits &= (1 << (8 * bl)) - 1
        if width < 64:
            bits &= (1 << width) - 1
        m, nl, pos = head(buf, pos, n)
        if m != 4:
            raise Bad
        if nl != 0:
            return None
        m, nvals, pos = head(buf, pos, n)
        if m != 4:
            raise Bad
        if nvals != bin(bits).count("1"):
            return None
        return nvals, bits, pos
    except Bad:
        return None


def test_amt_leaf_root_fast_agrees_with_the_general_decode():
    rng = random.Random(20260923)
    accepted = declined = 0
    for it in range(120_000):
        bw = rng.choice([1, 2, 3, 4, 5, 5, 5, 6, 7, 0])
        width = 1 << (bw if 1 <= bw <= 6 else 3)
        bl = (width + 7) // 8
        nv = rng.randrange(0, min(width, 40) + 1)
        idx = rng.sample(range(width), nv)
        bits = sum(1 << i for i in idx)
        if it % 11 == 10:
            bits ^= 1 << rng.randrange(width)  # bitmap and value count disagree
        count = rng.choice([nv, 0, 23, 24, 255, 256, 70000])
        height = 0 if it % 13 else rng.choice([1, 2])
        bmap = bits.to_bytes(bl, "little")
        if it % 17 == 16:
            bmap = bmap + b"\x00"  # a bitmap of the wrong length
        links = hdr(4, 0) if it % 19 else hdr(4, 1) + b"\xd8\x2a\x58\x27" + rng.randbytes(39)
        e = bytearray(b"\x84" + uint(bw) + uint(height) + uint(count) + b"\x83" + hdr(2, len(bmap)) + bmap + links + hdr(4, nv))
        kind = it % 5
        if kind == 1:
            e[rng.randrange(min(len(e), 18))] = rng.randrange(256)
        elif kind == 2:
            e = e[:rng.randrange(len(e) + 1)]
        elif kind == 3:
            e = bytearray(rng.randbytes(rng.randrange(0, 24)))
        body = rng.randbytes(rng.randrange(0, 12))
        n = len(e) + len(body) - (rng.randrange(0, 3) if kind == 4 else 0)
        n = max(n, 0)
        buf = bytes(e + body) + rng.randbytes(32)
        f, g = leaf_root_fast(buf, n), leaf_root_general(buf, n)
        if f is not None:
            accepted += 1
            assert g == f, (buf[:20].hex(), n, f, g)
        else:
            declined += 1
    assert accepted > 15_000 and declined > 15_000, (accepted, declined)
