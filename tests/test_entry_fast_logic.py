"""CPU: the arithmetic of `decode_entry_fast` (csrc/kernels/event_log_dev.h) — the branch-free decode of one
StampedEvent entry `[flags, key, codec, value]` from the 16 bytes at the reader's position — transcribed line by line
and held against a plain item-by-item DAG-CBOR decode of the same bytes on a few hundred thousand random and mutated
inputs.  The claim the kernel relies on: whenever the fast path accepts, the general decode accepts too and yields the
same key offset / length, value offset / length and end position; whenever it declines, nothing was consumed.  (The
device code itself is exercised by the GPU parity and fuzz suites; this test pins the case analysis.)"""
import random

M64 = (1 << 64) - 1


def fast(buf: bytes, p0: int, n: int):
    """transcription of decode_entry_fast; buf is padded so that 16 bytes at p0 are readable"""
    if n - p0 < 6 or p0 > n:
        return None
    w0 = int.from_bytes(buf[p0:p0 + 8], "little")
    w1 = int.from_bytes(buf[p0 + 8:p0 + 16], "little")
    b0, b1, b2 = w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF
    klen = (b2 - 0x60) & 0xFFFFFFFF
    ok = b0 == 0x84 and b1 < 0x18 and klen <= 6
    kq = klen if klen <= 6 else 0
    kbytes = ((w0 >> 24) | (w1 << 40)) & M64
    kmask = (1 << (8 * kq)) - 1
    ok = ok and (kbytes & kmask & 0x8080808080808080) == 0
    c = 3 + kq
    c_lo, c_hi = (c if c < 8 else 7), (0 if c < 8 else c - 8)
    if c < 8:
        rest = ((w0 >> (8 * c_lo)) | (((w1 << 1) & M64) << (63 - 8 * c_lo))) & M64
    else:
        rest = w1 >> (8 * c_hi)
    r0 = rest & 0xFF
    cl = 1 if r0 < 0x18 else 2
    ok = ok and r0 <= 0x18
    v = rest >> (8 * cl)
    vb = v & 0xFF
    v_imm, v_1, v_2 = 0x40 <= vb <= 0x57, vb == 0x58, vb == 0x59
    ok = ok and (v_imm or v_1 or v_2)
    hl = 1 if v_imm else (2 if v_1 else 3)
    len1 = (v >> 8) & 0xFF
    len2 = (len1 << 8) | ((v >> 16) & 0xFF)
    ln = (vb - 0x40) if v_imm else (len1 if v_1 else len2)
    value_at = p0 + c + cl + hl
    ok = ok and value_at <= n and ln <= n - value_at
    if not ok:
        return None
    return p0 + 3, kq, value_at, ln, value_at + ln


class Bad(Exception):
    pass


def head(buf, pos, n):
    if pos >= n:
        raise Bad
    b = buf[pos]
    m, ai = b >> 5, b & 31
    if ai < 24:
        nb, arg = 0, ai
    elif ai <= 27:
        nb = 1 << (ai - 24)
        if nb > n - pos - 1:
            raise Bad
        arg = int.from_bytes(buf[pos + 1:pos + 1 + nb], "big")
    else:
        raise Bad
    if m == 7 and not ((ai < 24 and 20 <= ai <= 22) or ai == 27):
        raise Bad
    return m, arg, pos + 1 + nb


def general(buf: bytes, p0: int, n: int):
    """expect_array(4); read_uint; read_text; read_uint; read_bytes — cbor_dev.h's rules"""
    try:
        m, a, pos = head(buf, p0, n)
        if m != 4 or a != 4:
            raise Bad
        m, a, pos = head(buf, pos, n)
        if m != 0:
            raise Bad
        m, a, pos = head(buf, pos, n)
        if m != 3 or a > n - pos:
            raise Bad
        ko, kl = pos, a
        try:
            bytes(buf[ko:ko + kl]).decode("utf-8")
        except UnicodeDecodeError:
            raise Bad
        pos += kl
        m, a, pos = head(buf, pos, n)
        if m != 0:
            raise Bad
        m, a, pos = head(buf, pos, n)
        if m != 2 or a > n - pos:
            raise Bad
        return ko, kl, pos, a, pos + a
    except Bad:
        return None


def typical_entry(rng):
    key = rng.choice([b"t1", b"t2", b"t3", b"t4", b"d", b"data", b"topics", b"", b"x" * rng.randrange(1, 9), b"\xc3\xa9"])
    flags = rng.choice([0, 3, 0x17, 0x18, 0xFF])
    codec = rng.choice([0x55, 0x17, 0x18, 0x51, 0x100])
    vlen = rng.choice([0, 1, 23, 24, 32, 64, 255, 256, 300])

    def uint(x):
        if x < 24:
            return bytes([x])
        if x < 256:
            return bytes([0x18, x])
        return bytes([0x19]) + x.to_bytes(2, "big")

    def hdr(major, x):
        u = uint(x)
        return bytes([u[0] | (major << 5)]) + u[1:]

    return b"\x84" + uint(flags) + hdr(3, len(key)) + key + uint(codec) + hdr(2, vlen) + rng.randbytes(vlen)


def test_fast_path_agrees_with_the_general_decode():
    rng = random.Random(20260921)
    accepted = declined = 0
    for it in range(150_000):
        e = bytearray(typical_entry(rng))
        kind = it % 5
        if kind == 1 and e:  # flip a byte in the header region
            e[rng.randrange(min(len(e), 14))] = rng.randrange(256)
        elif kind == 2:      # truncate
            e = e[:rng.randrange(len(e) + 1)]
        elif kind == 3:      # random garbage
            e = bytearray(rng.randbytes(rng.randrange(0, 40)))
        pre = rng.randbytes(rng.randrange(0, 9))  # any alignment of the entry
        n = len(pre) + len(e) - (rng.randrange(0, 3) if kind == 4 and e else 0)  # kind 4: the item ends inside the entry
        buf = bytes(pre + e) + rng.randbytes(32)  # what lies behind the item is arbitrary
        p0 = len(pre)
        f, g = fast(buf, p0, n), general(buf, p0, n)
        if f is not None:
            accepted += 1
            assert g == f, (buf[p0:p0 + 20].hex(), n - p0, f, g)
        else:
            declined += 1
    assert accepted > 15_000 and declined > 15_000, (accepted, declined)


def head_branchless(buf: bytes, pos: int, n: int, err: int):
    """transcription of Rd::head in its LDS form (cbor_dev.h, IPCFP_RD_LDS): → (major, arg, pos', err')"""
    inside = pos < n
    at_pos = pos if inside else 0
    raw = int.from_bytes(buf[at_pos:at_pos + 8], "little")
    b = raw & 0xFF
    m, ai = b >> 5, b & 31
    imm = ai < 24
    nb = 0 if imm else (1 << ((ai - 24) & 3))
    bad = (not inside) or ai > 27
    bad = bad or (m == 7 and ((not (20 <= ai <= 22)) if imm else ai != 27))
    bad = bad or (inside and nb > n - at_pos - 1)
    be = int.from_bytes((raw >> 8).to_bytes(8, "little"), "big")  # bswap64(raw >> 8)
    if nb == 8:
        be |= buf[at_pos + 8]
    v = ai if imm else (be if nb == 8 else (be >> ((64 - 8 * nb) & 63)))
    good = (not err) and (not bad)
    err2 = err if err else (66 if bad else 0)
    return (m if good else 8), (v if good else 0), pos + ((1 + nb) if good else 0), err2


def head_reference(buf: bytes, pos: int, n: int, err: int):
    """the windowed reader's head(): early returns"""
    if err:
        return 8, 0, pos, err
    try:
        m, a, p2 = head(buf, pos, n)
    except Bad:
        return 8, 0, pos, 66
    return m, a, p2, 0


def test_branchless_item_header_equals_the_early_return_form():
    rng = random.Random(7)
    for it in range(200_000):
        n = rng.randrange(0, 24)
        body = bytearray(rng.randbytes(n))
        if n and it % 3 == 0:  # bias the initial byte towards every (major, additional-info) pair
            body[0] = rng.randrange(256)
        buf = bytes(body) + rng.randbytes(24)
        pos = rng.randrange(0, n + 2)
        err = 66 if it % 11 == 0 else 0
        assert head_branchless(buf, pos, n, err) == head_reference(buf, pos, n, err), (buf[:n].hex(), pos, n, err)
