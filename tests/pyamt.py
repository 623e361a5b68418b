"""A small AMT *writer* in Python for tests that need shapes the synthetic tipset writer never produces
(sparse indices, lying counts, non-minimal heights).  Wire format per SURVEY.md A.5: v0 root
`[height, count, node]` (bit width 3), v3 root `[bit_width, height, count, node]`, node
`[bitmap bytes (LSB-first), [links…], [values…]]`, CID = CIDv1 dag-cbor blake2b-256."""
import hashlib


def head(major: int, n: int) -> bytes:
    m = major << 5
    if n < 24:
        return bytes([m | n])
    if n < 1 << 8:
        return bytes([m | 24, n])
    if n < 1 << 16:
        return bytes([m | 25]) + n.to_bytes(2, "big")
    if n < 1 << 32:
        return bytes([m | 26]) + n.to_bytes(4, "big")
    return bytes([m | 27]) + n.to_bytes(8, "big")


def uint(n):
    return head(0, n)


def bstr(b):
    return head(2, len(b)) + bytes(b)


def array(items):
    return head(4, len(items)) + b"".join(items)


def link(cid: bytes):
    return b"\xd8\x2a" + bstr(b"\x00" + cid)


NULL = b"\xf6"


def cid_of(block: bytes) -> bytes:
    return bytes.fromhex("0171a0e40220") + hashlib.blake2b(block, digest_size=32).digest()


class Store:
    def __init__(self):
        self.blocks = {}  # cid -> bytes, insertion ordered

    def put(self, block: bytes, sha256_cid: bool = False) -> bytes:
        """Store under the usual blake2b-256 CID, or under a CIDv1 dag-cbor sha2-256 CID (36 bytes: a legal
        link of a different length — the witness store never re-hashes)."""
        c = (bytes.fromhex("01711220") + hashlib.sha256(block).digest()) if sha256_cid else cid_of(block)
        self.blocks[c] = block
        return c

    def tables(self):
        import numpy as np
        cids = list(self.blocks)
        data = b"".join(self.blocks[c] for c in cids)
        lens = np.array([len(self.blocks[c]) for c in cids], dtype=np.uint32)
        off = np.zeros(len(cids), dtype=np.uint64)
        if len(cids):
            off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        c40 = np.zeros((len(cids), 40), dtype=np.uint8)
        for i, c in enumerate(cids):
            c40[i, : len(c)] = np.frombuffer(c, dtype=np.uint8)
        return np.frombuffer(data, dtype=np.uint8).copy(), off, lens, c40


def _node(store, bw, height, items, base, odd_links=()):
    """items: sorted [(index, encoded value)] inside [base, base + W^(height+1)) → encoded node bytes.
    odd_links: bases of child nodes that are linked through a sha2-256 CID instead of the standard one."""
    W = 1 << bw
    bmap = bytearray((W + 7) // 8)
    links, values = [], []
    if height == 0:
        for i, v in items:
            s = i - base
            bmap[s >> 3] |= 1 << (s & 7)
            values.append(v)
    else:
        span = W ** height
        for s in range(W):
            sub = [(i, v) for i, v in items if base + s * span <= i < base + (s + 1) * span]
            if sub:
                bmap[s >> 3] |= 1 << (s & 7)
                child_base = base + s * span
                child = _node(store, bw, height - 1, sub, child_base, odd_links)
                links.append(link(store.put(child, sha256_cid=(height - 1, child_base) in odd_links)))
    return array([bstr(bmap), array(links), array(values)])


def build_amt(store, items, version=0, bit_width=3, height=None, count=None, odd_links=()):
    """items: {index: encoded value}.  Returns the root CID.  `height` / `count` may lie.
    odd_links: {(child height, child base index)} linked through a 36-byte sha2-256 CID."""
    bw = 3 if version == 0 else bit_width
    W = 1 << bw
    srt = sorted(items.items())
    h = 0
    top = srt[-1][0] if srt else 0
    while top >= W ** (h + 1):
        h += 1
    if height is not None:
        h = height
    node = _node(store, bw, h, srt, 0, set(odd_links))
    cnt = len(srt) if count is None else count
    root = array(([uint(bw)] if version != 0 else []) + [uint(h), uint(cnt), node])
    return store.put(root)


def receipt(exit_code=0, ret=b"", gas=1000, events_root=None):
    return array([uint(exit_code), bstr(ret), uint(gas), NULL if events_root is None else link(events_root)])
