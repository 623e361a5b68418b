"""A small HAMT *writer* in Python (the second independent writer of the v3 format next to tools/synth's C++
one).  Wire format per SURVEY.md A.6 (fvm_ipld_hamt 0.10, `Hamt<BS, V, BytesKey, Sha256>`): a node is
`[bitfield: bytes, pointers: [P…]]`; the bitfield is a 2^bit_width-bit integer written big-endian with leading
zero bytes stripped; a pointer is a tag-42 link or a bucket `[[key bytes, value]…]` of at most 3 pairs sorted by
key; the child index at depth d is bits [d·bw, (d+1)·bw) of SHA-256(key), most significant bit first."""
import hashlib

from pyamt import array, bstr, link

BUCKET = 3


def index_at(key: bytes, depth: int, bw: int) -> int:
    h = int.from_bytes(hashlib.sha256(key).digest(), "big")
    return (h >> (256 - bw * (depth + 1))) & ((1 << bw) - 1)


def _node(store, items, depth, bw):
    """items: [(key bytes, encoded value)] → encoded node bytes"""
    groups = {}
    for k, v in items:
        groups.setdefault(index_at(k, depth, bw), []).append((k, v))
    bitfield = 0
    pointers = []
    for idx in sorted(groups):
        bitfield |= 1 << idx
        g = groups[idx]
        if len(g) <= BUCKET:
            pointers.append(array([array([bstr(k), v]) for k, v in sorted(g)]))
        else:
            pointers.append(link(store.put(_node(store, g, depth + 1, bw))))
    nbytes = (bitfield.bit_length() + 7) // 8
    return array([bstr(bitfield.to_bytes(nbytes, "big")), array(pointers)])


def build_hamt(store, items: dict, bit_width=5) -> bytes:
    """items: {key bytes: encoded value}.  Returns the root CID (the root block IS a node)."""
    return store.put(_node(store, list(items.items()), 0, bit_width))
