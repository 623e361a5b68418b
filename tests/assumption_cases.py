"""The ⚠ ASSUMPTIONS about the un-vendored crates (serde_ipld_dagcbor 0.6, cid 0.10/0.11, fvm_ipld_amt 0.7.4,
fvm_shared 4.7 — Cargo.toml:11-22, no lockfile, no tests in the reference), one hand-written block per assumption.

Neither the oracle nor the engine can be checked against the crates here (no Rust toolchain, no vendored sources): these
cases state, by NAME, what both were written to do, so that the day a real-chain fixture or the crate sources are at
hand each assumption is confirmed or flipped in ONE place — tests/test_assumptions.py runs the table against the
oracle (CPU) and against the engine (GPU) and both must give the status listed here.

Each case: name → (blocks, root cid, AMT version, value kind, index, expected status); statuses as in include/ipcfp.h
(1 = found, 32 = NOT_FOUND, 66 = ERR_DECODE).  The carrier is always a one-level Amt (v3) whose single value, or whose
own header, carries the spelling in question."""
import pyamt
from pyamt import array, bstr, head, link, uint

TRUE, NOT_FOUND, ERR_DECODE = 1, 32, 66


def _leaf_root(value: bytes, bitmap: int = 0x01, count: bytes = None, values_header: bytes = None, trailer: bytes = b""):
    """v3 root [bit_width 3, height 0, count, [bitmap, [], [value]]] as raw bytes"""
    vals = (values_header if values_header is not None else head(4, 1)) + value
    node = head(4, 3) + bstr(bytes([bitmap])) + head(4, 0) + vals
    return head(4, 4) + uint(3) + uint(0) + (count if count is not None else uint(1)) + node + trailer


def _case(block: bytes, kind: str, index: int, expect: int):
    s = pyamt.Store()
    root = s.put(block)
    return s, root, 3, kind, index, expect


CID = pyamt.cid_of(b"\x80")
CASES = {
    # --- serde_ipld_dagcbor: what a strict DAG-CBOR decode accepts -------------------------------------------------
    "baseline_minimal_encoding_is_found": lambda: _case(_leaf_root(uint(7)), "any", 0, TRUE),
    "nonminimal_uint_argument_is_accepted": lambda: _case(_leaf_root(uint(7), count=b"\x18\x01"), "any", 0, TRUE),
    "nonminimal_array_length_is_accepted": lambda: _case(_leaf_root(uint(7), values_header=b"\x98\x01"), "any", 0, TRUE),
    "indefinite_length_array_is_rejected": lambda: _case(_leaf_root(uint(7) + b"\xff", values_header=b"\x9f"), "any", 0, ERR_DECODE),
    "float64_is_accepted": lambda: _case(_leaf_root(b"\xfb" + bytes(8)), "any", 0, TRUE),
    "float32_is_rejected": lambda: _case(_leaf_root(b"\xfa" + bytes(4)), "any", 0, ERR_DECODE),
    "float16_is_rejected": lambda: _case(_leaf_root(b"\xf9" + bytes(2)), "any", 0, ERR_DECODE),
    "undefined_simple_value_is_rejected": lambda: _case(_leaf_root(b"\xf7"), "any", 0, ERR_DECODE),
    "only_tag_42_is_a_legal_tag": lambda: _case(_leaf_root(b"\xc1" + uint(5)), "any", 0, ERR_DECODE),
    "text_must_be_utf8": lambda: _case(_leaf_root(head(3, 2) + b"\xc3\x28"), "any", 0, ERR_DECODE),
    "trailing_bytes_after_the_block_item_are_rejected": lambda: _case(_leaf_root(uint(7), trailer=b"\x00"), "any", 0, ERR_DECODE),
    # --- cid: the bytes of a link ---------------------------------------------------------------------------------------
    "link_is_tag42_bytes_with_identity_prefix": lambda: _case(_leaf_root(link(CID)), "cid", 0, TRUE),
    "link_without_the_00_multibase_prefix_is_rejected": lambda: _case(_leaf_root(b"\xd8\x2a" + bstr(CID)), "cid", 0, ERR_DECODE),
    "cid_with_bytes_after_the_digest_is_rejected": lambda: _case(_leaf_root(b"\xd8\x2a" + bstr(b"\x00" + CID + b"\x00")), "cid", 0, ERR_DECODE),
    "cid_with_nonminimal_varint_is_rejected": lambda: _case(_leaf_root(b"\xd8\x2a" + bstr(b"\x00\x81\x00" + CID[1:])), "cid", 0, ERR_DECODE),
    # --- fvm_ipld_amt: the node --------------------------------------------------------------------------------------------
    "amt_bitmap_is_lsb_first_bit1_is_index1": lambda: _case(_leaf_root(uint(7), bitmap=0x02), "any", 1, TRUE),
    "amt_bitmap_is_lsb_first_bit1_is_not_index6": lambda: _case(_leaf_root(uint(7), bitmap=0x02), "any", 6, NOT_FOUND),
    "amt_value_count_must_match_the_bitmap": lambda: _case(_leaf_root(uint(7), bitmap=0x03), "any", 0, ERR_DECODE),
    "amt_count_field_is_not_checked_by_load_or_get": lambda: _case(_leaf_root(uint(7), count=uint(9)), "any", 0, TRUE),
    # --- fvm_shared / serde: typed values ----------------------------------------------------------------------------------
    "vec_u8_is_an_array_of_small_uints": lambda: _case(_leaf_root(array([uint(1), uint(255)])), "vec_u8", 0, TRUE),
    "vec_u8_is_not_a_byte_string": lambda: _case(_leaf_root(bstr(b"\x01\x02")), "vec_u8", 0, ERR_DECODE),
    "vec_u8_element_above_255_is_rejected": lambda: _case(_leaf_root(array([uint(256)])), "vec_u8", 0, ERR_DECODE),
    "receipt_is_a_4_tuple_with_nullable_events_root": lambda: _case(_leaf_root(pyamt.receipt()), "receipt", 0, TRUE),
    "receipt_with_a_fifth_field_is_rejected": lambda: _case(_leaf_root(head(4, 5) + uint(0) + bstr(b"") + uint(1) + b"\xf6" + uint(0)), "receipt", 0, ERR_DECODE),
    "receipt_exit_code_above_u32_is_rejected": lambda: _case(_leaf_root(head(4, 4) + uint(1 << 32) + bstr(b"") + uint(1) + b"\xf6"), "receipt", 0, ERR_DECODE),
}
