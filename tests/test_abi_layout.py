"""CPU: struct layouts across the ABI are COMPARED, not trusted.

include/ipcfp.h is mirrored by hand three times: binding.py's numpy dtypes and ctypes Structures (and tests/claims.py's),
and the `#[repr(C)]` types of bindings/rust/ffi.rs.  ADVICE r4 #2 was a drift between them (ipcfp_tipset_ref_t: 648 → 1 288
bytes).  tools/gen_abi_layout.py asks the C compiler for `sizeof` / `offsetof` of every public struct; here every mirror is
held to those numbers field by field, and a field added to the header alone is shown to fail.
Boundary: SURVEY.md §8(b); src/proofs/events/bundle.rs:5-23, src/proofs/storage/bundle.rs:5-14."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import claims  # noqa: E402
import gen_abi_layout  # noqa: E402
import ipc_filecoin_proofs_amd as ipcfp  # noqa: E402
from ipc_filecoin_proofs_amd import binding  # noqa: E402

HEADER = os.path.join(ROOT, "include", "ipcfp.h")

# every public struct → its Python mirror(s)
NUMPY = {
    "ipcfp_value_loc_t": binding.LOC_DTYPE,
    "ipcfp_tipset_ref_t": binding.TIPSET_DTYPE,
    "ipcfp_event_claim_t": binding.CLAIM_DTYPE,
    "ipcfp_event_claim_compact_t": binding.COMPACT_DTYPE,
    "ipcfp_event_claim_group_t": binding.GROUP_DTYPE,
    "ipcfp_storage_claim_t": binding.SCLAIM_DTYPE,
    "ipcfp_event_match_t": binding.MATCH_DTYPE,          # (`event` flattened into block / off / len)
    "ipcfp_generated_storage_t": binding.GEN_STORAGE_DTYPE,
    "ipcfp_storage_proof_spec_t": binding.STORAGE_SPEC_DTYPE,
}
CTYPES = {
    "ipcfp_event_proof_spec_t": binding.EventProofSpec,
    "ipcfp_shard_pull_stats_t": binding.ShardPullStats,
    "ipcfp_event_proof_t": claims.EventProof,
    "ipcfp_storage_proof_t": claims.StorageProof,
    "ipcfp_event_filter_t": claims.EventFilter,
    "ipcfp_trust_policy_t": claims.TrustPolicy,
}
FLATTENED = {"ipcfp_event_match_t": {"event": ("block", "off", "len")}}


@pytest.fixture(scope="module")
def lay():
    return gen_abi_layout.layout()


def numpy_diffs(name, rec, dt):
    out = []
    if dt.itemsize != rec["size"]:
        out.append("%s: numpy itemsize %d, C sizeof %d" % (name, dt.itemsize, rec["size"]))
    want = {}
    for f, (off, size) in rec["fields"].items():
        sub = FLATTENED.get(name, {}).get(f)
        if sub:
            inner = gen_abi_layout.layout()["ipcfp_value_loc_t"]["fields"]
            for s in sub:
                want[s] = (off + inner[s][0], inner[s][1])
        else:
            want[f] = (off, size)
    have = {f: (dt.fields[f][1], dt.fields[f][0].itemsize) for f in dt.names}
    if list(have) != list(want):
        out.append("%s: numpy fields %s, C fields %s" % (name, list(have), list(want)))
    for f in want:
        if f in have and have[f] != want[f]:
            out.append("%s.%s: numpy (offset, size) %s, C %s" % (name, f, have[f], want[f]))
    return out


def ctypes_diffs(name, rec, st):
    out = []
    if C.sizeof(st) != rec["size"]:
        out.append("%s: ctypes sizeof %d, C sizeof %d" % (name, C.sizeof(st), rec["size"]))
    have = {f: (getattr(st, f).offset, getattr(st, f).size) for f, _ in st._fields_}
    if list(have) != list(rec["fields"]):
        out.append("%s: ctypes fields %s, C fields %s" % (name, list(have), list(rec["fields"])))
    for f, w in rec["fields"].items():
        if f in have and have[f] != w:
            out.append("%s.%s: ctypes (offset, size) %s, C %s" % (name, f, have[f], w))
    return out


def all_diffs(lay):
    out = []
    for name in (k for k in lay if not k.startswith("_")):
        if name in NUMPY:
            out += numpy_diffs(name, lay[name], NUMPY[name])
        elif name in CTYPES:
            out += ctypes_diffs(name, lay[name], CTYPES[name])
        else:
            out.append("%s: a public struct with no Python mirror to compare" % name)
    return out


def test_probe_source_is_what_the_generator_writes():
    with open(HEADER) as f:
        assert open(gen_abi_layout.OUT_C).read() == gen_abi_layout.render_c(f.read())


def test_probe_sees_every_public_struct(lay):
    with open(HEADER) as f:
        names = [n for n, _ in gen_abi_layout.structs(f.read())]
    assert len(names) >= 15 and set(names) == {k for k in lay if not k.startswith("_")} == set(NUMPY) | set(CTYPES)
    assert lay["_values"] == {"abi_version": ipcfp.ABI_VERSION, "cid_slot": binding.CID_SLOT, "max_parents": binding.MAX_PARENTS}


def test_python_mirrors_have_the_compilers_layout(lay):
    assert all_diffs(lay) == []
    assert binding.MATCH_DTYPE.itemsize == 40 and binding.CLAIM_DTYPE.itemsize == 104  # (bench.py's MATCH_BYTES; kernels' EventClaimPacked)


# ---- the Rust mirrors: no rustc in the image, so #[repr(C)] layout is computed from the declarations ----
RUST_SCALARS = {"u8": (1, 1), "u16": (2, 2), "u32": (4, 4), "u64": (8, 8), "i64": (8, 8), "f64": (8, 8), "c_int": (4, 4)}


def rust_structs():
    text = open(os.path.join(ROOT, "bindings", "rust", "ffi.rs"), "rb").read().decode("utf-8", "replace")
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+): usize = (\d+);", text)}
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (ipcfp_\w+_t)\s*\{(.*?)\}", text, flags=re.S):
        body = m.group(2)
        if "_p:" in body:
            continue  # opaque handles
        fields, depth, cur = [], 0, ""
        for ch in body:  # split on top-level commas ([[u8; 40]; N] holds none, but stay general)
            depth += ch in "[(<"
            depth -= ch in "])>"
            if ch == "," and depth == 0:
                fields.append(cur)
                cur = ""
            else:
                cur += ch
        fields.append(cur)
        parsed = []
        for f in fields:
            f = " ".join(f.split())
            if not f:
                continue
            fm = re.match(r"^pub (\w+): (.+)$", f)
            assert fm, (m.group(1), f)
            parsed.append((fm.group(1), fm.group(2)))
        out[m.group(1)] = parsed
    return out, consts


def rust_size_align(ty, structs, consts):
    ty = ty.strip()
    if ty.startswith("*") or ty.startswith("Option<"):
        return 8, 8
    am = re.match(r"^\[(.+);\s*(\w+)\]$", ty)
    if am:
        n = consts[am.group(2)] if am.group(2) in consts else int(am.group(2))
        s, a = rust_size_align(am.group(1), structs, consts)
        return s * n, a
    if ty in RUST_SCALARS:
        return RUST_SCALARS[ty]
    size, align, _ = rust_layout(ty, structs, consts)
    return size, align


def rust_layout(name, structs, consts):
    off, align, fields = 0, 1, {}
    for f, ty in structs[name]:
        s, a = rust_size_align(ty, structs, consts)
        off = (off + a - 1) // a * a
        fields[f] = (off, s)
        off += s
        align = max(align, a)
    return (off + align - 1) // align * align, align, fields


def test_rust_mirrors_have_the_compilers_layout(lay):
    structs, consts = rust_structs()
    names = {k for k in lay if not k.startswith("_")}
    assert names <= set(structs), sorted(names - set(structs))
    for name in sorted(names):
        size, align, fields = rust_layout(name, structs, consts)
        assert (size, align) == (lay[name]["size"], lay[name]["align"]), name
        assert fields == lay[name]["fields"], name
    # ... and ffi_sys.rs asserts the same numbers at compile time, for the day a Rust toolchain sees it
    sys_text = open(os.path.join(ROOT, "bindings", "rust", "ffi_sys.rs")).read()
    assert gen_abi_layout.rust_asserts(lay) in sys_text
    for name in names:
        assert "size_of::<%s>() == %d" % (name, lay[name]["size"]) in sys_text


@pytest.mark.parametrize("struct_name, new_field", [
    ("ipcfp_event_claim", "uint32_t extra;"),
    ("ipcfp_shard_pull_stats", "uint64_t extra_bytes;"),
    ("ipcfp_event_match", "uint8_t tag;"),
])
def test_a_field_added_to_the_header_alone_is_caught(tmp_path, struct_name, new_field):
    """Done-criterion of VERDICT r5 #7: change the header, leave the bindings — the comparison must say so."""
    with open(HEADER) as f:
        text = f.read()
    head = "typedef struct %s {" % struct_name
    assert head in text
    changed = text.replace(head, head + "\n    " + new_field, 1)
    p = tmp_path / "ipcfp.h"
    p.write_text(changed)
    lay2 = gen_abi_layout.layout(str(p))
    diffs = all_diffs(lay2)
    assert diffs and any(struct_name in d for d in diffs)
    structs, consts = rust_structs()
    name = struct_name + "_t"
    assert rust_layout(name, structs, consts)[2] != lay2[name]["fields"]
