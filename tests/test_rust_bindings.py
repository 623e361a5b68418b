"""CPU: the Rust side of the boundary stays complete and in step with include/ipcfp.h.  There is no Rust toolchain in
the image, so nothing is compiled; what CAN be checked is that bindings/rust/ffi_sys.rs (generated) declares every entry
point of the header with the header's arity, and that ffi.rs — the safe wrappers mirroring the reference's
`verify_event_proof` / `verify_storage_proof` / `generate_proof_bundle` / `Blockstore` (src/proofs/events/verifier.rs:51-56,
src/proofs/storage/verifier.rs:24-28, src/proofs/generator.rs:25-31, src/proofs/common/blockstore.rs:26-39) — only calls
functions that exist."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_rust_ffi  # noqa: E402
from test_abi_symbols import declared_symbols  # noqa: E402

RUST = os.path.join(ROOT, "bindings", "rust")


def test_ffi_sys_is_what_the_generator_writes():
    assert open(os.path.join(RUST, "ffi_sys.rs")).read() == gen_rust_ffi.render()


def test_ffi_sys_declares_every_header_symbol_with_its_arity():
    protos = {name: args for _, name, args in gen_rust_ffi.prototypes()}
    assert sorted(protos) == declared_symbols()
    text = open(os.path.join(RUST, "ffi_sys.rs")).read()
    for name, args in protos.items():
        m = re.search(r"pub fn %s\(([^)]*)\)" % name, text)
        assert m, name
        n_c = 0 if args in ("", "void") else args.count(",") + 1
        n_r = 0 if not m.group(1).strip() else m.group(1).count(",") + 1
        assert n_c == n_r, name


def test_rust_type_lowering():
    rt = gen_rust_ffi.rust_type
    assert rt("const uint8_t*") == "*const u8"
    assert rt("uint64_t *") == "*mut u64"
    assert rt("const char* const*") == "*const *const c_char"
    assert rt("ipcfp_ctx_t**") == "*mut *mut ipcfp_ctx_t"
    assert rt("int") == "c_int"


def test_wrappers_only_call_declared_functions_and_cover_the_reference_api():
    text = open(os.path.join(RUST, "ffi.rs"), "rb").read().decode("utf-8", "replace")
    called = set(re.findall(r"\b(ipcfp_[a-z0-9_]+)\s*\(", text))
    unknown = sorted(called - set(declared_symbols()))
    assert not unknown, unknown
    for wrapper in ("fn verify_event_proof", "fn verify_storage_proof", "fn verify_proof_bundle", "fn generate_proof_bundle",
                    "impl Blockstore for"):
        assert wrapper in text, wrapper
    # interior NULs are an Err, not a panic
    assert "CString::new" not in text or ".unwrap()" not in text.split("CString::new", 1)[1].split(";", 1)[0]


def test_rust_constants_follow_the_header():
    """a struct whose size depends on a header constant (ipcfp_tipset_ref_t: IPCFP_MAX_PARENTS) must move the ABI
    version, and every binding compares versions before its first call."""
    header = open(os.path.join(ROOT, "include", "ipcfp.h")).read()
    text = open(os.path.join(RUST, "ffi.rs"), "rb").read().decode("utf-8", "replace")
    for name in ("IPCFP_ABI_VERSION", "IPCFP_MAX_PARENTS", "IPCFP_CID_SLOT", "IPCFP_SCAN_PHASE_RECEIPTS", "IPCFP_SCAN_PHASE_EVENTS"):
        h = int(re.search(r"#define %s (\d+)" % name, header).group(1))
        r = int(re.search(r"pub const %s: \w+ = (\d+);" % name, text).group(1))
        assert h == r, name
    assert "ipcfp_abi_version()" in text.split("pub fn new(device: i32)", 1)[1].split("ipcfp_ctx_create", 1)[0]
