"""ipc-filecoin-proofs_amd — MI355X-native batch AMT/HAMT Merkle-witness engine.

The product is ``libipcfp.so`` (hand-written HIP kernels for gfx950 + a C++ host,
behind the C ABI of ``include/ipcfp.h``).  This Python package is only the thin
``ctypes`` binding the tests and ``bench.py`` drive it through; it contains no
compute and NO CPU fallback: if the shared library or a GPU is missing, loading /
context creation fails loudly.

(The directory name contains a hyphen, so ``import ipc_filecoin_proofs_amd`` — a
two-line shim package at the repo root — is the importable spelling.)
"""
from .binding import (  # noqa: F401
    Engine,
    EngineError,
    Witness,
    Bundle,
    bundle_check_json,
    pack_event_proofs,
    pack_storage_proofs,
    GEN_STORAGE_DTYPE,
    lib_path,
    load_library,
    ST,
    CID_OK,
    CID_MISMATCH,
    CID_UNCHECKED,
    KERNEL_IDS,
    pack_event_claims,
    pack_storage_claims,
    SCLAIM_DTYPE,
    cid_from_string,
    cid_to_string,
    pack_cids,
    CLAIM_DTYPE,
    TIPSET_DTYPE,
    TipsetRefs,
    cid_slot,
    cid_slots,
    LOC_DTYPE,
    MATCH_DTYPE,
    witness_cut_host,
    route_event_claims,
)
