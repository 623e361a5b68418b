"""CPU: libipcfp.so loads and exports every symbol include/ipcfp.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ipcfp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ipcfp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "ipcfp_ctx_create" in syms and "ipcfp_witness_verify_cids" in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol():
    import ipc_filecoin_proofs_amd as ipcfp

    lib = ipcfp.load_library()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in ipcfp.h but not exported by libipcfp.so: {missing}"
    header = open(os.path.join(ROOT, "include", "ipcfp.h")).read()
    declared = int(re.search(r"#define IPCFP_ABI_VERSION (\d+)", header).group(1))
    assert lib.ipcfp_abi_version() == declared == ipcfp.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the context must refuse to exist (fail loudly)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ipc_filecoin_proofs_amd as ipcfp

    with pytest.raises(ipcfp.EngineError):
        ipcfp.Engine(0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import ipc_filecoin_proofs_amd.binding as b

    monkeypatch.setattr(b, "_lib", None)
    monkeypatch.setenv("IPCFP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(b.EngineError):
        b.load_library()


def test_graft_entry_build_passes_on_the_built_tree():
    """`__graft_entry__.build()` is the driver's "does it build" check: with the tree built it is a no-op make plus the import
    and the ABI-version check — which must follow the header's constant, not a number written down once."""
    import __graft_entry__ as entry

    entry.build()
