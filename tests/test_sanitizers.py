"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the two bodies of native host code (SURVEY.md §5 plan,
VERDICT r1 "missing" #8): the oracle (`make -C oracle asan`, g++) and the engine's host side (`make -C csrc asan`:
csrc/host/*.cpp instrumented, device code and launchers regular).  Each runs an existing CPU test selection in a
subprocess with the sanitizer runtime preloaded (Python itself is not instrumented) and must finish clean:
`halt_on_error=1` turns any report into a failing exit code.  Leak checking is off — the interpreter never frees."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV_BASE = {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _gcc_file(name):
    try:
        p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run_pytest(selection, env_extra, timeout):
    env = dict(os.environ, **ENV_BASE, **env_extra)
    p = subprocess.run([sys.executable, "-m", "pytest", *selection, "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail, tail


def test_oracle_under_asan_ubsan():
    asan, stdcpp = _gcc_file("libasan.so"), _gcc_file("libstdc++.so.6")
    if not asan or not stdcpp or not shutil.which("make"):
        pytest.skip("no gcc sanitizer runtime here")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, timeout=900)
    lib = os.path.join(ROOT, "oracle", "_asan", "libipcfp_oracle.so")
    # libstdc++ goes in with the runtime: ASan resolves __cxa_throw when it starts, and python links no C++
    _run_pytest(["tests/test_oracle_kat.py", "tests/test_oracle_hashes.py", "tests/test_oracle_generate.py",
                 "tests/test_golden.py", "tests/test_config1_plumbing.py"],
                {"LD_PRELOAD": f"{asan} {stdcpp}", "IPCFP_ORACLE_LIB": lib}, timeout=900)


def test_engine_host_side_under_asan_ubsan():
    rt = "/opt/rocm/lib/llvm/lib/clang"
    runtimes = []
    if os.path.isdir(rt):
        for v in sorted(os.listdir(rt)):
            c = os.path.join(rt, v, "lib", "linux", "libclang_rt.asan-x86_64.so")
            if os.path.exists(c):
                runtimes.append(c)
    stdcpp = _gcc_file("libstdc++.so.6")
    if not runtimes or not stdcpp or not shutil.which("make"):
        pytest.skip("no clang sanitizer runtime here")
    subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "ipc-filecoin-proofs_amd", "csrc"), "asan"], check=True,
                   timeout=1800)
    lib = os.path.join(ROOT, "ipc-filecoin-proofs_amd", "libipcfp_asan.so")
    # the host-only entry points: CID strings, claim lowering (parallel), the bundle tokeniser, symbol table
    _run_pytest(["tests/test_host_strings.py", "tests/test_pack_claims.py", "tests/test_bundle_tokeniser.py",
                 "tests/test_abi_symbols.py"],
                {"LD_PRELOAD": f"{runtimes[-1]} {stdcpp}", "IPCFP_LIB": lib}, timeout=900)
