"""CPU: the host half of the bundle parser (csrc/host/json_min.h + bundle.cpp, through
ipcfp_bundle_check_json — no GPU involved) against the restated serde rules (tests/bundle_ref.py):
the case table, and a seeded mutation fuzz of a well-formed bundle."""
import random

import pytest

import bundle_cases
import bundle_ref
import ipc_filecoin_proofs_amd as ipcfp


def ref_accepts(text: str):
    try:
        p = bundle_ref.parse_bundle(text, check_content=False)
        return True, (len(p["storage_proofs"]), len(p["event_proofs"]), len(p["blocks"]))
    except bundle_ref.BundleError:
        return False, None


def engine_accepts(text: str):
    ok, ns, ne, nb, err = ipcfp.bundle_check_json(text.encode("utf-8", "surrogatepass"))
    return ok, (ns, ne, nb), err


def test_case_table():
    for name, text in bundle_cases.cases():
        want, counts = ref_accepts(text)
        got, gcounts, err = engine_accepts(text)
        assert got == want, (name, err)
        if want:
            assert gcounts == counts, name


def test_content_checks_are_the_devices():
    """Cases the full rules reject but the host half lets through must be exactly the content ones."""
    host_only = []
    for name, text in bundle_cases.cases():
        try:
            bundle_ref.parse_bundle(text)
            continue
        except bundle_ref.BundleError:
            pass
        if ref_accepts(text)[0]:
            host_only.append(name)
    assert all(n.startswith("block ") for n in host_only)
    assert {"block only pads", "block nonzero tail 2", "block url alphabet", "block cid 256", "block cid short"} <= set(host_only)


ALPHABET = list('{}[]:,"\\ \n0123456789-+.eE') + ["true", "false", "null", "u00e9", "\\u0041", "\\ud800", "\\n", "=", "A", "é", "\x01",
                                                 '"cid"', '"data"', '"blocks"', '"slot"', "1e5", "-0", "18446744073709551616"]


def mutate(rng, text):
    k = rng.randrange(4)
    i = rng.randrange(len(text) + 1)
    if k == 0 and text:  # delete a run
        j = min(len(text), i + rng.randrange(1, 4))
        return text[:i] + text[j:]
    if k == 1:  # insert
        return text[:i] + rng.choice(ALPHABET) + text[i:]
    if k == 2 and text:  # replace
        i = min(i, len(text) - 1)
        return text[:i] + rng.choice(ALPHABET) + text[i + 1:]
    # duplicate a slice somewhere else
    a = rng.randrange(len(text))
    b = min(len(text), a + rng.randrange(1, 40))
    return text[:i] + text[a:b] + text[i:]


import re

_CID_BODY = re.compile(r'("cid"\s*:\s*\[)[^\]]*(\])')
_DATA_BODY = re.compile(r'("data"\s*:\s*")([^"\\]*)(")')


def blank_cid_bodies(text: str) -> str:
    """What lies between a cid's `[` and the first `]` is checked on the DEVICE (k_parse_cid_arrays), not by the
    host half under test here: both sides get it emptied."""
    text = _CID_BODY.sub(r"\1\2", text)
    # likewise the characters of an (unescaped) `data` string: the base64 alphabet is the device's check
    return _DATA_BODY.sub(lambda m: m.group(1) + "A" * (len(m.group(2)) // 4 * 4 + (0 if len(m.group(2)) % 4 == 0 else len(m.group(2)) % 4)) + m.group(3), text)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_mutation_fuzz(seed):
    import json

    st, ev, bl = bundle_cases.base_parts()
    compact = bundle_ref.bundle_json(st, ev, bl[:4])
    pretty = json.dumps(json.loads(compact), indent=1)
    rng = random.Random(seed)
    accepted = 0
    for it in range(4000):
        text = compact if it % 3 else pretty
        for _ in range(rng.randrange(1, 4)):
            text = mutate(rng, text)
        text = blank_cid_bodies(text)
        want, counts = ref_accepts(text)
        got, gcounts, err = engine_accepts(text)
        assert got == want, (seed, it, err, text)
        if want:
            accepted += 1
            assert gcounts == counts
    assert 100 < accepted < 3800  # the fuzz explores both sides
