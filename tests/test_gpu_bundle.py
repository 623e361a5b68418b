"""GPU: the bundle wire format through the C ABI — JSON → device base64 decode →
HBM witness → verify_proof_bundle — against the restated serde rules (tests/bundle_ref.py) and the CPU oracle."""
import base64

import numpy as np
import pytest

from conftest import fuzz_seed

import bundle_cases
import bundle_ref
import ipc_filecoin_proofs_amd as ipcfp
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


def test_malformed_and_unusual_json_agree_with_the_serde_rules(engine):
    for name, text in bundle_cases.cases():
        try:
            want = bundle_ref.parse_bundle(text)
        except bundle_ref.BundleError:
            want = None
        try:
            b = engine.bundle(text.encode("utf-8", "surrogatepass"))
        except ipcfp.EngineError as e:
            assert want is None, (name, str(e))
            assert "(-7)" in str(e) or "parse" in str(e).lower(), (name, str(e))
            continue
        assert want is not None, name
        assert (b.n_blocks, b.n_events, b.n_storage) == (len(want["blocks"]), len(want["event_proofs"]),
                                                          len(want["storage_proofs"])), name
        b.close()


def test_cid_strings_extension(engine, oracle):
    data = b"\x80"
    cid = oracle.cid_for_block(data)
    text = '{"storage_proofs":[],"event_proofs":[],"blocks":[{"cid":"%s","data":"%s"}]}' % (
        oracle.cid_to_string(cid), base64.b64encode(data).decode())
    with pytest.raises(ipcfp.EngineError):
        engine.bundle(text.encode())
    b = engine.bundle(text.encode(), flags=ipcfp.Bundle.CID_STRINGS)
    st, bad = b.witness.verify_cids()
    assert bad == 0 and st.tolist() == [1]
    b.close()


def test_base64_decode_is_bit_exact(engine, oracle):
    """Every tail length, unaligned string offsets, escaped strings: the decoded blocks hash to their CIDs."""
    rng = np.random.default_rng(fuzz_seed(7))
    blocks = []
    for n in list(range(0, 100)) + [127, 128, 129, 255, 256, 1000, 4096, 65537]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        blocks.append((oracle.cid_for_block(d), d))
    parts = []
    for k, (c, d) in enumerate(blocks):
        s = base64.b64encode(d).decode()
        if k % 7 == 3 and s:
            s = s.replace("/", "\\/").replace("A", "\\u0041", 1)  # legal JSON escapes
        parts.append('{"cid":[%s],%s"data":"%s"}' % (",".join(map(str, c)), " " * (k % 9), s))
    text = '{"storage_proofs":[],"event_proofs":[],"blocks":[%s]}' % ",".join(parts)
    b = engine.bundle(text.encode())
    st, bad = b.witness.verify_cids()
    assert bad == 0 and (st == 1).all() and len(st) == len(blocks)
    b.close()
    # one flipped character anywhere in a data string is caught (by the alphabet check or by the CID)
    for pos in (text.index('"data":"', 4000) + 12, text.rindex('"data":"') + 50):
        bad_text = text[:pos] + ("B" if text[pos] != "B" else "C") + text[pos + 1:]
        b = engine.bundle(bad_text.encode())
        st, bad = b.witness.verify_cids()
        assert bad == 1
        b.close()
        broken = text[:pos] + "*" + text[pos + 1:]
        with pytest.raises(ipcfp.EngineError):
            engine.bundle(broken.encode())


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=3000, n_parents=3, dup_permille=60, n_planted=7, variety=1, max_events=5,
                  no_events_permille=100, n_actors=3000, n_contracts=8, slots_per_contract=12, storage_layout_mix=1,
                  n_actor_queries=12)


def test_generate_serialise_parse_verify(tip, engine, oracle):
    """generate_proof_bundle → JSON → parse (device base64) → verify_proof_bundle, engine vs oracle."""
    w = engine.witness(tip.data, tip.off, tip.lens, tip.cids)
    gs, gm, gmsg, gids = w.generate_event_proofs(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1)
    assert gs == 1 and len(gm) > 0
    sidx = list(range(len(tip.sc_actor)))
    out, sids = w.generate_storage_proofs(tip.child_cid, tip.sc_actor, tip.sc_slot)
    assert (out["status"] == 1).all()
    w.close()
    ids = sorted(set(gids.tolist()) | set(sids.tolist()), key=lambda i: tip.cids[i, 6:38].tobytes())
    blocks = [(tip.cids[i, :38].tobytes(), tip.block(i)) for i in ids]
    events = bundle_ref.event_dicts(tip, generated=(gm, gmsg))
    storage = bundle_ref.storage_dicts(tip, sidx)
    # adversarial claims ride along: wrong value, wrong message CID, out-of-range index, garbage CID string
    storage += [dict(storage[0], value="0x" + "ff" * 32), dict(storage[1], storage_root="not-a-cid")]
    events += [dict(events[0], message_cid=events[-1]["message_cid"] if len(events) > 1 else events[0]["child_block_cid"]),
               dict(events[0], exec_index=10 ** 9), dict(events[0], data="0x00")]
    # "\u0000" inside a claim string: the suffix must not be cut off — a compared field is Ok(false),
    # a parsed field is Err, exactly what the reference does with the full Rust string
    n_plain_s, n_plain_e = len(storage), len(events)
    storage += [dict(storage[0], parent_state_root=storage[0]["parent_state_root"] + "\0junk"),   # 18 FALSE_STATE_ROOT
                dict(storage[0], child_block_cid=storage[0]["child_block_cid"] + "\0x"),          # 69 ERR_BAD_CLAIM
                dict(storage[0], value=storage[0]["value"] + "\0ff"),                             # 21 FALSE_VALUE
                dict(storage[0], slot=storage[0]["slot"] + "\0")]                                 # 69 ERR_BAD_CLAIM
    events += [dict(events[0], message_cid=events[0]["message_cid"] + "\0x"),                     # 69 ERR_BAD_CLAIM
               dict(events[0], data=events[0]["data"] + "\0"),                                    # 16 FALSE_DATA
               dict(events[0], topics=[events[0]["topics"][0] + "\0"] + events[0]["topics"][1:])]  # 15 FALSE_TOPIC
    text = bundle_ref.bundle_json(storage, events, blocks)
    assert "\\u0000" in text
    b = engine.bundle(text.encode())
    assert (b.n_blocks, b.n_events, b.n_storage) == (len(blocks), len(events), len(storage))
    st, bad = b.witness.verify_cids()
    assert bad == 0
    ss, es = b.verify()
    parsed = bundle_ref.parse_bundle(text)
    ev, sg = bundle_ref.claims_from_parsed(parsed)
    pst = oracle.store(*bundle_ref.tables_from_blocks(parsed["blocks"]))
    want_e = pst.verify_event_proofs(ev, mode=0)
    want_s = pst.verify_storage_proofs(sg, mode=0)
    pst.close()
    assert np.array_equal(es, want_e), (es.tolist(), want_e.tolist())
    assert np.array_equal(ss, want_s), (ss.tolist(), want_s.tolist())
    assert ss[n_plain_s:].tolist() == [18, 69, 21, 69] and es[n_plain_e:].tolist() == [69, 16, 15]
    assert (es[: len(gm)] == 1).all() and (ss[: len(sidx)] == 1).all()
    assert (es[len(gm):] != 1).all() and (ss[len(sidx):] != 1).all()
    # with the event filter of the generator
    import claims
    filt = claims.make_filter(tip.topic0, tip.topic1)
    _, es2 = b.verify(filt=filt)
    assert np.array_equal(es2[: len(gm)], es[: len(gm)])
    b.close()
