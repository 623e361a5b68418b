"""Malformed / unusual bundle JSON variants shared by the CPU (reference-in-Python) and GPU (engine) tests.
Each case is (name, text).  Whether a case parses is decided by tests/bundle_ref.py (the restated serde
rules); the engine must agree case by case."""
import base64
import json

import bundle_ref

CID_EMPTY_ARRAY = bytes.fromhex("0171a0e40220") + bytes(range(32))


def base_parts():
    storage = [dict(child_epoch=11, child_block_cid="bafyc", parent_state_root="bafyp", actor_id=1001,
                    actor_state_cid="bafya", storage_root="bafys", slot="0x" + "00" * 32, value="0x" + "11" * 32)]
    events = [dict(parent_epoch=10, child_epoch=11, parent_tipset_cids=["bafy1", "bafy2"], child_block_cid="bafyc",
                   message_cid="bafym", exec_index=3, event_index=1, emitter=1001, topics=["0x" + "aa" * 32],
                   data="0x1234")]
    blocks = [(CID_EMPTY_ARRAY, bytes(range(k)) * 3) for k in range(0, 9)]
    return storage, events, blocks


def cases():
    st, ev, bl = base_parts()
    good = bundle_ref.bundle_json(st, ev, bl)
    sj, ej = bundle_ref.storage_json(st[0]), bundle_ref.event_json(ev[0])
    bj = bundle_ref.block_json(*bl[3])
    cid_arr = "[" + ",".join(str(x) for x in CID_EMPTY_ARRAY) + "]"

    def top(s="[]", e="[]", b="[]", extra=""):
        return '{"storage_proofs":%s,"event_proofs":%s,"blocks":%s%s}' % (s, e, b, extra)

    def blk(cid=cid_arr, data='"QUJD"', extra=""):
        return top(b='[{"cid":%s,"data":%s%s}]' % (cid, data, extra))

    out = [("good", good), ("good pretty", json.dumps(json.loads(good), indent=2)),
           ("empty lists", top()), ("empty object", "{}"), ("array top", "[]"), ("null top", "null"), ("empty text", ""),
           ("whitespace", " \n\t" + top() + "\r\n "), ("trailing garbage", top() + "x"), ("two values", top() + top()),
           ("trailing comma", '{"storage_proofs":[],"event_proofs":[],"blocks":[],}'),
           ("unknown top field", top(extra=',"version":{"a":[1,2,{"b":null}],"c":true,"d":-1.5e3}')),
           ("unknown field first", '{"zzz":"\\u00e9\\ud83d\\ude00","storage_proofs":[],"event_proofs":[],"blocks":[]}'),
           ("lone surrogate ignored", '{"zzz":"\\ud800","storage_proofs":[],"event_proofs":[],"blocks":[]}'),
           ("lone low surrogate", '{"zzz":"\\udc00x","storage_proofs":[],"event_proofs":[],"blocks":[]}'),
           ("bad escape ignored", '{"zzz":"\\q","storage_proofs":[],"event_proofs":[],"blocks":[]}'),
           ("control char", '{"zzz":"a\x01b","storage_proofs":[],"event_proofs":[],"blocks":[]}'),
           ("NaN ignored", top(extra=',"x":NaN')), ("leading zero ignored", top(extra=',"x":01')),
           ("plus number ignored", top(extra=',"x":+1')), ("bare dot ignored", top(extra=',"x":1.')),
           ("missing blocks", '{"storage_proofs":[],"event_proofs":[]}'),
           ("duplicate blocks", '{"storage_proofs":[],"event_proofs":[],"blocks":[],"blocks":[]}'),
           ("blocks null", top(b="null")), ("blocks object", top(b="{}")), ("storage string", top(s='"x"')),
           ("deep ignored 126", top(extra=',"x":' + "[" * 126 + "]" * 126)),
           ("deep ignored 127", top(extra=',"x":' + "[" * 127 + "]" * 127)),
           ("deep ignored 128", top(extra=',"x":' + "[" * 128 + "]" * 128)),
           ("deep ignored 300", top(extra=',"x":' + "[" * 300 + "]" * 300)),
           ("deep ignored in block 124", blk(extra=',"x":' + "[" * 124 + "]" * 124)),
           ("deep ignored in block 125", blk(extra=',"x":' + "[" * 125 + "]" * 125)),
           ("storage ok", top(s="[" + sj + "]")), ("event ok", top(e="[" + ej + "]")),
           ("storage twice", top(s="[" + sj + "," + sj + "]"))]
    # field-level mutations of a StorageProof / EventProof
    s0 = json.loads(sj)
    e0 = json.loads(ej)

    def smod(**kw):
        d = dict(s0)
        for k, v in kw.items():
            if v is ...:
                d.pop(k)
            else:
                d[k] = v
        return top(s="[" + json.dumps(d) + "]")

    def emod(inner=None, **kw):
        d = json.loads(ej)
        for k, v in kw.items():
            if v is ...:
                d.pop(k)
            else:
                d[k] = v
        for k, v in (inner or {}).items():
            if v is ...:
                d["event_data"].pop(k)
            else:
                d["event_data"][k] = v
        return top(e="[" + json.dumps(d) + "]")

    out += [("storage missing slot", smod(slot=...)), ("storage actor float", smod(actor_id=1.0)),
            ("storage actor negative", smod(actor_id=-1)), ("storage actor string", smod(actor_id="1")),
            ("storage actor u64 max", smod(actor_id=(1 << 64) - 1)), ("storage actor 2^64", smod(actor_id=1 << 64)),
            ("storage epoch negative", smod(child_epoch=-5)), ("storage epoch i64 min", smod(child_epoch=-(1 << 63))),
            ("storage epoch below i64", smod(child_epoch=-(1 << 63) - 1)), ("storage epoch 2^63", smod(child_epoch=1 << 63)),
            ("storage epoch bool", smod(child_epoch=True)), ("storage epoch null", smod(child_epoch=None)),
            ("storage value null", smod(value=None)), ("storage value number", smod(value=7)),
            ("storage extra field", smod(extra={"k": [1, 2]})),
            ("storage epoch -0", top(s="[" + sj.replace('"child_epoch":11', '"child_epoch":-0') + "]")),
            ("storage epoch 1e1", top(s="[" + sj.replace('"child_epoch":11', '"child_epoch":1e1') + "]")),
            ("storage epoch 11.0", top(s="[" + sj.replace('"child_epoch":11', '"child_epoch":11.0') + "]")),
            ("storage dup field", top(s="[" + sj[:-1] + ',"slot":"0x00"}]')),
            ("storage escaped strings", top(s="[" + sj.replace("bafyc", "\\u0062afy\\/c\\n") + "]")),
            ("event missing data", emod(inner={"data": ...})), ("event missing event_data", emod(event_data=...)),
            ("event topics null", emod(inner={"topics": None})), ("event topics of numbers", emod(inner={"topics": [1]})),
            ("event topics empty", emod(inner={"topics": []})), ("event parents string", emod(parent_tipset_cids="x")),
            ("event parents empty", emod(parent_tipset_cids=[])), ("event exec negative", emod(exec_index=-1)),
            ("event emitter float", emod(inner={"emitter": 2.5})), ("event_data array", emod(event_data=[1, [], "x"])),
            ("event extra nested", emod(inner={"more": {"a": {"b": [1, {"c": 2}]}}}))]
    # blocks
    out += [("block ok", blk()), ("block empty data", blk(data='""')), ("block 1 pad", blk(data='"QUI="')),
            ("block 2 pads", blk(data='"QQ=="')), ("block bad len 1", blk(data='"Q"')), ("block bad len 2", blk(data='"QQ"')),
            ("block bad len 3", blk(data='"QUI"')), ("block only pads", blk(data='"===="')),
            ("block 3 pads", blk(data='"Q==="')), ("block pad in middle", blk(data='"QQ==QUJD"')),
            ("block pad then char", blk(data='"QQ=A"')), ("block nonzero tail 2", blk(data='"QR=="')),
            ("block nonzero tail 1", blk(data='"QUJ="')), ("block url alphabet", blk(data='"QU-_"')),
            ("block space", blk(data='"QU JD"')), ("block newline escape", blk(data='"QUJD\\n"')),
            ("block escaped ok", blk(data='"QU\\u004aD"')), ("block escaped slash", blk(data='"\\/\\/\\/\\/"')),
            ("block plus slash", blk(data='"+/+/"')), ("block data number", blk(data="5")), ("block data null", blk(data="null")),
            ("block missing data", top(b='[{"cid":%s}]' % cid_arr)), ("block missing cid", top(b='[{"data":"QUJD"}]')),
            ("block dup data", blk(extra=',"data":"QUJD"')), ("block extra", blk(extra=',"codec":113')),
            ("block cid string", blk(cid='"bafy2bzaceaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa"')),
            ("block cid empty", blk(cid="[]")), ("block cid 256", blk(cid=cid_arr.replace("[1,", "[256,"))),
            ("block cid negative", blk(cid=cid_arr.replace("[1,", "[-1,"))), ("block cid float", blk(cid=cid_arr.replace("[1,", "[1.0,"))),
            ("block cid non-minimal varint", blk(cid=cid_arr.replace("[1,113,", "[1,241,0,"))),
            ("block cid short", blk(cid=cid_arr[:-4] + "]")), ("block cid long", blk(cid=cid_arr[:-1] + ",7]")),
            ("block cid v0", blk(cid="[18,32," + ",".join(["9"] * 32) + "]")), ("block cid v2", blk(cid=cid_arr.replace("[1,", "[2,"))),
            ("block cid sha256", blk(cid="[1,113,18,32," + ",".join(["9"] * 32) + "]")),
            ("block cid identity empty", blk(cid="[1,85,0,0]")), ("block cid object", blk(cid='{"/":"bafy"}')),
            ("block cid nested", blk(cid="[[1]]")), ("block cid bool", blk(cid="[true]")),
            ("block long", blk(data='"' + base64.b64encode(bytes(range(256)) * 40).decode() + '"')),
            ("block trailing comma", top(b="[" + bj + ",]")), ("block array elem number", top(b="[1]"))]
    return out
