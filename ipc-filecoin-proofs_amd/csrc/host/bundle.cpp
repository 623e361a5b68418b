// csrc/host/bundle.cpp — the bundle wire format: `UnifiedProofBundle` as
// serde_json text → HBM-resident witness + claim arrays → `verify_proof_bundle`.
//
//   UnifiedProofBundle { storage_proofs, event_proofs, blocks }   src/proofs/common/bundle.rs:39-45
//   ProofBlock { cid: Cid, data: base64 }                          src/proofs/common/bundle.rs:10-37
//   EventProof / EventData                                         src/proofs/events/bundle.rs:5-22
//   StorageProof                                                   src/proofs/storage/bundle.rs:4-14
//   verify_proof_bundle                                            src/proofs/verifier.rs:12-62
//
// The host tokenises the JSON once (claim fields are copied out; each `data` string is only located);
// the text goes to HBM as is and ONE kernel decodes every block's base64 straight into the line-aligned
// arena (kernels/base64.hip).  The claim strings go through the same lowering as
// ipcfp_verify_event_proofs / ipcfp_verify_storage_proofs.
//
// `ProofBlock.cid` ⚠: with cid 0.11's serde feature a `Cid` serialises as a newtype struct around its
// bytes, which serde_json writes as an array of numbers; that is the form read here.  With
// IPCFP_BUNDLE_CID_STRINGS a multibase string is accepted as well (an extension, not reference behaviour).
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../common.h"
#include "../kernels/launch.h"
#include "cidstr.h"
#include "json_min.h"

using namespace ipcfp;

struct ipcfp_bundle {
    ipcfp_ctx* ctx = nullptr;
    ipcfp_witness_t* witness = nullptr;
    // owned claim strings; the public structs point into these (stable: all reserved before use)
    std::vector<std::string> strings;
    std::vector<std::vector<const char*>> string_arrays;
    std::vector<ipcfp_event_proof_t> events;
    std::vector<ipcfp_storage_proof_t> storage;
    uint64_t n_blocks = 0, payload_bytes = 0;
};

namespace {

struct Span {  // B64Span of kernels/base64.hip
    uint64_t src;
    uint32_t len;
    uint32_t unit0;
};

struct RawEvent {
    int64_t parent_epoch = 0, child_epoch = 0;
    std::vector<std::string> parents;
    std::string child, message;
    uint64_t exec_index = 0, event_index = 0, emitter = 0;
    std::vector<std::string> topics;
    std::string data;
};
struct RawStorage {
    int64_t child_epoch = 0;
    uint64_t actor_id = 0;
    std::string child, parent_state_root, actor_state, storage_root, slot, value;
};

// One `key: value` loop of a derived struct: `field(name)` returns false for unknown names.
template <class F>
bool parse_object(JsonCursor& js, int depth, F&& field) {
    if (!js.expect('{', "expected '{'")) return false;
    if (js.consume('}')) return true;
    for (;;) {
        std::string key;
        if (!js.string(key)) return false;
        if (!js.expect(':', "expected ':'")) return false;
        const int r = field(key);  // 1 handled, 0 unknown, -1 error
        if (r < 0) return false;
        if (r == 0 && !js.skip_value(depth)) return false;
        if (js.consume(',')) continue;
        return js.expect('}', "expected ',' or '}'");
    }
}
template <class F>
bool parse_array(JsonCursor& js, F&& element) {
    if (!js.expect('[', "expected '['")) return false;
    if (js.consume(']')) return true;
    for (;;) {
        if (!element()) return false;
        if (js.consume(',')) continue;
        return js.expect(']', "expected ',' or ']'");
    }
}
// `seen` bit bookkeeping for duplicate / missing fields
struct Fields {
    uint32_t seen = 0;
    JsonCursor& js;
    explicit Fields(JsonCursor& j) : js(j) {}
    bool first(int bit, const char* name) {
        if (seen & (1u << bit)) {
            js.fail((std::string("duplicate field `") + name + "`").c_str());
            return false;
        }
        seen |= 1u << bit;
        return true;
    }
    bool complete(uint32_t want, const char* what) {
        if ((seen & want) == want) return true;
        js.fail((std::string("missing field in ") + what).c_str());
        return false;
    }
};
bool string_array(JsonCursor& js, std::vector<std::string>& out) {
    return parse_array(js, [&] {
        out.emplace_back();
        return js.string(out.back());
    });
}

bool parse_event_data(JsonCursor& js, RawEvent& e) {
    Fields f(js);
    const bool ok = parse_object(js, 4, [&](const std::string& k) -> int {
        if (k == "emitter") return f.first(0, "emitter") && js.u64(e.emitter) ? 1 : -1;
        if (k == "topics") return f.first(1, "topics") && string_array(js, e.topics) ? 1 : -1;
        if (k == "data") return f.first(2, "data") && js.string(e.data) ? 1 : -1;
        return 0;
    });
    return ok && f.complete(7, "EventData");
}
bool parse_event(JsonCursor& js, RawEvent& e) {
    Fields f(js);
    const bool ok = parse_object(js, 3, [&](const std::string& k) -> int {
        if (k == "parent_epoch") return f.first(0, "parent_epoch") && js.i64(e.parent_epoch) ? 1 : -1;
        if (k == "child_epoch") return f.first(1, "child_epoch") && js.i64(e.child_epoch) ? 1 : -1;
        if (k == "parent_tipset_cids") return f.first(2, "parent_tipset_cids") && string_array(js, e.parents) ? 1 : -1;
        if (k == "child_block_cid") return f.first(3, "child_block_cid") && js.string(e.child) ? 1 : -1;
        if (k == "message_cid") return f.first(4, "message_cid") && js.string(e.message) ? 1 : -1;
        if (k == "exec_index") return f.first(5, "exec_index") && js.u64(e.exec_index) ? 1 : -1;
        if (k == "event_index") return f.first(6, "event_index") && js.u64(e.event_index) ? 1 : -1;
        if (k == "event_data") return f.first(7, "event_data") && parse_event_data(js, e) ? 1 : -1;
        return 0;
    });
    return ok && f.complete(0xff, "EventProof");
}
bool parse_storage(JsonCursor& js, RawStorage& s) {
    Fields f(js);
    const bool ok = parse_object(js, 3, [&](const std::string& k) -> int {
        if (k == "child_epoch") return f.first(0, "child_epoch") && js.i64(s.child_epoch) ? 1 : -1;
        if (k == "child_block_cid") return f.first(1, "child_block_cid") && js.string(s.child) ? 1 : -1;
        if (k == "parent_state_root") return f.first(2, "parent_state_root") && js.string(s.parent_state_root) ? 1 : -1;
        if (k == "actor_id") return f.first(3, "actor_id") && js.u64(s.actor_id) ? 1 : -1;
        if (k == "actor_state_cid") return f.first(4, "actor_state_cid") && js.string(s.actor_state) ? 1 : -1;
        if (k == "storage_root") return f.first(5, "storage_root") && js.string(s.storage_root) ? 1 : -1;
        if (k == "slot") return f.first(6, "slot") && js.string(s.slot) ? 1 : -1;
        if (k == "value") return f.first(7, "value") && js.string(s.value) ? 1 : -1;
        return 0;
    });
    return ok && f.complete(0xff, "StorageProof");
}

struct CidSpan {  // CidSpan of kernels/base64.hip: the body of the `[…]` byte array
    uint64_t src;
    uint32_t len;
    uint32_t pad;
};

struct RawBlocks {
    std::vector<CidSpan> cid_spans;
    std::vector<Span> spans;
    std::vector<uint64_t> dst_off;
    std::vector<uint32_t> len;
    uint64_t arena_bytes = 0, units = 0, payload = 0;
    std::string extra;  // unescaped data strings, addressed as if appended to the JSON text
};

// Cid as serde_json sees it: [b0, b1, …] (u8 each).  The host only LOCATES the array body — it cannot
// contain a ']' unless it is malformed — and the device parses and validates the numbers
// (k_parse_cid_arrays).  Optionally a multibase string, rewritten to the array form behind the text.
bool locate_cid(JsonCursor& js, uint32_t flags, RawBlocks& b, CidSpan& sp) {
    if (js.peek('"')) {
        std::string s;
        if (!js.string(s)) return false;
        if (!(flags & IPCFP_BUNDLE_CID_STRINGS)) return js.fail("invalid type: string, expected a CID as bytes");
        std::vector<uint8_t> bin;
        if (!cid_from_string(s.c_str(), bin)) return js.fail("invalid CID string");
        std::string body;
        for (size_t i = 0; i < bin.size(); ++i) body += (i ? "," : "") + std::to_string(unsigned(bin[i]));
        sp.src = uint64_t(js.end - js.begin) + b.extra.size();
        sp.len = uint32_t(body.size());
        b.extra += body;
        return true;
    }
    if (!js.expect('[', "invalid type: expected a CID as a byte sequence")) return false;
    const char* close = static_cast<const char*>(std::memchr(js.p, ']', size_t(js.end - js.p)));
    if (!close) return js.fail("unterminated array");
    if (size_t(close - js.p) > 4096) return js.fail("invalid CID bytes");  // 104 bytes need < 1 KiB of text
    sp.src = uint64_t(js.p - js.begin);
    sp.len = uint32_t(close - js.p);
    js.p = close + 1;
    return true;
}

bool parse_block(JsonCursor& js, uint32_t flags, RawBlocks& b) {
    Fields f(js);
    CidSpan cs{0, 0, 0};
    const char* ds = nullptr;
    size_t dn = 0;
    bool plain = true;
    std::string unescaped;
    const bool ok = parse_object(js, 3, [&](const std::string& k) -> int {
        if (k == "cid") return f.first(0, "cid") && locate_cid(js, flags, b, cs) ? 1 : -1;
        if (k == "data") {
            if (!f.first(1, "data")) return -1;
            if (!js.string_span(ds, dn, plain)) return -1;
            // an escaped string (`\u0041`, `\/`) is legal JSON: unescape that rare case on the host; the
            // text goes to the device behind the JSON (RawBlocks::extra)
            if (!plain) {
                if (!js.unescape(ds, dn, unescaped)) return -1;
                ds = unescaped.data();
                dn = unescaped.size();
            }
            return 1;
        }
        return 0;
    });
    if (!ok || !f.complete(3, "ProofBlock")) return false;
    // base64 0.21 STANDARD: canonical padding ⇒ a multiple of 4 characters
    if (dn & 3u) return js.fail("base64: invalid length / padding");
    if (dn >= 0xfffffff0ull) return js.fail("block too large");
    uint32_t pads = 0;
    if (dn && ds[dn - 1] == '=') pads = (ds[dn - 2] == '=') ? 2 : 1;
    const uint32_t dec = uint32_t(dn / 4 * 3) - pads;
    b.cid_spans.push_back(cs);
    Span sp;
    if (plain) sp.src = uint64_t(ds - js.begin);
    else {
        sp.src = uint64_t(js.end - js.begin) + b.extra.size();
        b.extra.append(ds, dn);
    }
    sp.len = uint32_t(dn);
    if (b.units >= 0xfffffff0ull) return js.fail("bundle too large");
    sp.unit0 = uint32_t(b.units);
    b.units += (dn + 15) / 16;
    b.spans.push_back(sp);
    b.dst_off.push_back(b.arena_bytes);
    b.len.push_back(dec);
    b.arena_bytes += dec == 0 ? 128 : (uint64_t(dec) + 127) & ~127ull;
    b.payload += dec;
    return true;
}

// Claim strings cross the C ABI NUL-terminated, but a JSON string may hold "\u0000".  Cutting the string there
// would let `"bafy…\u0000junk"` verify as if the suffix were absent, where the reference sees a string that
// neither parses (`Cid::try_from`, hex::decode) nor compares equal (`==`, eq_ignore_ascii_case).  Every NUL is
// replaced by 0x01: a byte that, like NUL, is in no multibase / hex alphabet and equals no character of a
// canonical form — so each field keeps the reference's outcome (Err for a parsed field, Ok(false) for a compared one).
const char* keep(ipcfp_bundle& b, std::string&& s) {
    for (char& c : s)
        if (c == '\0') c = '\x01';
    b.strings.push_back(std::move(s));
    return b.strings.back().c_str();
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// The host half of the parse: tokenise, copy the claim fields out, locate every block's `cid` array and `data`
// string.  What is left to the device is the CONTENT of those two (base64 alphabet / padding, CID bytes).
bool tokenise_bundle(JsonCursor& js, uint32_t flags, std::vector<RawEvent>& ev, std::vector<RawStorage>& stg,
                     RawBlocks& blocks) {
    Fields f(js);
    bool ok = parse_object(js, 1, [&](const std::string& k) -> int {
        if (k == "storage_proofs")
            return f.first(0, "storage_proofs") && parse_array(js, [&] {
                       stg.emplace_back();
                       return parse_storage(js, stg.back());
                   }) ? 1 : -1;
        if (k == "event_proofs")
            return f.first(1, "event_proofs") && parse_array(js, [&] {
                       ev.emplace_back();
                       return parse_event(js, ev.back());
                   }) ? 1 : -1;
        if (k == "blocks")
            return f.first(2, "blocks") && parse_array(js, [&] { return parse_block(js, flags, blocks); }) ? 1 : -1;
        return 0;
    });
    ok = ok && f.complete(7, "UnifiedProofBundle");
    if (ok && !js.at_end()) ok = js.fail("trailing characters");
    return ok;
}
}  // namespace

extern "C" {

// Host half only — no context, no device: IPCFP_OK iff the text tokenises as a UnifiedProofBundle (structure,
// field types, string escapes, base64 LENGTHS); the contents of `cid` arrays and `data` strings are what the
// device checks in ipcfp_bundle_parse_json.  Exists so that the tokeniser can be tested without a GPU.
int ipcfp_bundle_check_json(const char* json, uint64_t len, uint32_t flags, uint64_t* n_storage, uint64_t* n_events,
                            uint64_t* n_blocks, char* err_out, uint32_t err_cap) {
    if (len && !json) return IPCFP_E_INVALID;
    JsonCursor js(json, size_t(len));
    std::vector<RawEvent> ev;
    std::vector<RawStorage> stg;
    RawBlocks blocks;
    const bool ok = tokenise_bundle(js, flags, ev, stg, blocks);
    if (n_storage) *n_storage = stg.size();
    if (n_events) *n_events = ev.size();
    if (n_blocks) *n_blocks = blocks.spans.size();
    if (err_out && err_cap) {
        std::strncpy(err_out, ok ? "" : js.err.c_str(), err_cap - 1);
        err_out[err_cap - 1] = 0;
    }
    return ok ? IPCFP_OK : IPCFP_E_PARSE;
}

int ipcfp_bundle_parse_json(ipcfp_ctx_t* ctx, const char* json, uint64_t len, uint32_t flags, ipcfp_bundle_t** out) {
    if (!ctx || !out || (len && !json)) return IPCFP_E_INVALID;
    *out = nullptr;
    IPCFP_ENTER(ctx);
    JsonCursor js(json, size_t(len));
    std::vector<RawEvent> ev;
    std::vector<RawStorage> stg;
    RawBlocks blocks;
    bool ok = tokenise_bundle(js, flags, ev, stg, blocks);
    if (!ok)
        return set_error(ctx, IPCFP_E_PARSE, "bundle JSON: %s", js.err.c_str());
    const uint64_t n = blocks.spans.size();
    if (n >= 0xffffffffull) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 blocks");

    std::unique_ptr<ipcfp_bundle> b(new (std::nothrow) ipcfp_bundle());
    if (!b) return IPCFP_E_NOMEM;
    b->ctx = ctx;
    b->n_blocks = n;
    b->payload_bytes = blocks.payload;

    // ---- blocks: text → HBM, one decode kernel, then the ordinary device witness path ----
    {
        DevBuf<uint8_t> text_d, arena_d, cids_d;
        DevBuf<Span> spans_d;
        DevBuf<CidSpan> cspans_d;
        DevBuf<uint64_t> off_d;
        DevBuf<uint32_t> len_d;
        DevBuf<unsigned long long> bad_d;
        const uint64_t xlen = blocks.extra.size();
        IPCFP_HIP(ctx, text_d.alloc(len + xlen + 32));
        IPCFP_HIP(ctx, arena_d.alloc(blocks.arena_bytes + 256));
        IPCFP_HIP(ctx, cids_d.alloc(n * IPCFP_CID_SLOT));
        IPCFP_HIP(ctx, cspans_d.alloc(n));
        IPCFP_HIP(ctx, spans_d.alloc(n));
        IPCFP_HIP(ctx, off_d.alloc(n));
        IPCFP_HIP(ctx, len_d.alloc(n));
        IPCFP_HIP(ctx, bad_d.alloc(2));
        const unsigned long long none = ~0ULL, none2[2] = {none, none};
        IPCFP_HIP(ctx, hipMemcpyAsync(bad_d.p, none2, 16, hipMemcpyHostToDevice, ctx->stream));
        if (n) {
            IPCFP_HIP(ctx, hipMemcpyAsync(text_d.p, json, len, hipMemcpyHostToDevice, ctx->stream));
            if (xlen)
                IPCFP_HIP(ctx, hipMemcpyAsync(text_d.p + len, blocks.extra.data(), xlen, hipMemcpyHostToDevice, ctx->stream));
            IPCFP_HIP(ctx, hipMemsetAsync(text_d.p + len + xlen, 0, 32, ctx->stream));
            IPCFP_HIP(ctx, hipMemsetAsync(arena_d.p, 0, blocks.arena_bytes + 256, ctx->stream));
            IPCFP_HIP(ctx, hipMemcpyAsync(cspans_d.p, blocks.cid_spans.data(), n * sizeof(CidSpan), hipMemcpyHostToDevice, ctx->stream));
            IPCFP_HIP(ctx, hipMemcpyAsync(spans_d.p, blocks.spans.data(), n * sizeof(Span), hipMemcpyHostToDevice, ctx->stream));
            IPCFP_HIP(ctx, hipMemcpyAsync(off_d.p, blocks.dst_off.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
            IPCFP_HIP(ctx, hipMemcpyAsync(len_d.p, blocks.len.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
            int rc = launch_base64_decode(ctx, text_d.p, spans_d.p, uint32_t(n), uint32_t(blocks.units), off_d.p, arena_d.p,
                                          bad_d.p);
            if (rc) return rc;
            rc = launch_parse_cid_arrays(ctx, text_d.p, cspans_d.p, uint32_t(n), cids_d.p, bad_d.p + 1);
            if (rc) return rc;
        }
        unsigned long long bad[2] = {none, none};
        IPCFP_HIP(ctx, d2h_small(ctx, bad, bad_d.p, 16, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        // serde reads `cid` and `data` in text order, block by block: report the earlier block
        const unsigned long long b64_blk = bad[0], cid_blk = bad[1] == none ? none : bad[1] >> 2;
        if (cid_blk != none && cid_blk <= b64_blk) {
            return set_error(ctx, IPCFP_E_PARSE, "bundle JSON: blocks[%llu].cid is not a CID byte array", cid_blk);
        }
        if (b64_blk != none)
            return set_error(ctx, IPCFP_E_PARSE, "bundle JSON: blocks[%llu].data is not valid base64", b64_blk);
        int rc = ipcfp_witness_create_device(ctx, arena_d.p, blocks.arena_bytes, off_d.p, len_d.p, cids_d.p, n, &b->witness);
        if (rc) return rc;
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    }

    // ---- claims: the reference's structs, strings owned by the bundle ----
    size_t n_str = 0;
    for (const RawEvent& e : ev) n_str += 3 + e.parents.size() + e.topics.size();
    n_str += stg.size() * 6;
    b->strings.reserve(n_str);
    b->string_arrays.reserve(ev.size() * 2);
    b->events.resize(ev.size());
    for (size_t i = 0; i < ev.size(); ++i) {
        RawEvent& e = ev[i];
        ipcfp_event_proof_t& p = b->events[i];
        std::memset(&p, 0, sizeof p);
        p.parent_epoch = e.parent_epoch;
        p.child_epoch = e.child_epoch;
        b->string_arrays.emplace_back();
        for (auto& s : e.parents) b->string_arrays.back().push_back(keep(*b, std::move(s)));
        p.parent_tipset_cids = b->string_arrays.back().data();
        p.n_parent_tipset_cids = uint32_t(e.parents.size());
        p.child_block_cid = keep(*b, std::move(e.child));
        p.message_cid = keep(*b, std::move(e.message));
        p.exec_index = e.exec_index;
        p.event_index = e.event_index;
        p.emitter = e.emitter;
        b->string_arrays.emplace_back();
        for (auto& s : e.topics) b->string_arrays.back().push_back(keep(*b, std::move(s)));
        p.topics = b->string_arrays.back().data();
        p.n_topics = uint32_t(e.topics.size());
        p.data = keep(*b, std::move(e.data));
    }
    b->storage.resize(stg.size());
    for (size_t i = 0; i < stg.size(); ++i) {
        RawStorage& s = stg[i];
        ipcfp_storage_proof_t& p = b->storage[i];
        std::memset(&p, 0, sizeof p);
        p.child_epoch = s.child_epoch;
        p.actor_id = s.actor_id;
        p.child_block_cid = keep(*b, std::move(s.child));
        p.parent_state_root = keep(*b, std::move(s.parent_state_root));
        p.actor_state_cid = keep(*b, std::move(s.actor_state));
        p.storage_root = keep(*b, std::move(s.storage_root));
        p.slot = keep(*b, std::move(s.slot));
        p.value = keep(*b, std::move(s.value));
    }
    *out = b.release();
    return IPCFP_OK;
}

void ipcfp_bundle_destroy(ipcfp_bundle_t* b) {
    if (!b) return;
    ipcfp_witness_destroy(b->witness);
    delete b;
}

ipcfp_witness_t* ipcfp_bundle_witness(ipcfp_bundle_t* b) { return b ? b->witness : nullptr; }
uint64_t ipcfp_bundle_block_count(const ipcfp_bundle_t* b) { return b ? b->n_blocks : 0; }
uint64_t ipcfp_bundle_event_count(const ipcfp_bundle_t* b) { return b ? b->events.size() : 0; }
uint64_t ipcfp_bundle_storage_count(const ipcfp_bundle_t* b) { return b ? b->storage.size() : 0; }
const ipcfp_event_proof_t* ipcfp_bundle_event_proofs(const ipcfp_bundle_t* b) { return b ? b->events.data() : nullptr; }
const ipcfp_storage_proof_t* ipcfp_bundle_storage_proofs(const ipcfp_bundle_t* b) {
    return b ? b->storage.data() : nullptr;
}

int ipcfp_verify_proof_bundle(ipcfp_ctx_t* ctx, ipcfp_bundle_t* b, const ipcfp_trust_policy_t* trust,
                              const ipcfp_event_filter_t* filter, ipcfp_status_t* storage_status,
                              ipcfp_status_t* event_status) {
    if (!ctx || !b || b->ctx != ctx) return IPCFP_E_INVALID;
    if ((!b->storage.empty() && !storage_status) || (!b->events.empty() && !event_status)) return IPCFP_E_INVALID;
    int rc = ipcfp_verify_storage_proofs(ctx, b->witness, b->storage.data(), b->storage.size(), trust, storage_status);
    if (rc) return rc;
    return ipcfp_verify_event_proofs(ctx, b->witness, b->events.data(), b->events.size(), trust, filter, event_status);
}

}  // extern "C"
