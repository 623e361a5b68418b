// oracle/verify.cpp — TEST INFRASTRUCTURE.
//
// Line-for-line CPU restatement of the reference verifiers and of the two-pass event scan.
// Each function names the reference lines it follows (paths under /root/reference/).
// `Ok(false)` → a FALSE_* status; `Err` → throw orc::Err (status ERR_*).
#include "verify.hpp"

#include <cstdio>
#include <cstdlib>

#include <omp.h>

#include <algorithm>
#include <set>
#include <unordered_set>

#include "hashes.hpp"

namespace orc {

bool trusted(const ipcfp_trust_policy_t* t, int64_t epoch) {
    if (!t || t->kind == 0) return true;                    // AcceptAll (trust/mod.rs:55,69)
    if (t->ec_chain_empty) return false;                    // cert.rs:55-57
    return epoch >= t->min_epoch && epoch <= t->max_epoch;  // cert.rs:60-63
}

static Cid parse_claim_cid(const char* s) {
    Cid c;
    if (!s || !cid_from_string(s, c)) throw Err(IPCFP_ST_ERR_BAD_CLAIM, "Failed to parse CID");
    return c;
}

static std::vector<Cid> parse_claim_cids(const char* const* v, uint32_t n) {
    std::vector<Cid> out;
    for (uint32_t i = 0; i < n; ++i) out.push_back(parse_claim_cid(v[i]));
    return out;
}

// ---- execution order -------------------------------------------------------
// collect_exec_list (events/utils.rs:48-94), verify_txmeta = true
static std::vector<Cid> collect_exec_list(const Blockstore& bs, const std::vector<Cid>& txmeta_cids) {
    std::vector<Cid> out;
    std::unordered_set<Bytes, BytesHash> seen;  // HashSet<Cid> (events/utils.rs:54,57)
    for (const Cid& tx : txmeta_cids) {
        const Bytes& raw = must_get(bs, tx, "TxMeta");  // :58-60
        Reader r(raw);
        r.expect_array(2);  // (Cid, Cid)  :61
        Cid bls = read_cid(r), secp = read_cid(r);
        r.finish();
        // put_cbor(&(bls,secp), Blake2b256): re-encode the 2-tuple canonically and hash (:65-72)
        Bytes enc;
        enc.push_back(0x82);
        for (const Cid* c : {&bls, &secp}) {
            enc.push_back(0xd8); enc.push_back(0x2a);
            const size_t l = c->b.size() + 1;
            if (l < 24) enc.push_back(uint8_t(0x40 | l));
            else { enc.push_back(0x58); enc.push_back(uint8_t(l)); }
            enc.push_back(0x00);
            enc.insert(enc.end(), c->b.begin(), c->b.end());
        }
        if (cid_for_block(enc.data(), enc.size()) != tx) throw Err(IPCFP_ST_ERR_TXMETA_MISMATCH, "TxMeta mismatch");
        for (const Cid* root : {&bls, &secp}) {  // :76-90
            AmtRoot a = amt_load(bs, *root, 0, check_cid_value);
            amt_for_each(bs, a, check_cid_value, [&](uint64_t, const ValueLoc& v) {
                Reader vr(v.block->data() + v.off, v.len);
                Cid c = read_cid(vr);
                if (seen.insert(c.b).second) out.push_back(c);
            });
        }
    }
    return out;
}

std::vector<Cid> reconstruct_execution_order(const Blockstore& bs, const std::vector<Cid>& parents) {
    std::vector<Cid> tx;
    for (const Cid& p : parents) {  // events/utils.rs:20-27
        HeaderLite h = decode_header(must_get(bs, p, "parent header"));
        tx.push_back(h.messages);
    }
    return collect_exec_list(bs, tx);
}

// ---- hex compares ------------------------------------------------------------
static std::string hex0x(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s = "0x";
    for (size_t i = 0; i < n; ++i) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
    return s;
}
static bool eq_ignore_ascii_case(const std::string& a, const char* b) {
    if (!b) return false;
    const size_t n = std::strlen(b);
    if (a.size() != n) return false;
    for (size_t i = 0; i < n; ++i) {
        char x = a[i], y = b[i];
        if (x >= 'A' && x <= 'Z') x = char(x + 32);
        if (y >= 'A' && y <= 'Z') y = char(y + 32);
        if (x != y) return false;
    }
    return true;
}

// ---- event proof ---------------------------------------------------------------
// TxMeta of one parent block → its (bls, secp) AMT roots, with the re-hash check (events/utils.rs:58-72)
static void txmeta_roots(const Blockstore& bs, const Cid& tx, Cid& bls, Cid& secp) {
    const Bytes& raw = must_get(bs, tx, "TxMeta");
    Reader r(raw);
    r.expect_array(2);
    bls = read_cid(r);
    secp = read_cid(r);
    r.finish();
    Bytes enc;
    enc.push_back(0x82);
    for (const Cid* c : {&bls, &secp}) {
        enc.push_back(0xd8); enc.push_back(0x2a);
        const size_t l = c->b.size() + 1;
        if (l < 24) enc.push_back(uint8_t(0x40 | l));
        else { enc.push_back(0x58); enc.push_back(uint8_t(l)); }
        enc.push_back(0x00);
        enc.insert(enc.end(), c->b.begin(), c->b.end());
    }
    if (cid_for_block(enc.data(), enc.size()) != tx) throw Err(IPCFP_ST_ERR_TXMETA_MISMATCH, "TxMeta mismatch");
}

ExecCache build_exec_cache(const Blockstore& bs, const std::vector<Cid>& parents, int threads) {
    ExecCache c;
    try {
        if (threads == 1) {
            std::vector<Cid> exec = reconstruct_execution_order(bs, parents);
            c.shards[0].reserve(exec.size() * 2);
            for (size_t i = 0; i < exec.size(); ++i) c.shards[0].emplace(exec[i].b, i);  // exec is already de-duplicated
            c.len = exec.size();
            c.ok = true;
            return c;
        }
        const int nt = use_threads(threads);
        // reconstruct_execution_order (utils.rs:20-27) + collect_exec_list (:56-91), the for_each walks in parallel
        std::vector<Cid> tx;
        for (const Cid& p : parents) tx.push_back(decode_header(must_get(bs, p, "parent header")).messages);
        std::vector<Bytes> raw;  // the concatenated for_each sequences, duplicates included
        for (const Cid& t : tx) {
            Cid roots[2];
            txmeta_roots(bs, t, roots[0], roots[1]);
            for (const Cid& root : roots) {
                AmtRoot a = amt_load(bs, root, 0, check_cid_value);
                std::vector<AmtItem> items;
                amt_collect(bs, a, check_cid_value, items, threads);
                const size_t at = raw.size();
                raw.resize(at + items.size());
#pragma omp parallel for schedule(static)
                for (int64_t k = 0; k < int64_t(items.size()); ++k) {
                    Reader vr(items[size_t(k)].value.block->data() + items[size_t(k)].value.off, items[size_t(k)].value.len);
                    raw[at + size_t(k)] = read_cid(vr).b;  // validated by the walk: cannot throw
                }
            }
        }
        // `if seen.insert(c) { out.push(c) }`: sub-map s owns the keys whose hash selects it, visited in raw order
        size_t ns = 1;
        while (ns < size_t(nt) * 4) ns <<= 1;
        c.shards.assign(ns, ExecCache::Map());
        c.mask = ns - 1;
        const size_t n = raw.size();
        std::vector<uint32_t> shard_of(n);
        std::vector<uint8_t> first(n, 0);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < int64_t(n); ++i) shard_of[size_t(i)] = uint32_t((BytesHash()(raw[size_t(i)]) >> 40) & c.mask);
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t sidx = 0; sidx < int64_t(ns); ++sidx) {
            ExecCache::Map& m = c.shards[size_t(sidx)];
            m.reserve(n / ns * 2 + 16);
            for (size_t i = 0; i < n; ++i)
                if (shard_of[i] == uint32_t(sidx) && m.emplace(raw[i], i).second) first[i] = 1;
        }
        std::vector<uint64_t> exec_index(n);
        uint64_t run = 0;
        for (size_t i = 0; i < n; ++i) {
            exec_index[i] = run;
            run += first[i];
        }
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t sidx = 0; sidx < int64_t(ns); ++sidx)
            for (auto& kv : c.shards[size_t(sidx)]) kv.second = exec_index[kv.second];
        c.len = run;
        c.ok = true;
    } catch (const Err& e) {
        c.err_status = e.status;
    }
    return c;
}

uint8_t verify_event_proof_one(const Blockstore& bs, const ipcfp_event_proof_t& p, const ipcfp_trust_policy_t* trust,
                               const ipcfp_event_filter_t* filter, const ExecCache* exec_cache) {
    // Step 1: verify_trust_anchors (events/verifier.rs:124-144)
    std::vector<Cid> parent_cids = parse_claim_cids(p.parent_tipset_cids, p.n_parent_tipset_cids);  // :130
    Cid child_cid = parse_claim_cid(p.child_block_cid);                                                // :131
    if (!trusted(trust, p.parent_epoch)) return IPCFP_ST_FALSE_UNTRUSTED_PARENT;                        // :134
    if (!trusted(trust, p.child_epoch)) return IPCFP_ST_FALSE_UNTRUSTED_CHILD;                          // :139
    // Step 2: verify_header_consistency (:147-181)
    HeaderLite child_hdr = decode_header(must_get(bs, child_cid, "child header"));                      // :155-158
    if (child_hdr.parents != parent_cids) return IPCFP_ST_FALSE_PARENTS_MISMATCH;                       // :161
    if (child_hdr.height != p.child_epoch) return IPCFP_ST_FALSE_CHILD_EPOCH;                           // :166
    if (parent_cids.empty()) throw Err(IPCFP_ST_ERR_EMPTY_PARENTS, "parent_cids[0] on an empty tipset key");  // :172 panics
    HeaderLite parent_hdr = decode_header(must_get(bs, parent_cids[0], "parent header"));               // :171-174
    if (parent_hdr.height != p.parent_epoch) return IPCFP_ST_FALSE_PARENT_EPOCH;                        // :176
    // Step 3: verify_execution_order (:184-204)
    if (exec_cache) {
        if (!exec_cache->ok) throw Err(exec_cache->err_status, "reconstruct_execution_order failed");   // :190
        Cid msg = parse_claim_cid(p.message_cid);                                                       // :193
        const uint64_t* hit = exec_cache->find(msg.b);
        if (!hit) return IPCFP_ST_FALSE_MSG_NOT_IN_EXEC;                                                // :194
        if (*hit != p.exec_index) return IPCFP_ST_FALSE_EXEC_INDEX;                                     // :199
    } else {
        std::vector<Cid> exec = reconstruct_execution_order(bs, parent_cids);                           // :190
        Cid msg = parse_claim_cid(p.message_cid);                                                       // :193
        auto it = std::find(exec.begin(), exec.end(), msg);
        if (it == exec.end()) return IPCFP_ST_FALSE_MSG_NOT_IN_EXEC;                                    // :194
        if (uint64_t(it - exec.begin()) != p.exec_index) return IPCFP_ST_FALSE_EXEC_INDEX;              // :199
    }
    // Step 4: verify_receipt_and_event (:207-254)
    HeaderLite ch = decode_header(must_get(bs, child_cid, "child header"));                             // :214-217
    AmtRoot receipts = amt_load(bs, ch.parent_message_receipts, 0, check_receipt);                      // :220
    ValueLoc rloc;
    if (!amt_get(bs, receipts, p.exec_index, check_receipt, rloc)) return IPCFP_ST_FALSE_NO_RECEIPT;    // :224
    Receipt rc = decode_receipt(rloc);
    if (!rc.has_events_root) return IPCFP_ST_FALSE_NO_EVENTS_ROOT;                                      // :229
    AmtRoot events = amt_load(bs, rc.events_root, 3, check_stamped_event);                              // :234
    ValueLoc eloc;
    if (!amt_get(bs, events, p.event_index, check_stamped_event, eloc)) return IPCFP_ST_FALSE_NO_EVENT; // :237
    StampedEvent se = decode_stamped_event(eloc);
    // verify_event_data_matches (:257-290)
    if (se.emitter != p.emitter) return IPCFP_ST_FALSE_EMITTER;                                          // :262
    EvmLog log;
    if (!extract_evm_log(se, log)) return IPCFP_ST_FALSE_NOT_EVM_LOG;                                   // :267
    if (log.topics.size() != p.n_topics) return IPCFP_ST_FALSE_TOPIC_COUNT;                             // :272
    for (size_t i = 0; i < log.topics.size(); ++i)
        if (!eq_ignore_ascii_case(hex0x(log.topics[i].data(), 32), p.topics[i])) return IPCFP_ST_FALSE_TOPIC;  // :276-281
    if (!eq_ignore_ascii_case(hex0x(log.data.data(), log.data.size()), p.data)) return IPCFP_ST_FALSE_DATA;    // :284-287
    if (filter) {  // check_event = create_event_filter(..) (:247-251, :28-39)
        const bool ok = log.topics.size() >= 2 && std::memcmp(log.topics[0].data(), filter->topic0, 32) == 0 &&
                        std::memcmp(log.topics[1].data(), filter->topic1, 32) == 0;
        if (!ok) return IPCFP_ST_FALSE_FILTER;
    }
    return IPCFP_ST_TRUE;
}

// ---- storage proof ----------------------------------------------------------------
Bytes id_address_bytes(uint64_t id) {
    Bytes b{0x00};
    do {
        uint8_t c = id & 0x7f;
        id >>= 7;
        if (id) c |= 0x80;
        b.push_back(c);
    } while (id);
    return b;
}

ActorState get_actor_state(const Blockstore& bs, const Cid& state_root, uint64_t actor_id) {
    Cid actors = decode_state_root_actors(must_get(bs, state_root, "StateRoot"));  // common/decode.rs:23-26
    Bytes key = id_address_bytes(actor_id);                                        // :34
    ValueLoc loc;
    if (!hamt_get(bs, actors, 5, key.data(), key.size(), check_actor_state, loc))  // :29-37
        throw Err(IPCFP_ST_ERR_ACTOR_NOT_FOUND, "actor not found");                // :39
    return decode_actor_state(loc);
}

// SmallMap { v: [[key bytes, value bytes]…] } as a serde (non-tuple) struct: a CBOR map with the
// single required field "v"; unknown fields are ignored by serde's derive, duplicate "v" is an error ⚠.
// Try-reader form: a failed attempt is `r.bad`, not an exception (cbor.hpp ReaderT<false>).
static void try_small_map(TryReader& r, std::vector<std::pair<Bytes, Bytes>>* pairs) {
    const uint64_t n = r.read_map();
    bool have_v = false;
    for (uint64_t i = 0; i < n && !r.bad; ++i) {
        std::string k = r.read_text();
        if (r.bad) return;
        if (k == "v") {
            if (have_v) {
                r.fail("duplicate field v");
                return;
            }
            have_v = true;
            const uint64_t np = r.read_array();
            for (uint64_t j = 0; j < np && !r.bad; ++j) {
                r.expect_array(2);
                Bytes a = r.read_bytes_vec();
                Bytes b = r.read_bytes_vec();
                if (!r.bad && pairs) pairs->emplace_back(std::move(a), std::move(b));
            }
        } else {
            r.skip();
        }
    }
    if (!r.bad && !have_v) r.fail("missing field v");
}

static bool lookup_pairs(const std::vector<std::pair<Bytes, Bytes>>& pairs, const uint8_t slot[32], Bytes& value) {
    for (const auto& kv : pairs)
        if (kv.first.size() == 32 && std::memcmp(kv.first.data(), slot, 32) == 0) {
            value = kv.second;
            return true;
        }
    return false;
}

static bool hamt_value_bytes(const Blockstore& bs, const Cid& root, uint32_t bw, const uint8_t slot[32], Bytes& value) {
    ValueLoc loc;
    if (!hamt_get(bs, root, bw, slot, 32, check_vec_u8, loc)) return false;
    Reader vr(loc.block->data() + loc.off, loc.len);
    const uint64_t n = vr.read_array();
    value.clear();
    for (uint64_t i = 0; i < n; ++i) value.push_back(uint8_t(vr.read_uint()));
    return true;
}

static bool read_storage_slot_throwing(const Blockstore& bs, const Cid& root, const uint8_t slot[32], Bytes& value);
static bool read_storage_slot_try(const Blockstore& bs, const Cid& root, const uint8_t slot[32], Bytes& value);

bool read_storage_slot(const Blockstore& bs, const Cid& root, const uint8_t slot[32], Bytes& value) {
    // IPCFP_ORACLE_CHECK_TRY=1 (the CPU tests set it): every call is ALSO answered by the exception-based form this
    // function replaced, and the two must agree — found / not found, the value, or the Err's status
    static const bool cross_check = [] {
        const char* e = std::getenv("IPCFP_ORACLE_CHECK_TRY");
        return e && e[0] == '1';
    }();
    if (!cross_check) return read_storage_slot_try(bs, root, slot, value);
    Bytes v1, v2;
    bool r1 = false, r2 = false;
    int st1 = -1, st2 = -1;
    try { r1 = read_storage_slot_try(bs, root, slot, v1); } catch (const Err& e) { st1 = e.status; }
    try { r2 = read_storage_slot_throwing(bs, root, slot, v2); } catch (const Err& e) { st2 = e.status; }
    if (st1 != st2 || (st1 < 0 && (r1 != r2 || (r1 && v1 != v2)))) {
        std::fprintf(stderr, "oracle: read_storage_slot try/throw forms disagree (status %d/%d, found %d/%d)\n", st1, st2, int(r1), int(r2));
        std::abort();
    }
    if (st1 >= 0) throw Err(uint8_t(st1), "read_storage_slot");
    value = v1;
    return r1;
}

static bool read_storage_slot_try(const Blockstore& bs, const Cid& root, const uint8_t slot[32], Bytes& value) {
    const Bytes& raw = must_get(bs, root, "contract_state root");  // storage/decode.rs:41-43
    // A1) InlineTupleList(bytes, Vec<SmallMap>)   :46-55 — only the FIRST map is searched; an empty list falls through
    {
        TryReader r(raw);
        r.expect_array(2);
        (void)r.read_bytes_vec();
        const uint64_t n = r.read_array();
        std::vector<std::pair<Bytes, Bytes>> first;
        for (uint64_t i = 0; i < n && !r.bad; ++i) try_small_map(r, i == 0 ? &first : nullptr);
        r.finish();
        if (!r.bad && n > 0) return lookup_pairs(first, slot, value);
    }
    // A2) InlineTuple(bytes, SmallMap)   :58-65
    {
        TryReader r(raw);
        r.expect_array(2);
        (void)r.read_bytes_vec();
        std::vector<std::pair<Bytes, Bytes>> pairs;
        try_small_map(r, &pairs);
        r.finish();
        if (!r.bad) return lookup_pairs(pairs, slot, value);
    }
    // A3) SmallMap   :68-75
    {
        TryReader r(raw);
        std::vector<std::pair<Bytes, Bytes>> pairs;
        try_small_map(r, &pairs);
        r.finish();
        if (!r.bad) return lookup_pairs(pairs, slot, value);
    }
    // B1) MapTuple(Cid, u64)   :78-82
    {
        TryReader r(raw);
        r.expect_array(2);
        Cid inner = read_cid(r);
        uint64_t bw = r.read_uint();
        r.finish();
        if (!r.bad) return hamt_value_bytes(bs, inner, uint32_t(bw & 0xffffffffull), slot, value);  // `bw as u32`
    }
    // B2) MapStruct { root, bitwidth, .. }   :85-89
    {
        TryReader r(raw);
        bool have_root = false, have_bw = false;
        Cid inner;
        uint64_t bw = 0;
        const uint64_t n = r.read_map();
        for (uint64_t i = 0; i < n && !r.bad; ++i) {
            std::string k = r.read_text();
            if (r.bad) break;
            if (k == "root") {
                if (have_root) r.fail("duplicate field root");
                else inner = read_cid(r);
                have_root = true;
            } else if (k == "bitwidth") {
                if (have_bw) r.fail("duplicate field bitwidth");
                else bw = r.read_uint();
                have_bw = true;
            } else {
                r.skip();
            }
        }
        r.finish();
        if (!r.bad && have_root && have_bw) return hamt_value_bytes(bs, inner, uint32_t(bw & 0xffffffffull), slot, value);
    }
    // C) direct HAMT, bit width 5   :92-96
    return hamt_value_bytes(bs, root, 5, slot, value);
}

// ---- the exception-based form the try-reader form replaced, kept as its cross-check (IPCFP_ORACLE_CHECK_TRY) ----
static bool throwing_small_map(Reader& r, std::vector<std::pair<Bytes, Bytes>>& pairs) {
    const uint64_t n = r.read_map();
    bool have_v = false;
    for (uint64_t i = 0; i < n; ++i) {
        std::string k = r.read_text();
        if (k == "v") {
            if (have_v) decode_err("duplicate field v");
            have_v = true;
            const uint64_t np = r.read_array();
            for (uint64_t j = 0; j < np; ++j) {
                r.expect_array(2);
                Bytes a = r.read_bytes_vec();
                Bytes b = r.read_bytes_vec();
                pairs.emplace_back(std::move(a), std::move(b));
            }
        } else {
            r.skip();
        }
    }
    if (!have_v) decode_err("missing field v");
    return true;
}

static bool read_storage_slot_throwing(const Blockstore& bs, const Cid& root, const uint8_t slot[32], Bytes& value) {
    const Bytes& raw = must_get(bs, root, "contract_state root");  // storage/decode.rs:41-43
    // A1) InlineTupleList(bytes, Vec<SmallMap>)   :46-55
    try {
        Reader r(raw);
        r.expect_array(2);
        (void)r.read_bytes_vec();
        const uint64_t n = r.read_array();
        std::vector<std::vector<std::pair<Bytes, Bytes>>> maps(n);
        for (uint64_t i = 0; i < n; ++i) throwing_small_map(r, maps[i]);
        r.finish();
        if (!maps.empty()) return lookup_pairs(maps[0], slot, value);  // first map only; empty list falls through
    } catch (const Err&) {
    }
    // A2) InlineTuple(bytes, SmallMap)   :58-65
    try {
        Reader r(raw);
        r.expect_array(2);
        (void)r.read_bytes_vec();
        std::vector<std::pair<Bytes, Bytes>> pairs;
        throwing_small_map(r, pairs);
        r.finish();
        return lookup_pairs(pairs, slot, value);
    } catch (const Err&) {
    }
    // A3) SmallMap   :68-75
    try {
        Reader r(raw);
        std::vector<std::pair<Bytes, Bytes>> pairs;
        throwing_small_map(r, pairs);
        r.finish();
        return lookup_pairs(pairs, slot, value);
    } catch (const Err&) {
    }
    // B1) MapTuple(Cid, u64)   :78-82
    {
        bool ok = false;
        Cid inner;
        uint64_t bw = 0;
        try {
            Reader r(raw);
            r.expect_array(2);
            inner = read_cid(r);
            bw = r.read_uint();
            r.finish();
            ok = true;
        } catch (const Err&) {
        }
        if (ok) {
            ValueLoc loc;
            if (bw > 0xffffffffull) bw &= 0xffffffffull;  // `bw as u32`
            if (!hamt_get(bs, inner, uint32_t(bw), slot, 32, check_vec_u8, loc)) return false;
            Reader vr(loc.block->data() + loc.off, loc.len);
            const uint64_t n = vr.read_array();
            value.clear();
            for (uint64_t i = 0; i < n; ++i) value.push_back(uint8_t(vr.read_uint()));
            return true;
        }
    }
    // B2) MapStruct { root, bitwidth, .. }   :85-89
    {
        bool ok = false, have_root = false, have_bw = false;
        Cid inner;
        uint64_t bw = 0;
        try {
            Reader r(raw);
            const uint64_t n = r.read_map();
            for (uint64_t i = 0; i < n; ++i) {
                std::string k = r.read_text();
                if (k == "root") {
                    if (have_root) decode_err("duplicate field root");
                    inner = read_cid(r);
                    have_root = true;
                } else if (k == "bitwidth") {
                    if (have_bw) decode_err("duplicate field bitwidth");
                    bw = r.read_uint();
                    have_bw = true;
                } else {
                    r.skip();
                }
            }
            r.finish();
            ok = have_root && have_bw;
        } catch (const Err&) {
        }
        if (ok) {
            ValueLoc loc;
            if (!hamt_get(bs, inner, uint32_t(bw & 0xffffffffull), slot, 32, check_vec_u8, loc)) return false;
            Reader vr(loc.block->data() + loc.off, loc.len);
            const uint64_t n = vr.read_array();
            value.clear();
            for (uint64_t i = 0; i < n; ++i) value.push_back(uint8_t(vr.read_uint()));
            return true;
        }
    }
    // C) direct HAMT, bit width 5   :92-96
    ValueLoc loc;
    if (!hamt_get(bs, root, 5, slot, 32, check_vec_u8, loc)) return false;
    Reader vr(loc.block->data() + loc.off, loc.len);
    const uint64_t n = vr.read_array();
    value.clear();
    for (uint64_t i = 0; i < n; ++i) value.push_back(uint8_t(vr.read_uint()));
    return true;
}

static bool hex_decode_32(const char* s, uint8_t out[32]) {
    // hex::decode_to_slice(slot_hex.trim_start_matches("0x"), &mut [u8; 32])  storage/verifier.rs:155-157
    if (!s) return false;
    while (s[0] == '0' && s[1] == 'x') s += 2;  // trim_start_matches strips EVERY leading "0x"
    if (std::strlen(s) != 64) return false;
    for (int i = 0; i < 32; ++i) {
        auto hv = [](char c) -> int {
            if (c >= '0' && c <= '9') return c - '0';
            if (c >= 'a' && c <= 'f') return c - 'a' + 10;
            if (c >= 'A' && c <= 'F') return c - 'A' + 10;
            return -1;
        };
        const int h = hv(s[2 * i]), l = hv(s[2 * i + 1]);
        if (h < 0 || l < 0) return false;
        out[i] = uint8_t(h * 16 + l);
    }
    return true;
}

uint8_t verify_storage_proof_one(const Blockstore& bs, const ipcfp_storage_proof_t& p,
                                 const ipcfp_trust_policy_t* trust) {
    // Step 2: verify_trust_anchor (storage/verifier.rs:81-92)
    Cid child = parse_claim_cid(p.child_block_cid);                         // :85
    if (!trusted(trust, p.child_epoch)) return IPCFP_ST_FALSE_UNTRUSTED_CHILD;  // :87
    // Step 3: verify_parent_state_root (:95-111)
    (void)parse_claim_cid(p.child_block_cid);                               // :38
    HeaderLite hdr = decode_header(must_get(bs, child, "child header"));    // :101-107
    if (cid_to_string(hdr.parent_state_root) != std::string(p.parent_state_root ? p.parent_state_root : ""))
        return IPCFP_ST_FALSE_STATE_ROOT;                                   // :110 (case-sensitive)
    // Step 4: verify_actor_state (:114-127)
    Cid state_root = parse_claim_cid(p.parent_state_root);                  // :44
    ActorState actor = get_actor_state(bs, state_root, p.actor_id);         // :122
    if (cid_to_string(actor.state) != std::string(p.actor_state_cid ? p.actor_state_cid : ""))
        return IPCFP_ST_FALSE_ACTOR_STATE;                                  // :126
    // Step 5: verify_storage_root (:130-145)
    Cid actor_state_cid = parse_claim_cid(p.actor_state_cid);               // :55
    Cid contract_state = parse_evm_state_contract(must_get(bs, actor_state_cid, "EVM state"));  // :136-141
    if (cid_to_string(contract_state) != std::string(p.storage_root ? p.storage_root : ""))
        return IPCFP_ST_FALSE_STORAGE_ROOT;                                 // :144
    // Step 6: verify_storage_value (:148-170)
    Cid storage_root = parse_claim_cid(p.storage_root);                     // :61
    uint8_t slot[32];
    if (!hex_decode_32(p.slot, slot)) throw Err(IPCFP_ST_ERR_BAD_CLAIM, "Invalid slot hex format");  // :155-157
    Bytes raw;
    if (!read_storage_slot(bs, storage_root, slot, raw)) raw.clear();       // :160-162 missing ⇒ zero
    uint8_t padded[32] = {0};                                               // left_pad_32 (common/evm.rs:91-100)
    if (raw.size() >= 32) std::memcpy(padded, raw.data() + raw.size() - 32, 32);
    else if (!raw.empty()) std::memcpy(padded + 32 - raw.size(), raw.data(), raw.size());  // (an empty value has no data())
    return eq_ignore_ascii_case(hex0x(padded, 32), p.value) ? IPCFP_ST_TRUE : IPCFP_ST_FALSE_VALUE;  // :165-169
}

// ---- event scan ------------------------------------------------------------------------
void scan_events(const Blockstore& bs, const Cid& receipts_root, const ipcfp_event_filter_t& filter, bool has_actor,
                 uint64_t actor, std::vector<uint8_t>& receipt_has_match, std::vector<ScanMatch>& matches,
                 std::vector<Cid>* touched, int threads) {
    receipt_has_match.clear();
    matches.clear();
    std::set<Cid> needed;
    // `Amtv0::<Receipt>::load(&receipts_root, &rec_receipts)` (generator.rs:195-196)
    RecordingBlockStore rec_receipts(bs);
    AmtRoot r_amt = amt_load(rec_receipts, receipts_root, 0, check_receipt);
    // The reference takes (index, events_root) from the RPC receipt list (:199-204).  Offline, the
    // same list is the receipts AMT walked in index order on a throw-away store.
    struct Item { uint64_t index; Receipt rc; };
    std::vector<Item> receipts;
    {
        AmtRoot plain = amt_load(bs, receipts_root, 0, check_receipt);
        if (threads == 1) {
            amt_for_each(bs, plain, check_receipt,
                         [&](uint64_t i, const ValueLoc& v) { receipts.push_back({i, decode_receipt(v)}); });
        } else {
            std::vector<AmtItem> items;
            amt_collect(bs, plain, check_receipt, items, threads);
            receipts.resize(items.size());
#pragma omp parallel for schedule(static)
            for (int64_t k = 0; k < int64_t(items.size()); ++k)
                receipts[size_t(k)] = Item{items[size_t(k)].index, decode_receipt(items[size_t(k)].value)};
        }
    }
    auto matches_log = [&](const EvmLog& log) {  // EventMatcher::matches_log (:38-40)
        return log.topics.size() >= 2 && std::memcmp(log.topics[0].data(), filter.topic0, 32) == 0 &&
               std::memcmp(log.topics[1].data(), filter.topic1, 32) == 0;
    };
    uint64_t max_index = 0;
    for (const auto& it : receipts) max_index = std::max(max_index, it.index + 1);
    receipt_has_match.assign(max_index, 0);
    // PASS 1 (:209-239)
    auto pass1_one = [&](size_t k) -> bool {
        const auto& it = receipts[k];
        if (!it.rc.has_events_root) return false;
        RecordingBlockStore temp(bs);
        AmtRoot e_amt = amt_load(temp, it.rc.events_root, 3, check_stamped_event);
        bool has = false;
        amt_for_each(temp, e_amt, check_stamped_event, [&](uint64_t, const ValueLoc& v) {
            StampedEvent se = decode_stamped_event(v);
            if (has_actor && se.emitter != actor) return;
            EvmLog log;
            if (extract_evm_log(se, log) && matches_log(log)) has = true;
        });
        return has;
    };
    std::vector<size_t> matching;
    if (threads == 1) {
        for (size_t k = 0; k < receipts.size(); ++k)
            if (pass1_one(k)) {
                matching.push_back(k);
                receipt_has_match[receipts[k].index] = 1;
            }
    } else {
        use_threads(threads);
        std::vector<uint8_t> has(receipts.size(), 0);
        // the sequential loop stops at the first failing receipt: keep the failure with the smallest k
        size_t err_k = SIZE_MAX;
        uint8_t err_status = 0;
        std::string err_what;
#pragma omp parallel for schedule(dynamic, 512)
        for (int64_t k = 0; k < int64_t(receipts.size()); ++k) {
            try {
                has[size_t(k)] = pass1_one(size_t(k)) ? 1 : 0;
            } catch (const Err& e) {
#pragma omp critical(orc_scan_err)
                if (size_t(k) < err_k) {
                    err_k = size_t(k);
                    err_status = e.status;
                    err_what = e.what();
                }
            }
        }
        if (err_k != SIZE_MAX) throw Err(err_status, err_what);
        for (size_t k = 0; k < receipts.size(); ++k)
            if (has[k]) {
                matching.push_back(k);
                receipt_has_match[receipts[k].index] = 1;
            }
    }
    // PASS 2 (:242-301)
    for (size_t k : matching) {
        const auto& it = receipts[k];
        ValueLoc loc;
        if (!amt_get(rec_receipts, r_amt, it.index, check_receipt, loc)) continue;  // :249-251
        RecordingBlockStore rec_events(bs);
        AmtRoot e_amt = amt_load(rec_events, it.rc.events_root, 3, check_stamped_event);
        amt_for_each(rec_events, e_amt, check_stamped_event, [&](uint64_t j, const ValueLoc& v) {
            StampedEvent se = decode_stamped_event(v);
            if (has_actor && se.emitter != actor) return;
            EvmLog log;
            if (extract_evm_log(se, log) && matches_log(log))
                matches.push_back({it.index, j, se.emitter, log.topics, log.data});
        });
        for (const Cid& c : rec_events.take_seen()) needed.insert(c);  // collector.collect_from_recording (:99-101)
    }
    for (const Cid& c : rec_receipts.take_seen()) needed.insert(c);    // "Add receipts recording last" (:304)
    if (touched) touched->assign(needed.begin(), needed.end());
}

// ---- generator side --------------------------------------------------------------------------
GeneratedEventBundle generate_event_proof(const Blockstore& bs, const std::vector<Cid>& parent_cids, const Cid& child_cid,
                                          const ipcfp_event_filter_t& filter, bool has_actor, uint64_t actor) {
    GeneratedEventBundle out;
    std::set<Cid> needed;
    // Step 1: extract_child_info (generator.rs:112-119) — receipts root of the child header
    HeaderLite child_hdr = decode_header(must_get(bs, child_cid, "child header"));
    const Cid receipts_root = child_hdr.parent_message_receipts;
    // Step 2: collect_base_witness (:122-145)
    std::vector<Cid> txmeta;
    for (const Cid& p : parent_cids) {
        needed.insert(p);
        txmeta.push_back(decode_header(must_get(bs, p, "parent header")).messages);
    }
    needed.insert(child_cid);
    needed.insert(receipts_root);
    for (const Cid& t : txmeta) needed.insert(t);
    // Step 3: record_transaction_amts (:148-177)
    for (const Cid& t : txmeta) {
        RecordingBlockStore rec(bs);
        const Bytes& raw = must_get(rec, t, "TxMeta");
        Reader r(raw);
        r.expect_array(2);
        Cid bls = read_cid(r), secp = read_cid(r);
        r.finish();
        for (const Cid* root : {&bls, &secp}) {
            AmtRoot a = amt_load(rec, *root, 0, check_cid_value);
            amt_for_each(rec, a, check_cid_value, [](uint64_t, const ValueLoc&) {});
        }
        for (const Cid& c : rec.take_seen()) needed.insert(c);
    }
    // Step 4: build_execution_order → collect_exec_list(verify_txmeta = false) (utils.rs:33-45)
    std::vector<Cid> exec;
    {
        std::unordered_set<Bytes, BytesHash> seen;  // HashSet<Cid> (events/utils.rs:54)
        for (const Cid& t : txmeta) {
            const Bytes& raw = must_get(bs, t, "TxMeta");
            Reader r(raw);
            r.expect_array(2);
            Cid bls = read_cid(r), secp = read_cid(r);
            r.finish();
            for (const Cid* root : {&bls, &secp}) {
                AmtRoot a = amt_load(bs, *root, 0, check_cid_value);
                amt_for_each(bs, a, check_cid_value, [&](uint64_t, const ValueLoc& v) {
                    Reader vr(v.block->data() + v.off, v.len);
                    Cid c = read_cid(vr);
                    if (seen.insert(c.b).second) exec.push_back(c);
                });
            }
        }
    }
    // Step 5: find_matching_events (:180-307)
    std::vector<uint8_t> has;
    std::vector<ScanMatch> ms;
    std::vector<Cid> touched;
    scan_events(bs, receipts_root, filter, has_actor, actor, has, ms, &touched);
    for (const auto& m : ms) {
        if (m.exec_index >= exec.size()) throw Err(IPCFP_ST_ERR, "Missing message at index");  // :244-246
        out.proofs.push_back({m.exec_index, m.event_index, m.emitter, exec[m.exec_index], m.topics, m.data});
    }
    for (const Cid& c : touched) needed.insert(c);  // :98-101
    // Step 6: materialize (witness.rs:43-56)
    for (const Cid& c : needed) {
        (void)must_get(bs, c, "block");
        out.witness.push_back(c);
    }
    return out;
}

GeneratedStorageProof generate_storage_proof(const Blockstore& bs, const Cid& child_cid, uint64_t actor_id,
                                             const uint8_t slot[32]) {
    GeneratedStorageProof out;
    std::set<Cid> needed;
    // Step 1: extract_and_verify_parent_state (storage/generator.rs:72-103)
    HeaderLite hdr = decode_header(must_get(bs, child_cid, "child header"));
    out.parent_state_root = hdr.parent_state_root;
    needed.insert(child_cid);             // :41
    needed.insert(out.parent_state_root); // :42
    // Step 3: load_actor_and_storage_root (:106-134)
    {
        RecordingBlockStore rec(bs);
        ActorState actor = get_actor_state(rec, out.parent_state_root, actor_id);
        out.actor_state_cid = actor.state;
        out.storage_root = parse_evm_state_contract(must_get(rec, out.actor_state_cid, "EVM state"));
        needed.insert(out.actor_state_cid);
        needed.insert(out.storage_root);
        for (const Cid& c : rec.take_seen()) needed.insert(c);
    }
    // Step 4: read_storage_value (:137-155)
    {
        RecordingBlockStore rec(bs);
        Bytes raw;
        if (!read_storage_slot(rec, out.storage_root, slot, raw)) raw.clear();
        std::memset(out.value, 0, 32);
        if (raw.size() >= 32) std::memcpy(out.value, raw.data() + raw.size() - 32, 32);
        else if (!raw.empty()) std::memcpy(out.value + 32 - raw.size(), raw.data(), raw.size());  // (an empty value has no data())
        for (const Cid& c : rec.take_seen()) needed.insert(c);
    }
    for (const Cid& c : needed) {
        (void)must_get(bs, c, "block");
        out.witness.push_back(c);
    }
    return out;
}

}  // namespace orc
