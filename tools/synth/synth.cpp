// tools/synth/synth.cpp — seeded synthetic Filecoin tipset / state-tree generator.
//
// There is no network here, so the `RpcBlockstore` of the reference
// (src/client/blockstore.rs:20-37) is replaced by a generated chain fragment with REAL
// CIDs: parent headers → TxMeta → BLS/secp message AMTs (v0); a child header →
// receipts AMT (v0) → per-receipt events AMTs (v3, bit width 5) of StampedEvents;
// StateRoot → actors HAMT (v3, bit width 5) → ActorState → EvmState → storage HAMT.
// Shapes follow SURVEY.md Appendix A and §8(d); PRNG = splitmix64(seed).
//
// This is the WRITER side (encoders, AMT/HAMT builders, its own hashes in hash_min.hpp);
// the oracle and the HIP engine are independent READERS of what it writes.  It is input
// tooling: it is not part of the product and does not use oracle/.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "hash_min.hpp"

namespace synth {

using Bytes = std::vector<uint8_t>;
using Cid = std::array<uint8_t, 38>;

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    void fill(uint8_t* p, size_t n) {
        for (size_t i = 0; i < n; i += 8) {
            uint64_t v = next();
            std::memcpy(p + i, &v, n - i < 8 ? n - i : 8);
        }
    }
};

static Cid cid_of(const uint8_t* p, size_t n) {
    Cid c = {0x01, 0x71, 0xa0, 0xe4, 0x02, 0x20};
    blake2b256(p, n, c.data() + 6);
    return c;
}
static Cid fake_cid(Rng& r) {
    Cid c = {0x01, 0x71, 0xa0, 0xe4, 0x02, 0x20};
    r.fill(c.data() + 6, 32);
    return c;
}

// ---- DAG-CBOR writer ----
struct W {
    Bytes b;
    void head(int major, uint64_t v) {
        const uint8_t m = uint8_t(major << 5);
        if (v < 24) b.push_back(m | uint8_t(v));
        else if (v <= 0xff) { b.push_back(m | 24); b.push_back(uint8_t(v)); }
        else if (v <= 0xffff) { b.push_back(m | 25); b.push_back(uint8_t(v >> 8)); b.push_back(uint8_t(v)); }
        else if (v <= 0xffffffffULL) { b.push_back(m | 26); for (int s = 24; s >= 0; s -= 8) b.push_back(uint8_t(v >> s)); }
        else { b.push_back(m | 27); for (int s = 56; s >= 0; s -= 8) b.push_back(uint8_t(v >> s)); }
    }
    void uint(uint64_t v) { head(0, v); }
    void sint(int64_t v) { if (v >= 0) head(0, uint64_t(v)); else head(1, uint64_t(-1 - v)); }
    void bytes(const uint8_t* p, size_t n) { head(2, n); b.insert(b.end(), p, p + n); }
    void bytes(const Bytes& v) { bytes(v.data(), v.size()); }
    void text(const char* s) { size_t n = std::strlen(s); head(3, n); b.insert(b.end(), s, s + n); }
    void array(uint64_t n) { head(4, n); }
    void map(uint64_t n) { head(5, n); }
    void null() { b.push_back(0xf6); }
    void link(const uint8_t* cid, size_t n) {
        b.push_back(0xd8); b.push_back(0x2a);
        head(2, n + 1);
        b.push_back(0x00);
        b.insert(b.end(), cid, cid + n);
    }
    void link(const Cid& c) { link(c.data(), c.size()); }
    void raw(const Bytes& v) { b.insert(b.end(), v.begin(), v.end()); }
};

// ---- witness sink ----
struct Sink {
    Bytes bytes;
    std::vector<uint64_t> off;
    std::vector<uint32_t> len;
    Bytes cids;  // 40 per block
    uint64_t payload = 0;
    Cid add(const Bytes& block, bool keep = true) {
        Cid c = cid_of(block.data(), block.size());
        if (keep) put(c, block);
        return c;
    }
    void put(const Cid& c, const Bytes& block) {
        while (bytes.size() % 16) bytes.push_back(0);
        off.push_back(bytes.size());
        len.push_back(uint32_t(block.size()));
        bytes.insert(bytes.end(), block.begin(), block.end());
        cids.insert(cids.end(), c.begin(), c.end());
        cids.push_back(0); cids.push_back(0);
        payload += block.size();
    }
};

// ---- AMT builder ----
struct AmtVal { uint64_t index; Bytes enc; };

static Bytes amt_node(Sink& sink, uint32_t bw, uint64_t height, const AmtVal* v, size_t n, uint64_t base, bool keep) {
    const uint32_t width = 1u << bw;
    Bytes bmap((width + 7) / 8, 0);
    W w;
    w.array(3);
    if (height == 0) {
        for (size_t i = 0; i < n; ++i) { const uint64_t k = v[i].index - base; bmap[k / 8] |= uint8_t(1u << (k % 8)); }
        w.bytes(bmap);
        w.array(0);
        w.array(n);
        for (size_t i = 0; i < n; ++i) w.raw(v[i].enc);
        return w.b;
    }
    uint64_t span = 1;
    for (uint64_t h = 0; h < height; ++h) span *= width;
    std::vector<Cid> links;
    size_t i = 0;
    while (i < n) {
        const uint64_t sub = (v[i].index - base) / span;
        size_t j = i;
        while (j < n && (v[j].index - base) / span == sub) ++j;
        Bytes child = amt_node(sink, bw, height - 1, v + i, j - i, base + sub * span, keep);
        links.push_back(sink.add(child, keep));
        bmap[sub / 8] |= uint8_t(1u << (sub % 8));
        i = j;
    }
    w.bytes(bmap);
    w.array(links.size());
    for (auto& c : links) w.link(c);
    w.array(0);
    return w.b;
}

// version 0: [height, count, node] (bw 3); version 3: [bw, height, count, node]
static Cid build_amt(Sink& sink, int version, uint32_t bw, const std::vector<AmtVal>& vals, bool keep = true) {
    const uint32_t width = 1u << bw;
    uint64_t height = 0;
    if (!vals.empty()) {
        const uint64_t maxi = vals.back().index;
        uint64_t cap = width;
        while (maxi >= cap) { cap *= width; ++height; }
    }
    Bytes node = amt_node(sink, bw, height, vals.data(), vals.size(), 0, keep);
    W w;
    if (version == 0) w.array(3); else { w.array(4); w.uint(bw); }
    w.uint(height);
    w.uint(vals.size());
    w.raw(node);
    return sink.add(w.b, keep);
}

// ---- HAMT builder ----
struct HEntry {
    std::array<uint8_t, 32> hash;
    Bytes key, val;
};
static uint32_t hbits(const std::array<uint8_t, 32>& h, uint32_t pos, uint32_t bw) {
    uint32_t v = 0;
    for (uint32_t b = 0; b < bw; ++b) { const uint32_t bit = pos + b; v = (v << 1) | ((h[bit / 8] >> (7 - bit % 8)) & 1u); }
    return v;
}
struct HamtKeep {
    bool all = true;
    const std::vector<std::array<uint8_t, 32>>* wanted = nullptr;  // sorted hashes whose paths must be kept
    bool wants(const std::array<uint8_t, 32>& lo_hash, uint32_t prefix_bits) const {
        if (all) return true;
        if (!wanted || wanted->empty()) return false;
        // any wanted hash sharing the first prefix_bits bits with lo_hash?
        std::array<uint8_t, 32> lo = lo_hash, hi = lo_hash;
        for (uint32_t bit = prefix_bits; bit < 256; ++bit) { lo[bit / 8] &= uint8_t(~(0x80u >> (bit % 8))); hi[bit / 8] |= uint8_t(0x80u >> (bit % 8)); }
        auto it = std::lower_bound(wanted->begin(), wanted->end(), lo);
        return it != wanted->end() && !(hi < *it);
    }
};

// entries[lo,hi) sorted by hash and sharing the first depth*bw bits
static Bytes hamt_node(Sink& sink, uint32_t bw, uint32_t depth, const HEntry* e, size_t n, const HamtKeep& keep) {
    uint8_t bf[32] = {0};
    std::vector<Bytes> ptrs;
    size_t i = 0;
    while (i < n) {
        const uint32_t idx = hbits(e[i].hash, depth * bw, bw);
        size_t j = i;
        while (j < n && hbits(e[j].hash, depth * bw, bw) == idx) ++j;
        bf[31 - idx / 8] |= uint8_t(1u << (idx % 8));
        W p;
        if (j - i <= 3) {
            std::vector<const HEntry*> b;
            for (size_t k = i; k < j; ++k) b.push_back(&e[k]);
            std::sort(b.begin(), b.end(), [](const HEntry* x, const HEntry* y) { return x->key < y->key; });
            p.array(b.size());
            for (auto* kv : b) { p.array(2); p.bytes(kv->key); p.raw(kv->val); }
        } else {
            Bytes child = hamt_node(sink, bw, depth + 1, e + i, j - i, keep);
            const bool k = keep.wants(e[i].hash, (depth + 1) * bw);
            p.link(sink.add(child, k));
        }
        ptrs.push_back(std::move(p.b));
        i = j;
    }
    size_t lead = 0;
    while (lead < 32 && bf[lead] == 0) ++lead;
    W w;
    w.array(2);
    w.bytes(bf + lead, 32 - lead);
    w.array(ptrs.size());
    for (auto& p : ptrs) w.raw(p);
    return w.b;
}

static Cid build_hamt(Sink& sink, uint32_t bw, std::vector<HEntry>& entries, const HamtKeep& keep) {
    for (auto& e : entries) sha256(e.key.data(), e.key.size(), e.hash.data());
    std::sort(entries.begin(), entries.end(), [](const HEntry& a, const HEntry& b) { return a.hash < b.hash; });
    Bytes root = hamt_node(sink, bw, 0, entries.data(), entries.size(), keep);
    return sink.add(root, true);
}

static Bytes id_addr(uint64_t id) {
    Bytes b{0x00};
    do { uint8_t c = id & 0x7f; id >>= 7; if (id) c |= 0x80; b.push_back(c); } while (id);
    return b;
}

// ---- parameters / outputs (plain C structs for ctypes) ----
extern "C" {
struct synth_params {
    uint64_t seed;
    int64_t parent_epoch;
    uint32_t n_parents;        // parent block headers (>= 1)
    uint32_t dup_permille;     // per-mille of extra duplicate messages re-included by later parent blocks
    uint64_t n_receipts;       // = executed messages
    uint32_t max_events;       // events per receipt = 1 + PRNG % max_events; 0 ⇒ no receipt has events
    uint32_t no_events_permille;  // receipts whose events_root is null
    uint32_t n_emitters;       // emitters = emitter_base … +n_emitters-1
    uint32_t emitter_base;
    uint32_t n_sigs;           // topic0 drawn from n_sigs event signatures (index 0 = the filter's)
    uint32_t n_subnets;        // topic1 drawn from "calib-subnet-k", k < n_subnets (k = 1 is the filter's)
    uint64_t n_planted;        // receipts forced to contain one matching event
    uint64_t filter_actor;     // emitter of planted matches
    uint32_t variety;          // 1 ⇒ mix in Case-A logs, non-EVM events, bad topic lengths, duplicate keys, 3-4 topics
    uint32_t events_bit_width; // bit width of the events AMTs (FVM: 5)
    // state tree
    uint64_t n_actors;         // ids actor_base … ; 0 ⇒ no state tree
    uint64_t actor_base;
    uint32_t n_contracts;      // the first n_contracts actors are EVM actors with storage
    uint32_t slots_per_contract;
    uint32_t storage_layout_mix;  // 1 ⇒ contracts cycle through layouts C,B1,B2,A1,A2,A3; 0 ⇒ all C
    uint32_t keep_full_state;  // 1 ⇒ every state-tree node is put in the witness; 0 ⇒ only query paths
    uint64_t n_actor_queries;  // PRNG-chosen present ids (+1% absent) whose paths are kept
};
}

struct EventRec {
    uint64_t exec_index, event_index, emitter;
    uint32_t n_topics;       // what extract_evm_log yields (0 ⇒ not an EVM log)
    std::array<uint8_t, 128> topics;
    uint32_t data_len;
    std::array<uint8_t, 64> data;
};

struct Tipset {
    synth_params p;
    Sink sink;
    std::vector<Cid> parent_cids;
    Cid child_cid, receipts_root, state_root_cid, actors_root;
    std::vector<Cid> exec_order;
    std::vector<EventRec> claims;      // one per receipt that has events (event chosen by PRNG)
    std::vector<uint64_t> planted;     // exec indices with a planted match
    std::array<uint8_t, 32> topic0, topic1;
    // storage side
    std::vector<uint64_t> query_ids;       // actor get queries (present + absent)
    std::vector<uint8_t> query_present;
    struct StorageClaim { uint64_t actor_id; Cid actor_state, storage_root; std::array<uint8_t, 32> slot, value; uint8_t present; };
    std::vector<StorageClaim> storage_claims;
    uint64_t receipts_amt_bytes = 0, events_amt_bytes = 0, msg_amt_bytes = 0, state_bytes = 0;
};

static const char* kSigs[16] = {
    "NewTopDownMessage(bytes32,uint256)", "Transfer(address,address,uint256)", "Approval(address,address,uint256)",
    "Deposit(address,uint256)", "Withdrawal(address,uint256)", "Swap(address,uint256,uint256,uint256,uint256,address)",
    "Sync(uint112,uint112)", "Mint(address,uint256,uint256)", "Burn(address,uint256,uint256,address)",
    "OwnershipTransferred(address,address)", "Upgraded(address)", "Paused(address)", "Unpaused(address)",
    "RoleGranted(bytes32,address,address)", "RoleRevoked(bytes32,address,address)", "Initialized(uint8)"};

static std::array<uint8_t, 32> ascii32(const std::string& s) {
    std::array<uint8_t, 32> o{};
    std::memcpy(o.data(), s.data(), s.size() < 32 ? s.size() : 32);
    return o;
}

static Bytes header_block(Rng& r, const std::vector<Cid>& parents, int64_t height, const Cid& state, const Cid& receipts,
                          const Cid& messages) {
    W w;
    uint8_t blob[192];
    w.array(16);
    { Bytes a = id_addr(1000 + r.below(5000)); w.bytes(a); }                     // 0 miner
    r.fill(blob, 96); w.array(1); w.bytes(blob, 96);                              // 1 ticket
    r.fill(blob, 96); w.array(2); w.uint(1 + r.below(3)); w.bytes(blob, 96);      // 2 election proof
    w.array(1); w.array(2); w.uint(3000000 + r.below(1000)); r.fill(blob, 96); w.bytes(blob, 96);  // 3 beacon entries
    w.array(1); w.array(2); w.uint(3); r.fill(blob, 192); w.bytes(blob, 192);     // 4 winpost proof
    w.array(parents.size()); for (auto& c : parents) w.link(c);                   // 5 parents
    { uint8_t wt[9] = {0}; r.fill(wt + 1, 8); w.bytes(wt, 9); }                    // 6 parent weight
    w.sint(height);                                                                // 7
    w.link(state); w.link(receipts); w.link(messages);                             // 8 9 10
    w.array(2); w.uint(2); r.fill(blob, 96); w.bytes(blob, 96);                    // 11 bls aggregate
    w.uint(1700000000 + uint64_t(height) * 30);                                    // 12 timestamp
    w.array(2); w.uint(2); r.fill(blob, 96); w.bytes(blob, 96);                    // 13 block sig
    w.uint(0);                                                                     // 14 fork signaling
    { uint8_t fee[3] = {0, 0x64, 0}; w.bytes(fee, 2); }                            // 15 parent base fee
    return w.b;
}

static void encode_event(W& w, Rng& r, uint64_t emitter, const std::array<uint8_t, 32>& t0, const std::array<uint8_t, 32>& t1,
                         uint32_t shape, EventRec& rec) {
    // shape 0: compact t1,t2,d (flags 3, codec 0x55) — the default of SURVEY.md §8(d)
    uint8_t data[64];
    r.fill(data, 64);
    rec.emitter = emitter;
    rec.n_topics = 0;
    rec.data_len = 0;
    rec.topics.fill(0);
    rec.data.fill(0);
    auto entry = [&](const char* key, const uint8_t* v, size_t n) { w.array(4); w.uint(3); w.text(key); w.uint(0x55); w.bytes(v, n); };
    auto set_topics = [&](std::initializer_list<const uint8_t*> ts) { rec.n_topics = 0; for (auto t : ts) { std::memcpy(rec.topics.data() + 32 * rec.n_topics, t, 32); ++rec.n_topics; } };
    auto set_data = [&](const uint8_t* d, size_t n) { rec.data_len = uint32_t(n); std::memcpy(rec.data.data(), d, n); };
    w.array(2);
    w.uint(emitter);
    uint8_t t3[32], t4[32];
    r.fill(t3, 32); r.fill(t4, 32);
    switch (shape) {
        default:
        case 0:
            w.array(3); entry("t1", t0.data(), 32); entry("t2", t1.data(), 32); entry("d", data, 32);
            set_topics({t0.data(), t1.data()}); set_data(data, 32);
            break;
        case 1: {  // Case A: "topics" (concatenated) + "data"
            uint8_t cat[64];
            std::memcpy(cat, t0.data(), 32); std::memcpy(cat + 32, t1.data(), 32);
            w.array(2); entry("topics", cat, 64); entry("data", data, 48);
            set_topics({t0.data(), t1.data()}); set_data(data, 48);
            break;
        }
        case 2:  // not an EVM log: no t1 / topics key
            w.array(2); entry("foo", data, 8); entry("d", data, 32);
            break;
        case 3:  // t2 has the wrong length ⇒ None
            w.array(3); entry("t1", t0.data(), 32); entry("t2", t1.data(), 31); entry("d", data, 32);
            break;
        case 4:  // duplicate key: the LAST t2 wins; 4 topics; no data entry
            w.array(5); entry("t1", t0.data(), 32); entry("t2", t3, 32); entry("t3", t3, 32); entry("t4", t4, 32); entry("t2", t1.data(), 32);
            set_topics({t0.data(), t1.data(), t3, t4});
            break;
        case 5:  // single topic, empty data
            w.array(2); entry("t1", t0.data(), 32); entry("d", data, 0);
            set_topics({t0.data()});
            break;
        case 6:  // Case A with a length that is not a multiple of 32 ⇒ None
            w.array(1); entry("topics", data, 40);
            break;
        case 7:  // t1 then a gap (t3 without t2): stops at the first missing ⇒ 1 topic
            w.array(3); entry("t1", t0.data(), 32); entry("t3", t3, 32); entry("d", data, 16);
            set_topics({t0.data()}); set_data(data, 16);
            break;
    }
}

static Tipset* build(const synth_params& p) {
    Tipset* T = new Tipset();
    T->p = p;
    Rng r(p.seed);
    Sink& sink = T->sink;
    const uint32_t n_sigs = p.n_sigs ? (p.n_sigs > 16 ? 16 : p.n_sigs) : 1;
    std::vector<std::array<uint8_t, 32>> sig_hash(n_sigs);
    for (uint32_t i = 0; i < n_sigs; ++i) keccak256(reinterpret_cast<const uint8_t*>(kSigs[i]), std::strlen(kSigs[i]), sig_hash[i].data());
    const uint32_t n_sub = p.n_subnets ? p.n_subnets : 1;
    std::vector<std::array<uint8_t, 32>> subnets(n_sub);
    for (uint32_t k = 0; k < n_sub; ++k) subnets[k] = ascii32("calib-subnet-" + std::to_string(k));
    T->topic0 = sig_hash[0];
    T->topic1 = ascii32("calib-subnet-1");

    // ---- messages → per-parent-block BLS / secp lists → execution order ----
    const uint32_t P = p.n_parents ? p.n_parents : 1;
    std::vector<Cid> uniq(p.n_receipts);
    for (auto& c : uniq) c = fake_cid(r);
    std::vector<std::vector<Cid>> bls(P), secp(P);
    {
        uint64_t next = 0;
        for (uint32_t b = 0; b < P; ++b) {
            const uint64_t take = (b + 1 == P) ? p.n_receipts - next : p.n_receipts / P;
            const uint64_t nb = take * 7 / 10;
            for (uint64_t k = 0; k < take; ++k) {
                auto& dst = k < nb ? bls[b] : secp[b];
                if (b > 0 && p.dup_permille && r.below(1000) < p.dup_permille && next > 0) dst.push_back(uniq[r.below(next)]);
                dst.push_back(uniq[next + k]);
            }
            next += take;
        }
        // first-seen order == uniq order by construction (duplicates always point backwards)
        T->exec_order = uniq;
    }
    std::vector<Cid> txmeta_cids(P);
    for (uint32_t b = 0; b < P; ++b) {
        const uint64_t before = sink.payload;
        Cid roots[2];
        for (int side = 0; side < 2; ++side) {
            const auto& lst = side == 0 ? bls[b] : secp[b];
            std::vector<AmtVal> vals(lst.size());
            for (size_t i = 0; i < lst.size(); ++i) { W w; w.link(lst[i]); vals[i] = {i, std::move(w.b)}; }
            roots[side] = build_amt(sink, 0, 3, vals);
        }
        W tx; tx.array(2); tx.link(roots[0]); tx.link(roots[1]);
        txmeta_cids[b] = sink.add(tx.b);
        T->msg_amt_bytes += sink.payload - before;
    }

    // ---- events AMTs + receipts AMT ----
    std::vector<uint8_t> planted_flag(p.n_receipts, 0);
    for (uint64_t k = 0; k < p.n_planted && p.n_receipts; ++k) planted_flag[r.below(p.n_receipts)] = 1;
    std::vector<AmtVal> receipt_vals(p.n_receipts);
    const uint32_t ebw = p.events_bit_width ? p.events_bit_width : 5;
    for (uint64_t i = 0; i < p.n_receipts; ++i) {
        bool has_events = p.max_events > 0 && !(p.no_events_permille && r.below(1000) < p.no_events_permille);
        if (planted_flag[i]) has_events = p.max_events > 0;
        Cid events_root{};
        if (has_events) {
            const uint32_t ne = 1 + uint32_t(r.below(p.max_events));
            const uint32_t plant_at = planted_flag[i] ? uint32_t(r.below(ne)) : ne;
            const uint32_t claim_at = uint32_t(r.below(ne));
            std::vector<AmtVal> evs(ne);
            const uint64_t before = sink.payload;
            for (uint32_t j = 0; j < ne; ++j) {
                W w;
                EventRec rec{};
                rec.exec_index = i;
                rec.event_index = j;
                if (j == plant_at) {
                    encode_event(w, r, p.filter_actor, T->topic0, T->topic1, 0, rec);
                } else {
                    const uint64_t em = p.emitter_base + r.below(p.n_emitters ? p.n_emitters : 1);
                    const auto& t0 = sig_hash[r.below(n_sigs)];
                    const auto& t1 = subnets[r.below(n_sub)];
                    const uint32_t shape = p.variety ? uint32_t(r.below(16)) : 0;  // shapes 8..15 fold to 0
                    encode_event(w, r, em, t0, t1, shape < 8 ? shape : 0, rec);
                }
                evs[j] = {j, std::move(w.b)};
                if (j == claim_at) T->claims.push_back(rec);
            }
            if (planted_flag[i]) T->planted.push_back(i);
            events_root = build_amt(sink, 3, ebw, evs);
            T->events_amt_bytes += sink.payload - before;
        }
        W w;
        uint8_t ret[8];
        r.fill(ret, 8);
        w.array(4);
        w.uint(r.below(50) == 0 ? 1 + r.below(30) : 0);  // exit code
        w.bytes(ret, r.below(4) == 0 ? 8 : 0);            // return data
        w.uint(r.below(1ull << 24));                       // gas used
        if (has_events) w.link(events_root); else w.null();
        receipt_vals[i] = {i, std::move(w.b)};
    }
    {
        const uint64_t before = sink.payload;
        T->receipts_root = build_amt(sink, 0, 3, receipt_vals);
        T->receipts_amt_bytes = sink.payload - before;
    }

    // ---- state tree ----
    Cid state_root = fake_cid(r);
    if (p.n_actors) {
        const uint64_t before = sink.payload;
        // code CIDs: 8 identity-multihash raw CIDs like the builtin-actors manifest uses
        std::vector<Bytes> code_cids;
        for (int k = 0; k < 8; ++k) {
            std::string name = "fil/12/actor" + std::to_string(k);
            Bytes c{0x01, 0x55, 0x00, uint8_t(name.size())};
            c.insert(c.end(), name.begin(), name.end());
            code_cids.push_back(c);
        }
        // queries first (their paths decide what is kept)
        std::vector<std::array<uint8_t, 32>> wanted;
        auto want_key = [&](const Bytes& k) { std::array<uint8_t, 32> h; sha256(k.data(), k.size(), h.data()); wanted.push_back(h); };
        for (uint64_t q = 0; q < p.n_actor_queries; ++q) {
            const bool absent = r.below(100) == 0;
            const uint64_t id = absent ? p.actor_base + p.n_actors + r.below(1000000) : p.actor_base + r.below(p.n_actors);
            T->query_ids.push_back(id);
            T->query_present.push_back(absent ? 0 : 1);
            want_key(id_addr(id));
        }
        for (uint32_t c = 0; c < p.n_contracts && c < p.n_actors; ++c) want_key(id_addr(p.actor_base + c));
        std::sort(wanted.begin(), wanted.end());
        HamtKeep keep;
        keep.all = p.keep_full_state != 0;
        keep.wanted = &wanted;

        std::vector<HEntry> actors(p.n_actors);
        for (uint64_t a = 0; a < p.n_actors; ++a) {
            const uint64_t id = p.actor_base + a;
            Cid actor_state = fake_cid(r);
            if (a < p.n_contracts) {
                // storage: slots key = keccak(ascii32("s<k>") ‖ u256(0)), value = 1..32 PRNG bytes as Vec<u8> (CBOR array of u8)
                const uint32_t layout = p.storage_layout_mix ? uint32_t(a % 6) : 0;
                std::vector<HEntry> slots(p.slots_per_contract);
                std::vector<std::pair<Bytes, Bytes>> inline_pairs;
                for (uint32_t k = 0; k < p.slots_per_contract; ++k) {
                    uint8_t pre[64] = {0};
                    auto key32 = ascii32("s" + std::to_string(k));
                    std::memcpy(pre, key32.data(), 32);
                    std::array<uint8_t, 32> slot;
                    keccak256(pre, 64, slot.data());
                    uint32_t vl = 1 + uint32_t(r.below(32));
                    if (r.below(16) == 0) vl = 33 + uint32_t(r.below(8));  // longer than 32: left_pad_32 keeps the LAST 32
                    Bytes v(vl);
                    r.fill(v.data(), vl);
                    if (r.below(4) == 0) v[0] = 0;  // leading zero byte kept as stored
                    W vw;
                    vw.array(vl);
                    for (uint8_t x : v) vw.uint(x);
                    slots[k].key.assign(slot.begin(), slot.end());
                    slots[k].val = vw.b;
                    inline_pairs.emplace_back(slots[k].key, v);
                    Tipset::StorageClaim sc{};
                    sc.actor_id = id;
                    sc.slot = slot;
                    sc.present = 1;
                    if (vl >= 32) std::memcpy(sc.value.data(), v.data() + vl - 32, 32);
                    else { sc.value.fill(0); std::memcpy(sc.value.data() + 32 - vl, v.data(), vl); }
                    T->storage_claims.push_back(sc);
                }
                {   // one absent slot per contract: value reads as zero
                    Tipset::StorageClaim sc{};
                    sc.actor_id = id;
                    r.fill(sc.slot.data(), 32);
                    sc.value.fill(0);
                    sc.present = 0;
                    T->storage_claims.push_back(sc);
                }
                HamtKeep all;  // contract storage is small: keep every node
                Cid storage_root;
                auto small_map = [&](W& w) {  // { "v": [[k, v]…] }
                    w.map(1); w.text("v"); w.array(inline_pairs.size());
                    for (auto& kv : inline_pairs) { w.array(2); w.bytes(kv.first); w.bytes(kv.second); }
                };
                const uint8_t params[3] = {1, 2, 3};
                switch (layout) {
                    default:
                    case 0: storage_root = build_hamt(sink, 5, slots, all); break;                    // C: direct HAMT, bw 5
                    case 1: { Cid inner = build_hamt(sink, 5, slots, all); W w; w.array(2); w.link(inner); w.uint(5); storage_root = sink.add(w.b); break; }  // B1
                    case 2: { Cid inner = build_hamt(sink, 3, slots, all); W w; w.map(3); w.text("root"); w.link(inner); w.text("bitwidth"); w.uint(3); w.text("extra"); w.uint(7); storage_root = sink.add(w.b); break; }  // B2
                    case 3: { W w; w.array(2); w.bytes(params, 3); w.array(1); small_map(w); storage_root = sink.add(w.b); break; }  // A1
                    case 4: { W w; w.array(2); w.bytes(params, 3); small_map(w); storage_root = sink.add(w.b); break; }               // A2
                    case 5: { W w; small_map(w); storage_root = sink.add(w.b); break; }                                               // A3
                }
                // EvmState: V6 for even contracts, V5 for odd
                W ev;
                uint8_t bh[32];
                r.fill(bh, 32);
                if (a % 2 == 0) { ev.array(6); ev.link(fake_cid(r)); ev.bytes(bh, 32); ev.link(storage_root); ev.null(); ev.uint(1 + r.below(100)); ev.null(); }
                else { ev.array(5); ev.link(fake_cid(r)); ev.bytes(bh, 32); ev.link(storage_root); ev.uint(1 + r.below(100)); ev.null(); }
                actor_state = sink.add(ev.b);
                for (size_t k = T->storage_claims.size() - p.slots_per_contract - 1; k < T->storage_claims.size(); ++k) {
                    T->storage_claims[k].actor_state = actor_state;
                    T->storage_claims[k].storage_root = storage_root;
                }
            }
            W w;
            w.array(5);
            const Bytes& code = code_cids[a < p.n_contracts ? 0 : 1 + r.below(7)];
            w.link(code.data(), code.size());
            w.link(actor_state);
            w.uint(r.below(1000));
            { uint8_t bal[9] = {0}; const size_t bl = r.below(9); r.fill(bal + 1, 8); w.bytes(bal, bl ? bl + 1 : 0); }
            if (r.below(8) == 0) { uint8_t da[22] = {4, 10}; r.fill(da + 2, 20); w.bytes(da, 22); } else w.null();
            actors[a].key = id_addr(id);
            actors[a].val = std::move(w.b);
        }
        T->actors_root = build_hamt(sink, 5, actors, keep);
        W sr;
        sr.array(3); sr.uint(5); sr.link(T->actors_root); sr.link(fake_cid(r));
        state_root = sink.add(sr.b);
        T->state_bytes = sink.payload - before;
    }
    T->state_root_cid = state_root;

    // ---- headers ----
    std::vector<Cid> grandparents{fake_cid(r), fake_cid(r)};
    for (uint32_t b = 0; b < P; ++b) {
        Bytes h = header_block(r, grandparents, p.parent_epoch, fake_cid(r), fake_cid(r), txmeta_cids[b]);
        T->parent_cids.push_back(sink.add(h));
    }
    Bytes ch = header_block(r, T->parent_cids, p.parent_epoch + 1, state_root, T->receipts_root, fake_cid(r));
    T->child_cid = sink.add(ch);
    return T;
}

}  // namespace synth

using namespace synth;

extern "C" {

void* synth_build(const synth_params* p) { return build(*p); }
void synth_free(void* t) { delete static_cast<Tipset*>(t); }

// witness tables
uint64_t synth_block_count(void* t) { return static_cast<Tipset*>(t)->sink.off.size(); }
uint64_t synth_byte_count(void* t) { return static_cast<Tipset*>(t)->sink.bytes.size(); }
const uint8_t* synth_bytes(void* t) { return static_cast<Tipset*>(t)->sink.bytes.data(); }
const uint64_t* synth_off(void* t) { return static_cast<Tipset*>(t)->sink.off.data(); }
const uint32_t* synth_len(void* t) { return static_cast<Tipset*>(t)->sink.len.data(); }
const uint8_t* synth_cids(void* t) { return static_cast<Tipset*>(t)->sink.cids.data(); }

// what[]: 0 parent count, 1 exec-order length, 2 claims, 3 planted, 4 actor queries, 5 storage claims,
//         6 receipts-AMT bytes, 7 events-AMT bytes, 8 message-AMT bytes, 9 state bytes, 10 payload bytes
uint64_t synth_count(void* t, int what) {
    Tipset* T = static_cast<Tipset*>(t);
    switch (what) {
        case 0: return T->parent_cids.size();
        case 1: return T->exec_order.size();
        case 2: return T->claims.size();
        case 3: return T->planted.size();
        case 4: return T->query_ids.size();
        case 5: return T->storage_claims.size();
        case 6: return T->receipts_amt_bytes;
        case 7: return T->events_amt_bytes;
        case 8: return T->msg_amt_bytes;
        case 9: return T->state_bytes;
        case 10: return T->sink.payload;
        default: return 0;
    }
}

// named CIDs: 0 child header, 1 receipts root, 2 state root, 3 actors HAMT root; 100+k parent header k
void synth_cid(void* t, int which, uint8_t out[40]) {
    Tipset* T = static_cast<Tipset*>(t);
    std::memset(out, 0, 40);
    const Cid* c = nullptr;
    if (which == 0) c = &T->child_cid;
    else if (which == 1) c = &T->receipts_root;
    else if (which == 2) c = &T->state_root_cid;
    else if (which == 3) c = &T->actors_root;
    else if (which >= 100 && size_t(which - 100) < T->parent_cids.size()) c = &T->parent_cids[which - 100];
    if (c) std::memcpy(out, c->data(), 38);
}

void synth_filter(void* t, uint8_t topic0[32], uint8_t topic1[32]) {
    Tipset* T = static_cast<Tipset*>(t);
    std::memcpy(topic0, T->topic0.data(), 32);
    std::memcpy(topic1, T->topic1.data(), 32);
}

void synth_exec_order(void* t, uint8_t* out40) {
    Tipset* T = static_cast<Tipset*>(t);
    for (size_t i = 0; i < T->exec_order.size(); ++i) {
        std::memset(out40 + 40 * i, 0, 40);
        std::memcpy(out40 + 40 * i, T->exec_order[i].data(), 38);
    }
}

void synth_planted(void* t, uint64_t* out) {
    Tipset* T = static_cast<Tipset*>(t);
    std::copy(T->planted.begin(), T->planted.end(), out);
}

// event claims (SoA): exec_index, event_index, emitter, n_topics, topics[4*32], data_len, data[64]
void synth_event_claims(void* t, uint64_t* exec_index, uint64_t* event_index, uint64_t* emitter, uint32_t* n_topics,
                        uint8_t* topics128, uint32_t* data_len, uint8_t* data64) {
    Tipset* T = static_cast<Tipset*>(t);
    for (size_t i = 0; i < T->claims.size(); ++i) {
        const EventRec& c = T->claims[i];
        exec_index[i] = c.exec_index; event_index[i] = c.event_index; emitter[i] = c.emitter;
        n_topics[i] = c.n_topics; data_len[i] = c.data_len;
        std::memcpy(topics128 + 128 * i, c.topics.data(), 128);
        std::memcpy(data64 + 64 * i, c.data.data(), 64);
    }
}

void synth_actor_queries(void* t, uint64_t* ids, uint8_t* present) {
    Tipset* T = static_cast<Tipset*>(t);
    std::copy(T->query_ids.begin(), T->query_ids.end(), ids);
    std::copy(T->query_present.begin(), T->query_present.end(), present);
}

void synth_storage_claims(void* t, uint64_t* actor_id, uint8_t* actor_state40, uint8_t* storage_root40, uint8_t* slot32,
                          uint8_t* value32, uint8_t* present) {
    Tipset* T = static_cast<Tipset*>(t);
    for (size_t i = 0; i < T->storage_claims.size(); ++i) {
        const auto& c = T->storage_claims[i];
        actor_id[i] = c.actor_id;
        std::memset(actor_state40 + 40 * i, 0, 40); std::memcpy(actor_state40 + 40 * i, c.actor_state.data(), 38);
        std::memset(storage_root40 + 40 * i, 0, 40); std::memcpy(storage_root40 + 40 * i, c.storage_root.data(), 38);
        std::memcpy(slot32 + 32 * i, c.slot.data(), 32);
        std::memcpy(value32 + 32 * i, c.value.data(), 32);
        present[i] = c.present;
    }
}

}  // extern "C"
