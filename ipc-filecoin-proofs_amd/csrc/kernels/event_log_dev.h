// csrc/kernels/event_log_dev.h — StampedEvent decode, `extract_evm_log` and `matches_log` on the device.
//
// Replaces (per event, no heap, no HashMap):
//   extract_evm_log            src/proofs/common/evm.rs:13-59
//   EventMatcher::matches_log  src/proofs/events/generator.rs:38-40
//   create_event_filter        src/proofs/events/verifier.rs:28-39
// The reference builds a `HashMap<&str, &[u8]>` of the entries (a repeated key keeps the LAST
// value) and then looks up "topics"/"data" (Case A) or "t1".."t4"/"d" (Case B).  Only those seven
// keys matter, so the device keeps the location of the last occurrence of each in registers.
#pragma once
#include "cbor_dev.h"

namespace ipcfp {

struct ByteRange {
    uint32_t off, len;
    bool present;
};

struct EvmLogLoc {
    bool is_log;        // extract_evm_log returned Some
    bool case_a;        // topics come from one concatenated "topics" value
    uint32_t n_topics;
    uint32_t topic_off[4];  // Case B: offsets of t1..t4 values; Case A: topic_off[0] = start of the concatenation
    ByteRange data;
    __device__ __forceinline__ uint32_t topic_at(uint32_t i) const {
        // masks, not topic_off[i]: a dynamically indexed member would push the struct to scratch
        const uint32_t o = (topic_off[0] & (i == 0 ? ~0u : 0u)) | (topic_off[1] & (i == 1 ? ~0u : 0u)) |
                           (topic_off[2] & (i == 2 ? ~0u : 0u)) | (topic_off[3] & (i == 3 ? ~0u : 0u));
        return case_a ? topic_off[0] + 32u * i : o;
    }
};

// One entry `[flags, key, codec, value]` in the shape FVM writes it, decoded from the 16 bytes at the reader's
// position without a branch: 0x84, a one-byte flags uint, a text key of at most 6 ASCII bytes, a codec uint with an
// immediate or one-byte argument, a byte string with a 1-3 byte header.  All of it is what the general decode below
// accepts for these very bytes (same offsets, same lengths); any other spelling — a non-minimal integer, a longer or
// non-ASCII key, a value that overruns the item — returns false with the reader untouched and the general decode
// takes the entry (and names the error, if it is one).  An entry is five CBOR headers; decoded one by one they are
// ≈400 instructions and two dozen branches per entry, which is what the event kernels were spending their time on.
__device__ __forceinline__ bool decode_entry_fast(Rd& r, uint32_t& ko, uint32_t& kl, uint32_t& vo, uint32_t& vl) {
    const uint32_t p0 = r.pos;
    if (r.err || r.n - p0 < 6u || p0 > r.n) return false;  // (the shortest entry is 84 00 60 00 40)
    uint64_t w0, w1;
    r.peek128(p0, w0, w1);
    const uint32_t b0 = uint32_t(w0) & 0xffu, b1 = uint32_t(w0 >> 8) & 0xffu, b2 = uint32_t(w0 >> 16) & 0xffu;
    const uint32_t klen = b2 - 0x60u;                        // key length when b2 is a short text header
    bool ok = b0 == 0x84u && b1 < 0x18u && klen <= 6u;
    const uint32_t kq = klen <= 6u ? klen : 0u;
    const uint64_t kbytes = (w0 >> 24) | (w1 << 40);         // bytes 3..10
    const uint64_t kmask = (1ull << (8u * kq)) - 1ull;
    ok = ok && (kbytes & kmask & 0x8080808080808080ull) == 0;  // ASCII ⇒ valid UTF-8
    const uint32_t c = 3u + kq;                              // offset of the codec header: 3..9
    const uint32_t c_lo = c < 8u ? c : 7u, c_hi = c < 8u ? 0u : c - 8u;  // (both arms are computed: keep the shifts defined)
    const uint64_t rest = c < 8u ? ((w0 >> (8u * c_lo)) | ((w1 << 1) << (63u - 8u * c_lo))) : (w1 >> (8u * c_hi));  // bytes c..
    const uint32_t r0 = uint32_t(rest) & 0xffu;
    const uint32_t cl = r0 < 0x18u ? 1u : 2u;
    ok = ok && r0 <= 0x18u;
    const uint64_t v = rest >> (8u * cl);
    const uint32_t vb = uint32_t(v) & 0xffu;
    const bool v_imm = vb >= 0x40u && vb <= 0x57u, v_1 = vb == 0x58u, v_2 = vb == 0x59u;
    ok = ok && (v_imm || v_1 || v_2);
    const uint32_t hl = v_imm ? 1u : (v_1 ? 2u : 3u);
    const uint32_t len1 = uint32_t(v >> 8) & 0xffu;
    const uint32_t len2 = (len1 << 8) | (uint32_t(v >> 16) & 0xffu);  // big-endian u16
    const uint32_t len = v_imm ? vb - 0x40u : (v_1 ? len1 : len2);
    const uint32_t value_at = p0 + c + cl + hl;
    ok = ok && value_at <= r.n && len <= r.n - value_at;
    if (!ok) return false;
    ko = p0 + 3u;
    kl = kq;
    vo = value_at;
    vl = len;
    r.pos = value_at + len;
    return true;
}

// 16 bytes as two little-endian words → the 8 bytes that start at byte o (o ≤ 8; beyond byte 15: zeros)
__device__ __forceinline__ uint64_t bytes_from(uint64_t w0, uint64_t w1, uint32_t o) {
    const uint32_t sh = 8u * (o & 7u);
    const uint64_t lo = o < 8u ? w0 : w1, hi = o < 8u ? w1 : 0ull;
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}

// The head of a StampedEvent — `82`, the emitter (a uint in any of its five widths), the header of the entries array
// (up to 255 entries) — from the 16 bytes at the reader's position, without a branch: what expect_array(2) / read_uint /
// read_array accept for these very bytes, with the same values and the same position afterwards.  Any other spelling
// (an array header with a 2-8 byte count, an item that runs off the block, a reader that has failed) returns false
// with the reader untouched and the three general decodes take it — three item headers are ≈ 200 instructions.
__device__ __forceinline__ bool decode_event_head_fast(Rd& r, uint64_t& emitter, uint64_t& n_entries) {
    const uint32_t p0 = r.pos;
    if (r.err || p0 > r.n || r.n - p0 < 3u) return false;
    uint64_t w0, w1;
    r.peek128(p0, w0, w1);
    const uint32_t b0 = uint32_t(w0) & 0xffu, b1 = uint32_t(w0 >> 8) & 0xffu;
    const uint32_t ai = b1 & 31u;
    bool ok = b0 == 0x82u && (b1 >> 5) == 0u && ai <= 27u;
    const uint32_t nb = ai < 24u ? 0u : (1u << ((ai - 24u) & 3u));            // 0, 1, 2, 4, 8 argument bytes
    const uint64_t arg = __builtin_bswap64(bytes_from(w0, w1, 2u));            // bytes 2..9, big-endian
    const uint64_t em = ai < 24u ? uint64_t(ai) : (nb == 8u ? arg : (arg >> ((64u - 8u * nb) & 63u)));
    const uint32_t ho = 2u + nb;                                               // the entries array's header: byte 2..10
    const uint64_t hw = bytes_from(w0, w1, ho > 8u ? 8u : ho) >> (ho > 8u ? 8u * (ho - 8u) : 0u);
    const uint32_t h = uint32_t(hw) & 0xffu;
    const bool h_imm = (h >> 5) == 4u && (h & 31u) < 24u, h_1 = h == 0x98u;
    ok = ok && (h_imm || h_1);
    const uint32_t total = ho + (h_imm ? 1u : 2u);
    ok = ok && total <= r.n - p0;
    if (!ok) return false;
    emitter = em;
    n_entries = h_imm ? uint64_t(h & 31u) : uint64_t(uint32_t(hw >> 8) & 0xffu);
    r.pos = p0 + total;
    return true;
}

// Decode one StampedEvent `[emitter, [[flags, key, codec, value]…]]` located at r (already
// type-checked by the AMT walk) and extract the EVM log view.  Offsets are relative to r.p.
//
// The last occurrence of each of the seven keys is kept as SCALARS (an offset and a length per key, a bit per key in
// `have`) and every update is a select on a scalar.  Round 1-4 kept them as seven {off, len, present} structs updated by
// `slot = is_key ? v : slot`: hipcc turns a select between two structs into a load from a SELECTED ADDRESS, so all seven
// lived in scratch memory and every entry of every event cost 14 scratch loads and 7 scratch stores, each behind its
// own s_waitcnt vmcnt(0) — the "121 load instructions per wavefront" and the 45 % of wave cycles spent waiting that
// profiles/r04_final_pmc.txt shows for k_block_events were mostly these, not the reader's line refills.
__device__ __forceinline__ void decode_event_log(Rd& r, uint64_t& emitter, EvmLogLoc& log) {
    // key slots: 0..3 = "t1".."t4", 4 = "d", 5 = "data", 6 = "topics"
    uint32_t have = 0;
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o_d = 0, o_data = 0, o_topics = 0;
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0, l_d = 0, l_data = 0, l_topics = 0;
    uint64_t ne;
    if (!decode_event_head_fast(r, emitter, ne)) {
        r.expect_array(2);
        emitter = r.read_uint();
        ne = r.read_array();
    }
    for (uint64_t i = 0; i < ne && r.ok(); ++i) {
        uint32_t ko, kl, vo, vl;
        if (!decode_entry_fast(r, ko, kl, vo, vl)) {
            r.expect_array(4);
            (void)r.read_uint();
            r.read_text(ko, kl);
            (void)r.read_uint();
            r.read_bytes(vo, vl);
            if (!r.ok()) break;
        }
        // keys that matter are at most 6 bytes: ONE fetch, compared as little-endian words; every outcome is a
        // select (the lanes of a wavefront sit on different keys: a branch per key would run them all)
        const uint64_t kw = kl <= 6 ? (r.peek64(ko) & ((1ull << (8u * (kl & 7u))) - 1ull)) : 0ull;
        const uint32_t digit = uint32_t(kw >> 8) & 0xffu;
        const bool is_d = kl == 1 && kw == 0x64ull;                                   // "d"
        const bool is_data = kl == 4 && kw == 0x61746164ull;                          // "data"
        const bool is_topics = kl == 6 && kw == 0x736369706f74ull;                    // "topics"
        const bool is_t = kl == 2 && (kw & 0xffull) == 0x74ull && digit >= '1' && digit <= '4';  // "t1".."t4"
        const uint32_t bit = is_t ? (1u << ((digit - uint32_t('1')) & 3u)) : (is_d ? 16u : (is_data ? 32u : (is_topics ? 64u : 0u)));
        have |= bit;
        o0 = (bit & 1u) ? vo : o0;
        l0 = (bit & 1u) ? vl : l0;
        o1 = (bit & 2u) ? vo : o1;
        l1 = (bit & 2u) ? vl : l1;
        o2 = (bit & 4u) ? vo : o2;
        l2 = (bit & 4u) ? vl : l2;
        o3 = (bit & 8u) ? vo : o3;
        l3 = (bit & 8u) ? vl : l3;
        o_d = (bit & 16u) ? vo : o_d;
        l_d = (bit & 16u) ? vl : l_d;
        o_data = (bit & 32u) ? vo : o_data;
        l_data = (bit & 32u) ? vl : l_data;
        o_topics = (bit & 64u) ? vo : o_topics;
        l_topics = (bit & 64u) ? vl : l_topics;
    }
    log.is_log = false;
    log.case_a = false;
    log.n_topics = 0;
    log.data = ByteRange{0, 0, false};
#pragma unroll
    for (int q = 0; q < 4; ++q) log.topic_off[q] = 0;
    if (!r.ok()) return;
    if (have & 64u) {  // Case A (evm.rs:19-30): wins whenever the key exists, even when empty
        if (l_topics % 32u != 0) return;
        log.is_log = true;
        log.case_a = true;
        log.n_topics = l_topics / 32u;
        log.topic_off[0] = o_topics;
        if (have & 32u) log.data = ByteRange{o_data, l_data, true};
        return;
    }
    // Case B (evm.rs:32-58): t1, t2, … until the first missing one; every present one must be 32 bytes.
    // g_q: t(q+1) is there with 32 bytes.  n = the run of good ones from t1; the key that ends the run spoils the log
    // when it is there (then its length is wrong), and ends it quietly when it is missing.
    const bool g0 = (have & 1u) && l0 == 32u, g1 = (have & 2u) && l1 == 32u, g2 = (have & 4u) && l2 == 32u,
               g3 = (have & 8u) && l3 == 32u;
    const uint32_t n = !g0 ? 0u : (!g1 ? 1u : (!g2 ? 2u : (!g3 ? 3u : 4u)));
    const bool bad = n < 4u && ((have >> n) & 1u) != 0u;
    if (bad || n == 0u) return;
    log.is_log = true;
    log.n_topics = n;
    log.topic_off[0] = o0;
    log.topic_off[1] = n > 1u ? o1 : 0u;
    log.topic_off[2] = n > 2u ? o2 : 0u;
    log.topic_off[3] = n > 3u ? o3 : 0u;
    if (have & 16u) log.data = ByteRange{o_d, l_d, true};
}

__device__ __forceinline__ bool bytes32_equal(const uint8_t* a, const uint8_t* b) {
    bool eq = true;
    for (int i = 0; i < 32; ++i) eq &= a[i] == b[i];
    return eq;
}

// matches_log / create_event_filter: topics.len() >= 2 && topics[0] == topic0 && topics[1] == topic1
__device__ __forceinline__ bool log_matches(Rd& r, const EvmLogLoc& log, const ipcfp_event_filter_t& f) {
    if (!log.is_log || log.n_topics < 2) return false;
    return r.equal32(log.topic_at(0), f.topic0) && r.equal32(log.topic_at(1), f.topic1);
}

}  // namespace ipcfp
