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

// Decode one StampedEvent `[emitter, [[flags, key, codec, value]…]]` located at r (already
// type-checked by the AMT walk) and extract the EVM log view.  Offsets are relative to r.p.
__device__ __forceinline__ void decode_event_log(Rd& r, uint64_t& emitter, EvmLogLoc& log) {
    ByteRange topics{0, 0, false}, data{0, 0, false}, d{0, 0, false};
    ByteRange t[4] = {{0, 0, false}, {0, 0, false}, {0, 0, false}, {0, 0, false}};
    r.expect_array(2);
    emitter = r.read_uint();
    const uint64_t ne = r.read_array();
    for (uint64_t i = 0; i < ne && r.ok(); ++i) {
        uint32_t ko, kl, vo, vl;
        r.expect_array(4);
        (void)r.read_uint();
        r.read_text(ko, kl);
        (void)r.read_uint();
        r.read_bytes(vo, vl);
        if (!r.ok()) break;
        // keys that matter are at most 6 bytes: fetch them once
        uint32_t k[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) k[q] = uint32_t(q) < kl ? r.at(ko + q) : 0u;
        const ByteRange v{vo, vl, true};
        if (kl == 1 && k[0] == 'd') d = v;
        else if (kl == 2 && k[0] == 't' && k[1] >= '1' && k[1] <= '4') {
            // t[k[1]-'1'] without dynamic register indexing
            const uint32_t which = k[1] - '1';
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (which == uint32_t(q)) t[q] = v;
        } else if (kl == 4 && k[0] == 'd' && k[1] == 'a' && k[2] == 't' && k[3] == 'a') data = v;
        else if (kl == 6 && k[0] == 't' && k[1] == 'o' && k[2] == 'p' && k[3] == 'i' && k[4] == 'c' && k[5] == 's') topics = v;
    }
    log.is_log = false;
    log.case_a = false;
    log.n_topics = 0;
    log.data = ByteRange{0, 0, false};
#pragma unroll
    for (int q = 0; q < 4; ++q) log.topic_off[q] = 0;
    if (!r.ok()) return;
    if (topics.present) {  // Case A (evm.rs:19-30): wins whenever the key exists, even when empty
        if (topics.len % 32u != 0) return;
        log.is_log = true;
        log.case_a = true;
        log.n_topics = topics.len / 32u;
        log.topic_off[0] = topics.off;
        if (data.present) log.data = data;
        return;
    }
    // Case B (evm.rs:32-58): t1, t2, … until the first missing one; every present one must be 32 bytes
    uint32_t n = 0;
    bool bad = false, stop = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!stop) {
            if (!t[q].present) stop = true;
            else if (t[q].len != 32) { bad = true; stop = true; }
            else { log.topic_off[q] = t[q].off; n = uint32_t(q) + 1; }
        }
    }
    if (bad || n == 0) return;
    log.is_log = true;
    log.n_topics = n;
    if (d.present) log.data = d;
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
