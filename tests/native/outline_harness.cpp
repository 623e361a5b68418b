// tests/native/outline_harness.cpp — TEST INFRASTRUCTURE.  The outline of a state-tree HAMT node (csrc/kernels/hamt_outline.h)
// on the host: the sequential reader, and the PARALLEL outline with its 32 lanes simulated in lock step, phase by phase,
// exactly as k_hamt_lv_parse_actor (csrc/kernels/hamt_levels.hip) runs them.  tests/test_hamt_outline.py holds the two
// against each other on honest, mutated and adversarial nodes: whatever the parallel outline accepts, the sequential one
// accepts with the same record.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../ipc-filecoin-proofs_amd/csrc/kernels/hamt_outline.h"

using namespace ipcfp;

namespace {
constexpr uint32_t kLanes = 32, kStage = 6912;

struct Stage {
    alignas(16) uint8_t bytes[kStage + 512];
    Stage(const uint8_t* node, uint32_t len, uint32_t garbage_seed) {
        uint32_t x = garbage_seed * 2654435761u + 12345u;  // what lies behind the node in LDS is somebody else's bytes
        for (auto& b : bytes) {
            x = x * 1664525u + 1013904223u;
            b = uint8_t(x >> 24);
        }
        std::memcpy(bytes, node, len);
    }
};
}  // namespace

extern "C" {

// out: np, ne, links, bf low, bf high.  Returns 1 when the node has this outline.
// endv / klen / first_of: the entry table's extras (where entry e ends, its key's length, a bucket pointer's first entry)
int outline_seq(const uint8_t* node, uint32_t len, uint32_t seed, uint32_t* out, uint16_t* ptr, uint16_t* val, uint16_t* l2, uint16_t* adr,
                uint16_t* endv, uint8_t* klen, uint8_t* first_of) {
    if (len < 3 || len + 24 > kStage) return 0;
    Stage st(node, len, seed);
    outline::Result r{0, 0, 0, 0};
    const bool ok = outline::outline_sequential(st.bytes, len, r, ptr, val, l2, adr, outline::kMaxEntries, endv, klen, first_of);
    out[0] = r.np;
    out[1] = r.ne;
    out[2] = r.links;
    out[3] = uint32_t(r.bf);
    out[4] = uint32_t(r.bf >> 32);
    return ok ? 1 : 0;
}

// the parallel outline alone (no fallback): 1 = accepted.  *n_anchors: how many `85 d8 2a` were found.
int outline_par(const uint8_t* node, uint32_t len, uint32_t seed, uint32_t* out, uint16_t* ptr, uint16_t* val, uint16_t* l2, uint16_t* adr,
                uint32_t* n_anchors, uint16_t* endv, uint8_t* klen, uint8_t* first_of) {
    *n_anchors = 0;
    out[5] = 0;  // which check declined (bits: 1 entry, 2 gap, 4 pointer count, 8 bucket hops) — diagnostics of the tests
    if (len < 3 || len + 24 > kStage) return 0;
    Stage st(node, len, seed);
    const uint8_t* S = st.bytes;
    const outline::Header hd = outline::header(S, len);
    if (!hd.ok) return 0;
    // phase 1: anchors, a contiguous share of the node per lane
    const uint32_t span = (((len + kLanes - 1u) / kLanes) + 7u) & ~7u;
    const uint32_t scan_end = len >= 2u ? len - 2u : 0u;
    uint32_t cnt[kLanes], from[kLanes], to[kLanes], na = 0;
    uint64_t packed[kLanes];
    bool crowded = false;  // (a lane with more than four anchors: the group scans again, writing)
    for (uint32_t sub = 0; sub < kLanes; ++sub) {
        from[sub] = sub * span > hd.pos0 ? sub * span : hd.pos0;
        to[sub] = (sub + 1u) * span < scan_end ? (sub + 1u) * span : scan_end;
        packed[sub] = 0;
        cnt[sub] = from[sub] < to[sub] ? outline::scan_anchors_packed(S, from[sub], to[sub], packed[sub]) : 0u;
        crowded = crowded || cnt[sub] > 4u;
        na += cnt[sub];
    }
    *n_anchors = na;
    if (na > outline::kMaxEntries) return 0;
    uint16_t end[outline::kMaxEntries + 1] = {};
    uint8_t gn[outline::kMaxEntries + 4] = {}, gc[outline::kMaxEntries + 4] = {}, gb[outline::kMaxEntries + 4] = {};
    {
        uint32_t at = 0;
        for (uint32_t sub = 0; sub < kLanes; ++sub) {
            if (cnt[sub] && crowded) (void)outline::scan_anchors<true>(S, from[sub], to[sub], val + at, cnt[sub]);
            else
                for (uint32_t k = 0; k < cnt[sub]; ++k) val[at + k] = uint16_t(packed[sub] >> (16u * k));
            at += cnt[sub];
        }
    }
    bool ok_all = true;
    // phase 2: entries forward; an anchor that does not parse is dropped (hamt_levels.hip), the kept ones move down
    {
        uint32_t kept = 0;
        for (uint32_t e = 0; e < na; ++e) {
            uint32_t a2 = 0, ad = 0, en = 0;
            const uint32_t a = val[e];
            if (!outline::entry_forward(S, a, len, a2, ad, en)) continue;
            val[kept] = uint16_t(a);
            l2[kept] = uint16_t(a2);
            adr[kept] = uint16_t(ad);
            end[kept] = uint16_t(en);
            ++kept;
        }
        na = kept;
    }
    // phase 3: gaps
    for (uint32_t e = 0; e <= na; ++e) {
        uint32_t n_ptr = 0, count = 0, kl = 0;
        const uint32_t f = e ? uint32_t(end[e - 1]) : hd.pos0, target = e < na ? uint32_t(val[e]) : len;
        const bool ok = outline::gap_walk(S, f, target, len, e == na, e == 0u, n_ptr, count, nullptr, 0u, nullptr, &kl);
        if (e < na) klen[e] = uint8_t(kl);
        ok_all = ok_all && ok;
        if (!ok) out[5] |= 2u;
        gn[e] = uint8_t(ok ? n_ptr : 0u);
        gc[e] = uint8_t(ok ? count : 0u);
    }
    // phase 4: pointer numbers, bucket hops
    uint32_t carry = 0;
    for (uint32_t e = 0; e <= na; ++e) {
        gb[e] = uint8_t(carry);
        carry += gn[e];
    }
    ok_all = ok_all && carry == hd.np;
    if (carry != hd.np) out[5] |= 4u;
    for (uint32_t e = 0; e < na; ++e)
        if (!outline::bucket_spans(gc, e, na)) {
            ok_all = false;
            out[5] |= 8u;
        }
    // phase 5: pointer positions
    uint32_t links = 0;
    for (uint32_t e = 0; e <= na; ++e) {
        uint32_t n_ptr = 0, count = 0;
        const uint32_t f = e ? uint32_t(end[e - 1]) : hd.pos0, target = e < na ? uint32_t(val[e]) : len;
        if (gn[e] == 0) continue;
        if (gn[e] == 1 && gc[e] != 0 && gb[e] < outline::kMaxPointers) {  // (the kernel's shortcut: the gap's one pointer is the header at its start)
            ptr[gb[e]] = uint16_t(f);
            first_of[gb[e]] = uint8_t(e);
            continue;
        }
        (void)outline::gap_walk(S, f, target, len, e == na, e == 0u, n_ptr, count, ptr, uint32_t(gb[e]), &links, nullptr, first_of, e);
    }
    for (uint32_t e = 0; e < na; ++e) endv[e] = end[e];
    out[0] = hd.np;
    out[1] = na;
    out[2] = links;
    out[3] = uint32_t(hd.bf);
    out[4] = uint32_t(hd.bf >> 32);
    return ok_all ? 1 : 0;
}

}  // extern "C"
