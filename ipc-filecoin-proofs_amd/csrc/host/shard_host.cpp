// csrc/host/shard_host.cpp — the host half of "plan once, scatter" (SURVEY.md §8e): cutting a shard's witness and
// its claims out of a bundle that lies in HOST memory, so that rank r uploads only what it needs.
//
// The reference walks one bundle on one thread (src/proofs/verifier.rs:19-28,49-54; src/proofs/events/verifier.rs:62-71).
// A node with G GPUs has G PCIe links: the window that includes the upload (T2) divides by G only if every rank's
// upload is its own shard — the block-id lists of ipcfp_shard_plan_tipset_all, computed ONCE (by whoever holds the
// whole witness: the bundle's producer, or one rank) and handed to the ranks.  Nothing here touches a device.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "../common.h"

namespace {

unsigned worker_count(uint64_t bytes) {
    if (bytes < (8u << 20)) return 1;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return std::min(hw, 8u);
}

template <class F>
void parallel_ranges(uint64_t n, unsigned T, F&& body) {
    if (T <= 1 || n < T) {
        body(0, n);
        return;
    }
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < T; ++t) pool.emplace_back([&, t] { body(n * t / T, n * (t + 1) / T); });
    } catch (...) {  // thread creation failed: whatever did not start is done here
        const unsigned started = unsigned(pool.size()) + 1;
        body(n * started / T, n);
    }
    body(0, n / T);
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int ipcfp_witness_cut_host(const uint8_t* bytes, uint64_t nbytes, const uint64_t* off, const uint32_t* len,
                           const uint8_t* cids40, uint64_t n_src, const uint32_t* block_ids, uint64_t n,
                           uint8_t* out_bytes, uint64_t cap_bytes, uint64_t* out_off, uint32_t* out_len,
                           uint8_t* out_cids40, uint64_t* nbytes_out) {
    if (!nbytes_out || (n && (!block_ids || !off || !len || !cids40)) || (nbytes && !bytes)) return IPCFP_E_INVALID;
    // sizes and bounds first: nothing is written unless every id and every block is in range
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t id = block_ids[i];
        if (id >= n_src || off[id] > nbytes || uint64_t(len[id]) > nbytes - off[id]) return IPCFP_E_INVALID;
        total += len[id];
    }
    *nbytes_out = total;
    if (!out_bytes && !out_off && !out_len && !out_cids40) return IPCFP_OK;  // the sizing call
    if (!out_off || !out_len || !out_cids40 || (total && !out_bytes) || cap_bytes < total) return IPCFP_E_INVALID;
    uint64_t at = 0;
    for (uint64_t i = 0; i < n; ++i) {
        out_off[i] = at;
        out_len[i] = len[block_ids[i]];
        at += out_len[i];
    }
    parallel_ranges(n, worker_count(total), [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t id = block_ids[i];
            if (out_len[i]) std::memcpy(out_bytes + out_off[i], bytes + off[id], out_len[i]);
            std::memcpy(out_cids40 + i * IPCFP_CID_SLOT, cids40 + uint64_t(id) * IPCFP_CID_SLOT, IPCFP_CID_SLOT);
        }
    });
    return IPCFP_OK;
}

int ipcfp_route_event_claims(const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob, uint64_t blob_len,
                             uint64_t receipt_lo, uint64_t receipt_hi, int last_shard, uint64_t* positions,
                             ipcfp_event_claim_t* out_claims, uint64_t cap_claims, uint8_t* out_blob, uint64_t cap_blob,
                             uint64_t* n_out, uint64_t* blob_out) {
    if (!n_out || !blob_out || (n && !claims)) return IPCFP_E_INVALID;
    auto mine = [&](const ipcfp_event_claim_t& c) {
        return c.exec_index >= receipt_lo && (last_shard || c.exec_index < receipt_hi);
    };
    auto spans_ok = [&](const ipcfp_event_claim_t& c) {  // a claim whose offsets lie outside the blob carries none of it
        const uint64_t tl = uint64_t(c.n_topics) * 33;
        return blob && c.n_topics <= (1u << 20) && uint64_t(c.topics_off) <= blob_len && tl <= blob_len - c.topics_off &&
               uint64_t(c.data_off) <= blob_len && uint64_t(c.data_len) <= blob_len - c.data_off;
    };
    uint64_t cnt = 0, bytes = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (mine(claims[i])) {
            ++cnt;
            if (spans_ok(claims[i])) bytes += uint64_t(claims[i].n_topics) * 33 + claims[i].data_len;
        }
    *n_out = cnt;
    *blob_out = bytes;
    if (!out_claims && !out_blob && !positions) return IPCFP_OK;  // the sizing call
    if ((cnt && !out_claims) || cap_claims < cnt || (bytes && !out_blob) || cap_blob < bytes) return IPCFP_E_INVALID;
    if (bytes > 0xffffffffULL) return IPCFP_E_UNSUPPORTED;  // blob offsets are u32
    uint64_t k = 0, at = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (!mine(claims[i])) continue;
        ipcfp_event_claim_t c = claims[i];
        if (spans_ok(c)) {
            const uint64_t tl = uint64_t(c.n_topics) * 33;
            if (tl) std::memcpy(out_blob + at, blob + c.topics_off, tl);
            c.topics_off = uint32_t(at);
            at += tl;
            if (c.data_len) std::memcpy(out_blob + at, blob + c.data_off, c.data_len);
            c.data_off = uint32_t(at);
            at += c.data_len;
        } else {  // keep it out of range in the shard's blob too: the device gives it ERR_BAD_CLAIM either way
            c.topics_off = c.data_off = 0xffffffffu;
        }
        out_claims[k] = c;
        if (positions) positions[k] = i;
        ++k;
    }
    return IPCFP_OK;
}

}  // extern "C"
