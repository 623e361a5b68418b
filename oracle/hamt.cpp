// oracle/hamt.cpp — TEST INFRASTRUCTURE (see hamt.hpp for provenance).
#include "hamt.hpp"

#include "hashes.hpp"

namespace orc {

namespace {

struct Pointer {
    bool is_link = false;
    Cid link;
    struct KV {
        const uint8_t* key;
        size_t key_len;
        ValueLoc value;
    };
    std::vector<KV> bucket;
};

struct HNode {
    uint8_t bits[32];  // big-endian 256-bit value
    std::vector<Pointer> ptrs;
    bool test(uint32_t idx) const { return bits[31 - idx / 8] & (1u << (idx % 8)); }
    uint32_t rank(uint32_t idx) const {
        uint32_t r = 0;
        for (uint32_t k = 0; k < idx; ++k) r += test(k) ? 1 : 0;
        return r;
    }
};

HNode read_hnode(const Bytes& raw, const ValueChecker& check) {
    Reader r(raw);
    HNode nd;
    r.expect_array(2);
    const uint8_t* bf; size_t bl;
    r.read_bytes(bf, bl);
    if (bl > 32) decode_err("HAMT bitfield longer than 32 bytes");
    std::memset(nd.bits, 0, 32);
    std::memcpy(nd.bits + (32 - bl), bf, bl);
    const uint64_t np = r.read_array();
    for (uint64_t i = 0; i < np; ++i) {
        Pointer p;
        const int major = r.peek_major();
        if (major == 6) {
            p.is_link = true;
            p.link = read_cid(r);
        } else if (major == 4) {
            const uint64_t nkv = r.read_array();
            for (uint64_t k = 0; k < nkv; ++k) {
                r.expect_array(2);
                Pointer::KV kv;
                r.read_bytes(kv.key, kv.key_len);
                kv.value.block = &raw;
                kv.value.off = r.pos;
                check(r);
                kv.value.len = r.pos - kv.value.off;
                p.bucket.push_back(kv);
            }
        } else {
            decode_err("HAMT pointer is neither a link nor a bucket");
        }
        nd.ptrs.push_back(std::move(p));
    }
    r.finish();
    return nd;
}

}  // namespace

bool hamt_get(const Blockstore& bs, const Cid& root, uint32_t bit_width, const uint8_t* key, size_t key_len,
              const ValueChecker& check, ValueLoc& loc) {
    if (bit_width < 1 || bit_width > 8) decode_err("HAMT bit width out of the supported range");
    uint8_t h[32];
    sha256(key, key_len, h);
    uint32_t consumed = 0;
    const Bytes* raw = &must_get(bs, root, "HAMT root");
    for (;;) {
        HNode nd = read_hnode(*raw, check);
        // HashBits::next(bit_width)
        if (consumed + bit_width > 256) throw Err(IPCFP_ST_ERR_MAX_DEPTH, "HAMT max depth");
        uint32_t idx = 0;
        for (uint32_t b = 0; b < bit_width; ++b) {
            const uint32_t bit = consumed + b;
            idx = (idx << 1) | ((h[bit / 8] >> (7 - bit % 8)) & 1u);
        }
        consumed += bit_width;
        if (!nd.test(idx)) return false;
        const uint32_t ci = nd.rank(idx);
        if (ci >= nd.ptrs.size()) decode_err("HAMT bitfield names a pointer that is not there");
        const Pointer& p = nd.ptrs[ci];
        if (p.is_link) {
            raw = &must_get(bs, p.link, "HAMT node");
            continue;
        }
        for (const auto& kv : p.bucket) {
            if (kv.key_len == key_len && std::memcmp(kv.key, key, key_len) == 0) {
                loc = kv.value;
                return true;
            }
        }
        return false;
    }
}

}  // namespace orc
