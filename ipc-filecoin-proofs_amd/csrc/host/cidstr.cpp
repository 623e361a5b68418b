// csrc/host/cidstr.cpp — see cidstr.h.  Host-side string handling of the product; written
// for the engine (it does not share code with oracle/).
#include "cidstr.h"

#include <cstring>

namespace ipcfp {

namespace {

bool uvarint(const uint8_t* p, size_t n, size_t& pos, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 63; shift += 7) {
        if (pos >= n) return false;
        const uint8_t c = p[pos++];
        v |= uint64_t(c & 0x7f) << shift;
        if (!(c & 0x80)) return !(c == 0 && shift > 0);
    }
    return false;
}

// multibase 'b' is the LOWER-case RFC 4648 alphabet and 'B' the upper-case one: the other case is not in the alphabet
// (multibase decodes through data-encoding specifications without case translation)
int b32val(char c, bool upper) {
    if (!upper && c >= 'a' && c <= 'z') return c - 'a';
    if (upper && c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= '2' && c <= '7') return 26 + (c - '2');
    return -1;
}

bool base32_decode(const char* s, size_t n, bool upper, std::vector<uint8_t>& out) {
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        const int v = b32val(s[i], upper);
        if (v < 0) return false;
        acc = (acc << 5) | uint32_t(v);
        bits += 5;
        if (bits >= 8) {
            out.push_back(uint8_t(acc >> (bits - 8)));
            bits -= 8;
        }
    }
    if (bits >= 5) return false;
    return !(bits && (acc & ((1u << bits) - 1)));
}

const char kB58[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";

bool base58_decode(const char* s, size_t n, std::vector<uint8_t>& out) {
    size_t i = 0, zeros = 0;
    while (i < n && s[i] == '1') { ++zeros; ++i; }
    std::vector<uint8_t> num;  // little-endian
    for (; i < n; ++i) {
        const char* f = static_cast<const char*>(std::memchr(kB58, s[i], 58));
        if (!f) return false;
        uint32_t carry = uint32_t(f - kB58);
        for (auto& b : num) {
            carry += uint32_t(b) * 58;
            b = uint8_t(carry);
            carry >>= 8;
        }
        while (carry) {
            num.push_back(uint8_t(carry));
            carry >>= 8;
        }
    }
    out.assign(zeros, 0);
    out.insert(out.end(), num.rbegin(), num.rend());
    return true;
}

int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

}  // namespace

bool hex_decode(const char* s, size_t n, std::vector<uint8_t>& out) {
    if (n % 2) return false;
    out.clear();
    out.reserve(n / 2);
    for (size_t i = 0; i < n; i += 2) {
        const int h = hexval(s[i]), l = hexval(s[i + 1]);
        if (h < 0 || l < 0) return false;
        out.push_back(uint8_t(h * 16 + l));
    }
    return true;
}

// ---- Blake2b-256 on the host (RFC 7693; digest length 32, no key): only for folding CIDs longer than the ABI slot ----
namespace {
inline uint64_t rotr64(uint64_t x, unsigned n) { return (x >> n) | (x << (64 - n)); }
const uint64_t kB2bIV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                            0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
const uint8_t kB2bSigma[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
void b2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) {
        m[i] = 0;
        for (int b = 0; b < 8; ++b) m[i] |= uint64_t(block[8 * i + b]) << (8 * b);
    }
    for (int i = 0; i < 8; ++i) {
        v[i] = h[i];
        v[8 + i] = kB2bIV[i];
    }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
        v[a] = v[a] + v[b] + x; v[d] = rotr64(v[d] ^ v[a], 32);
        v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 24);
        v[a] = v[a] + v[b] + y; v[d] = rotr64(v[d] ^ v[a], 16);
        v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = kB2bSigma[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[8 + i];
}
}  // namespace

void blake2b256_host(const uint8_t* data, size_t len, uint8_t out32[32]) {
    uint64_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = kB2bIV[i];
    h[0] ^= 0x01010020ULL;  // digest length 32, no key, fanout 1, depth 1
    size_t done = 0;
    uint8_t block[128];
    for (;;) {
        const size_t left = len - done;
        const bool last = left <= 128;
        const size_t take = last ? left : 128;
        std::memset(block, 0, sizeof block);
        if (take) std::memcpy(block, data + done, take);
        done += take;
        b2b_compress(h, block, uint64_t(done), last);
        if (last) break;
    }
    for (int i = 0; i < 32; ++i) out32[i] = uint8_t(h[i >> 3] >> (8 * (i & 7)));
}

void cid_to_slot(const uint8_t* cid, size_t len, uint8_t slot40[40]) {
    std::memset(slot40, 0, 40);
    if (len <= 40) {
        if (len) std::memcpy(slot40, cid, len);
        return;
    }
    slot40[0] = 0xff;
    slot40[1] = uint8_t(len);
    blake2b256_host(cid, len, slot40 + 2);
}

bool cid_binary_ok(const uint8_t* p, size_t n) {
    if (n == 34 && p[0] == 0x12 && p[1] == 0x20) return true;
    size_t pos = 0;
    uint64_t version, codec, code, size;
    if (!uvarint(p, n, pos, version) || version != 1) return false;
    if (!uvarint(p, n, pos, codec) || !uvarint(p, n, pos, code)) return false;
    if (!uvarint(p, n, pos, size) || size > 64) return false;
    return n - pos == size;
}

// hex digits of ONE case (multibase 'f' / 'F')
bool hex_decode_cased(const char* s, size_t n, bool upper, std::vector<uint8_t>& out) {
    for (size_t i = 0; i < n; ++i) {
        const char c = s[i];
        if (upper ? (c >= 'a' && c <= 'f') : (c >= 'A' && c <= 'F')) return false;
    }
    return hex_decode(s, n, out);
}

// `Cid::try_from(&str)` of the cid crate: everything up to and including the first "/ipfs/" is dropped, a 46-character
// "Qm…" string is a CIDv0 in base58btc, anything else a multibase string.  Of the multibase alphabets the ones Lotus
// and the reference produce or print are decoded — b / B (base32 lower / upper, unpadded), f / F (base16), z
// (base58btc); the others (k, m, u, …) are an engine limit: such a claim string is reported as unparsable.
bool cid_from_string(const char* s, std::vector<uint8_t>& out) {
    out.clear();
    if (!s) return false;
    if (const char* cut = std::strstr(s, "/ipfs/")) s = cut + 6;
    const size_t n = std::strlen(s);
    if (n < 2) return false;
    if (n == 46 && s[0] == 'Q' && s[1] == 'm') {
        if (!base58_decode(s, n, out)) return false;
    } else {
        bool ok;
        switch (s[0]) {
            case 'b': ok = base32_decode(s + 1, n - 1, false, out); break;
            case 'B': ok = base32_decode(s + 1, n - 1, true, out); break;
            case 'z': ok = base58_decode(s + 1, n - 1, out); break;
            case 'f': ok = hex_decode_cased(s + 1, n - 1, false, out); break;
            case 'F': ok = hex_decode_cased(s + 1, n - 1, true, out); break;
            default: ok = false;
        }
        if (!ok) return false;
    }
    return cid_binary_ok(out.data(), out.size());
}

std::string cid_to_string(const uint8_t* cid, size_t len) {
    if (len == 34 && cid[0] == 0x12 && cid[1] == 0x20) {
        std::vector<uint8_t> digits;
        size_t zeros = 0;
        while (zeros < len && cid[zeros] == 0) ++zeros;
        for (size_t i = zeros; i < len; ++i) {
            uint32_t carry = cid[i];
            for (auto& d : digits) {
                carry += uint32_t(d) << 8;
                d = uint8_t(carry % 58);
                carry /= 58;
            }
            while (carry) {
                digits.push_back(uint8_t(carry % 58));
                carry /= 58;
            }
        }
        std::string out(zeros, '1');
        for (auto it = digits.rbegin(); it != digits.rend(); ++it) out.push_back(kB58[*it]);
        return out;
    }
    static const char a[] = "abcdefghijklmnopqrstuvwxyz234567";
    std::string out = "b";
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < len; ++i) {
        acc = (acc << 8) | cid[i];
        bits += 8;
        while (bits >= 5) {
            out.push_back(a[(acc >> (bits - 5)) & 31]);
            bits -= 5;
        }
    }
    if (bits) out.push_back(a[(acc << (5 - bits)) & 31]);
    return out;
}

}  // namespace ipcfp
