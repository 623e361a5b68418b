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
