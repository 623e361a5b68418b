// oracle/cid.cpp — TEST INFRASTRUCTURE (see cid.hpp).
#include "cid.hpp"

#include "hashes.hpp"

namespace orc {

Cid cid_for_block(const uint8_t* data, size_t len) {
    Cid c;
    c.b = {0x01, 0x71, 0xa0, 0xe4, 0x02, 0x20};
    c.b.resize(38);
    blake2b256(data, len, c.b.data() + 6);
    return c;
}

static const char B32[] = "abcdefghijklmnopqrstuvwxyz234567";

std::string base32_lower(const uint8_t* p, size_t n) {
    std::string out;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        acc = (acc << 8) | p[i];
        bits += 8;
        while (bits >= 5) {
            out.push_back(B32[(acc >> (bits - 5)) & 31]);
            bits -= 5;
        }
    }
    if (bits) out.push_back(B32[(acc << (5 - bits)) & 31]);
    return out;
}

// `upper`: multibase 'B' (RFC 4648 upper-case alphabet); 'b' is the lower-case one.  multibase decodes through
// data-encoding specifications that do not translate case ⚠ (recollection, cid 0.10 / multibase 0.9)
bool base32_decode(const std::string& s, size_t from, Bytes& out, bool upper) {
    out.clear();
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = from; i < s.size(); ++i) {
        const char c = s[i];
        int v;
        if (!upper && c >= 'a' && c <= 'z') v = c - 'a';
        else if (upper && c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= '2' && c <= '7') v = 26 + (c - '2');
        else return false;
        acc = (acc << 5) | uint32_t(v);
        bits += 5;
        if (bits >= 8) {
            out.push_back(uint8_t(acc >> (bits - 8)));
            bits -= 8;
        }
    }
    // leftover bits must be zero padding of fewer than 5 bits … multibase decoders (data-encoding,
    // permissive mode) accept non-zero trailing bits ⚠; we require zero.
    if (bits >= 5) return false;
    if (bits && (acc & ((1u << bits) - 1))) return false;
    return true;
}

static const char B58[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";

std::string base58btc(const uint8_t* p, size_t n) {
    size_t zeros = 0;
    while (zeros < n && p[zeros] == 0) ++zeros;
    std::vector<uint8_t> digits;  // little-endian base-58
    for (size_t i = zeros; i < n; ++i) {
        uint32_t carry = p[i];
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
    for (auto it = digits.rbegin(); it != digits.rend(); ++it) out.push_back(B58[*it]);
    return out;
}

bool base58btc_decode(const std::string& s, size_t from, Bytes& out) {
    out.clear();
    size_t zeros = 0;
    size_t i = from;
    while (i < s.size() && s[i] == '1') { ++zeros; ++i; }
    std::vector<uint8_t> bytes;  // little-endian base-256
    for (; i < s.size(); ++i) {
        const char* f = nullptr;
        for (const char* q = B58; *q; ++q)
            if (*q == s[i]) { f = q; break; }
        if (!f) return false;
        uint32_t carry = uint32_t(f - B58);
        for (auto& b : bytes) {
            carry += uint32_t(b) * 58;
            b = uint8_t(carry & 0xff);
            carry >>= 8;
        }
        while (carry) {
            bytes.push_back(uint8_t(carry & 0xff));
            carry >>= 8;
        }
    }
    out.assign(zeros, 0);
    out.insert(out.end(), bytes.rbegin(), bytes.rend());
    return true;
}

std::string cid_to_string(const Cid& c) {
    CidParts parts;
    if (cid_parse_binary(c.b.data(), c.b.size(), parts) && parts.version == 0)
        return base58btc(c.b.data(), c.b.size());
    return "b" + base32_lower(c.b.data(), c.b.size());
}

// `impl TryFrom<&str> for Cid` (cid crate): `cid_str.find("/ipfs/")` → the text after it; shorter than 2 → Err;
// Version::is_v0_str (46 characters, "Qm") → base58btc; else multibase::decode.
bool cid_from_string(const std::string& full, Cid& out) {
    Bytes raw;
    const size_t cut = full.find("/ipfs/");
    const std::string s = cut == std::string::npos ? full : full.substr(cut + 6);
    if (s.size() == 46 && s[0] == 'Q' && s[1] == 'm') {
        if (!base58btc_decode(s, 0, raw)) return false;
    } else {
        if (s.size() < 2) return false;
        switch (s[0]) {
            case 'b':
                if (!base32_decode(s, 1, raw, false)) return false;
                break;
            case 'B':
                if (!base32_decode(s, 1, raw, true)) return false;
                break;
            case 'z':
                if (!base58btc_decode(s, 1, raw)) return false;
                break;
            case 'f': case 'F': {
                if ((s.size() - 1) % 2) return false;
                for (size_t i = 1; i < s.size(); i += 2) {
                    const bool up = s[0] == 'F';  // one case per prefix
                    auto hv = [up](char c) -> int {
                        if (c >= '0' && c <= '9') return c - '0';
                        if (!up && c >= 'a' && c <= 'f') return c - 'a' + 10;
                        if (up && c >= 'A' && c <= 'F') return c - 'A' + 10;
                        return -1;
                    };
                    const int h = hv(s[i]), l = hv(s[i + 1]);
                    if (h < 0 || l < 0) return false;
                    raw.push_back(uint8_t(h * 16 + l));
                }
                break;
            }
            default:
                return false;  // other multibases are not produced by Lotus / this path
        }
    }
    CidParts parts;
    if (!cid_parse_binary(raw.data(), raw.size(), parts)) return false;
    out.b = raw;
    return true;
}

}  // namespace orc
