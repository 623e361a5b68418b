// oracle/cbor.hpp — TEST INFRASTRUCTURE.  Strict DAG-CBOR reader.
//
// Restates the decoding rules of serde_ipld_dagcbor 0.6 / fvm_ipld_encoding 0.5.3
// (NOT under /root/reference: Cargo.toml:14,18; "parity unpinned").  Every decode in
// the reference goes through `from_slice` — call sites
// src/proofs/common/decode.rs:26,81,90,122; src/proofs/storage/decode.rs:46-85;
// src/proofs/events/utils.rs:25,61; src/proofs/events/verifier.rs:158,174,217.
//
// Rules restated (⚠ = recollection of crate behaviour, applied
// identically by the HIP path so parity is well defined):
//   * definite lengths only (additional info 31 is an error);
//   * the only tag is 42, and its payload is a byte string starting with 0x00;
//   * map keys must be text strings where a map is decoded into a struct;
//   * text strings must be valid UTF-8;
//   * simple values: false / true / null; floats: 64-bit only ⚠;
//   * non-minimal integer/length encodings are accepted ⚠;
//   * `from_slice` rejects trailing bytes after the top-level item.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ipcfp.h"

namespace orc {

// An `Err(..)` of the reference, carrying the status byte the engine reports for it.
struct Err : std::runtime_error {
    uint8_t status;
    Err(uint8_t st, const std::string& what) : std::runtime_error(what), status(st) {}
};
[[noreturn]] inline void decode_err(const char* what) { throw Err(IPCFP_ST_ERR_DECODE, what); }

using Bytes = std::vector<uint8_t>;

inline bool utf8_valid(const uint8_t* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c < 0x80) {
            i += 1;
            continue;
        }
        size_t need;
        uint32_t cp;
        if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1F; }
        else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07; }
        else return false;
        if (need > n - i - 1) return false;
        for (size_t k = 1; k <= need; ++k) {
            const uint8_t cc = s[i + k];
            if ((cc & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (need == 1 && cp < 0x80) return false;
        if (need == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
        if (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
        i += need + 1;
    }
    return true;
}

// ---- binary CID structure (cid 0.11 `Cid::read_bytes`; see oracle/cid.hpp) ----
inline bool read_varint(const uint8_t* p, size_t n, size_t& pos, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 63; shift += 7) {
        if (pos >= n) return false;
        const uint8_t c = p[pos++];
        v |= uint64_t(c & 0x7f) << shift;
        if (!(c & 0x80)) {
            if (c == 0 && shift > 0) return false;  // unsigned-varint: non-minimal encoding rejected
            return true;
        }
    }
    return false;
}

struct CidParts {
    uint64_t version = 0, codec = 0, mh_code = 0, mh_size = 0;
    const uint8_t* digest = nullptr;
};

// Validate the binary form; true iff `p[0..n)` is exactly one well-formed CID.
inline bool cid_parse_binary(const uint8_t* p, size_t n, CidParts& out) {
    if (n == 34 && p[0] == 0x12 && p[1] == 0x20) {
        out.version = 0; out.codec = 0x70; out.mh_code = 0x12; out.mh_size = 32; out.digest = p + 2;
        return true;
    }
    size_t pos = 0;
    if (!read_varint(p, n, pos, out.version) || out.version != 1) return false;
    if (!read_varint(p, n, pos, out.codec)) return false;
    if (!read_varint(p, n, pos, out.mh_code)) return false;
    if (!read_varint(p, n, pos, out.mh_size) || out.mh_size > 64) return false;
    if (n - pos != out.mh_size) return false;
    out.digest = p + pos;
    return true;
}

struct Reader {
    const uint8_t* p;
    size_t n;
    size_t pos = 0;
    Reader(const uint8_t* data, size_t len) : p(data), n(len) {}
    explicit Reader(const Bytes& b) : p(b.data()), n(b.size()) {}

    bool eof() const { return pos >= n; }
    uint8_t peek() const {
        if (pos >= n) decode_err("unexpected end of input");
        return p[pos];
    }
    int peek_major() const { return peek() >> 5; }

    // header: major type + argument
    void head(int& major, uint64_t& arg) {
        uint8_t b = peek();
        ++pos;
        major = b >> 5;
        const int ai = b & 31;
        if (ai < 24) {
            if (major == 7 && !(ai >= 20 && ai <= 22)) decode_err("unsupported simple value");
            arg = uint64_t(ai);
            return;
        }
        if (major == 7 && ai != 27) decode_err("unsupported simple/float width");
        int nb;
        if (ai == 24) nb = 1;
        else if (ai == 25) nb = 2;
        else if (ai == 26) nb = 4;
        else if (ai == 27) nb = 8;
        else { decode_err("indefinite length or reserved additional info"); }
        if (pos + size_t(nb) > n) decode_err("truncated argument");
        uint64_t v = 0;
        for (int k = 0; k < nb; ++k) v = (v << 8) | p[pos + k];
        pos += size_t(nb);
        arg = v;
    }

    uint64_t read_uint() {
        int m; uint64_t a;
        head(m, a);
        if (m != 0) decode_err("expected unsigned integer");
        return a;
    }
    int64_t read_int() {  // i64
        int m; uint64_t a;
        head(m, a);
        if (m == 0) {
            if (a > uint64_t(INT64_MAX)) decode_err("integer out of i64 range");
            return int64_t(a);
        }
        if (m == 1) {
            if (a > uint64_t(INT64_MAX)) decode_err("integer out of i64 range");
            return -1 - int64_t(a);
        }
        decode_err("expected integer");
    }
    // byte string → [ptr, len)
    void read_bytes(const uint8_t*& out, size_t& len) {
        int m; uint64_t a;
        head(m, a);
        if (m != 2) decode_err("expected byte string");
        if (a > n - pos) decode_err("byte string runs past the end");
        out = p + pos;
        len = size_t(a);
        pos += len;
    }
    Bytes read_bytes_vec() {
        const uint8_t* q; size_t l;
        read_bytes(q, l);
        return Bytes(q, q + l);
    }
    std::string read_text() {
        int m; uint64_t a;
        head(m, a);
        if (m != 3) decode_err("expected text string");
        if (a > n - pos) decode_err("text string runs past the end");
        if (!utf8_valid(p + pos, size_t(a))) decode_err("invalid UTF-8 in text string");
        std::string s(reinterpret_cast<const char*>(p + pos), size_t(a));
        pos += size_t(a);
        return s;
    }
    uint64_t read_array() {
        int m; uint64_t a;
        head(m, a);
        if (m != 4) decode_err("expected array");
        return a;
    }
    void expect_array(uint64_t len) {
        if (read_array() != len) decode_err("tuple arity mismatch");
    }
    uint64_t read_map() {
        int m; uint64_t a;
        head(m, a);
        if (m != 5) decode_err("expected map");
        return a;
    }
    bool is_null() const { return peek() == 0xf6; }
    void read_null() {
        if (peek() != 0xf6) decode_err("expected null");
        ++pos;
    }
    // tag 42 link → raw CID bytes (without the 0x00 multibase prefix); structure validated by the caller
    void read_link(const uint8_t*& cid, size_t& len) {
        int m; uint64_t a;
        head(m, a);
        if (m != 6 || a != 42) decode_err("expected tag 42");
        const uint8_t* q; size_t l;
        read_bytes(q, l);
        if (l < 1 || q[0] != 0x00) decode_err("CID link must start with the identity multibase prefix");
        cid = q + 1;
        len = l - 1;
    }
    // IgnoredAny: skip exactly one well-formed item
    void skip() {
        uint64_t todo = 1;
        while (todo) {
            --todo;
            int m; uint64_t a;
            head(m, a);
            switch (m) {
                case 0: case 1: break;
                case 2:
                    if (a > n - pos) decode_err("byte string runs past the end");
                    pos += size_t(a);
                    break;
                case 3:
                    if (a > n - pos) decode_err("text string runs past the end");
                    if (!utf8_valid(p + pos, size_t(a))) decode_err("invalid UTF-8 in text string");
                    pos += size_t(a);
                    break;
                case 4:
                    if (a > n - pos) decode_err("array longer than the input");
                    todo += a;
                    break;
                case 5:
                    // IgnoredAny over a map: 2·a items.  (Key typing is not enforced when
                    // skipping ⚠; typed map decodes in storage.cpp do enforce text keys.)
                    if (a > (n - pos) / 2) decode_err("map longer than the input");
                    todo += 2 * a;
                    break;
                case 6: {
                    if (a != 42) decode_err("unsupported tag");
                    const uint8_t* q; size_t l;
                    read_bytes(q, l);
                    if (l < 1 || q[0] != 0x00) decode_err("CID link must start with 0x00");
                    CidParts parts;  // deserialize_any on tag 42 builds a Cid, so the bytes must parse
                    if (!cid_parse_binary(q + 1, l - 1, parts)) decode_err("malformed CID in link");
                    break;
                }
                case 7: break;  // false/true/null/f64: validated by head()
            }
        }
    }
    void finish() {
        if (pos != n) decode_err("trailing bytes after the top-level item");
    }
};

}  // namespace orc
