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

// `kThrow` = true: the first violation throws Err (the `?` of the reference) — every decode of the oracle.
// `kThrow` = false ("try" reader): the first violation sets `bad`, every later call is a no-op returning zero and the
// caller looks at `bad` where a `?` would return.  Same rules, no C++ exception: read_storage_slot attempts five
// decodes of one block per proof (storage/decode.rs:46-89), serde's failures there are plain `Err` values — while a
// C++ throw takes a process-wide lock in the unwinder, which made the all-cores storage baseline run at the speed of
// ONE thread.
template <bool kThrow>
struct ReaderT {
    const uint8_t* p;
    size_t n;
    size_t pos = 0;
    bool bad = false;
    ReaderT(const uint8_t* data, size_t len) : p(data), n(len) {}
    explicit ReaderT(const Bytes& b) : p(b.data()), n(b.size()) {}

    // returns false so that call sites read `return fail("…")` / `if (…) return fail(…)`
    bool fail(const char* what) {
        if constexpr (kThrow) decode_err(what);
        bad = true;
        return false;
    }
    bool eof() const { return pos >= n; }
    uint8_t peek() {
        if (bad) return 0xff;
        if (pos >= n) {
            fail("unexpected end of input");
            return 0xff;
        }
        return p[pos];
    }
    int peek_major() { return peek() >> 5; }

    // header: major type + argument (major = -1 once the reader has failed)
    void head(int& major, uint64_t& arg) {
        major = -1;
        arg = 0;
        uint8_t b = peek();
        if (bad) return;
        ++pos;
        const int m = b >> 5;
        const int ai = b & 31;
        if (ai < 24) {
            if (m == 7 && !(ai >= 20 && ai <= 22)) { fail("unsupported simple value"); return; }
            major = m;
            arg = uint64_t(ai);
            return;
        }
        if (m == 7 && ai != 27) { fail("unsupported simple/float width"); return; }
        int nb;
        if (ai == 24) nb = 1;
        else if (ai == 25) nb = 2;
        else if (ai == 26) nb = 4;
        else if (ai == 27) nb = 8;
        else { fail("indefinite length or reserved additional info"); return; }
        if (pos + size_t(nb) > n) { fail("truncated argument"); return; }
        uint64_t v = 0;
        for (int k = 0; k < nb; ++k) v = (v << 8) | p[pos + k];
        pos += size_t(nb);
        major = m;
        arg = v;
    }

    uint64_t read_uint() {
        int m; uint64_t a;
        head(m, a);
        if (bad) return 0;
        if (m != 0) { fail("expected unsigned integer"); return 0; }
        return a;
    }
    int64_t read_int() {  // i64
        int m; uint64_t a;
        head(m, a);
        if (bad) return 0;
        if (m == 0 || m == 1) {
            if (a > uint64_t(INT64_MAX)) { fail("integer out of i64 range"); return 0; }
            return m == 0 ? int64_t(a) : -1 - int64_t(a);
        }
        fail("expected integer");
        return 0;
    }
    // byte string → [ptr, len)
    void read_bytes(const uint8_t*& out, size_t& len) {
        out = p;
        len = 0;
        int m; uint64_t a;
        head(m, a);
        if (bad) return;
        if (m != 2) { fail("expected byte string"); return; }
        if (a > n - pos) { fail("byte string runs past the end"); return; }
        out = p + pos;
        len = size_t(a);
        pos += len;
    }
    Bytes read_bytes_vec() {
        const uint8_t* q; size_t l;
        read_bytes(q, l);
        return bad ? Bytes() : Bytes(q, q + l);
    }
    std::string read_text() {
        int m; uint64_t a;
        head(m, a);
        if (bad) return std::string();
        if (m != 3) { fail("expected text string"); return std::string(); }
        if (a > n - pos) { fail("text string runs past the end"); return std::string(); }
        if (!utf8_valid(p + pos, size_t(a))) { fail("invalid UTF-8 in text string"); return std::string(); }
        std::string s(reinterpret_cast<const char*>(p + pos), size_t(a));
        pos += size_t(a);
        return s;
    }
    uint64_t read_array() {
        int m; uint64_t a;
        head(m, a);
        if (bad) return 0;
        if (m != 4) { fail("expected array"); return 0; }
        return a;
    }
    void expect_array(uint64_t len) {
        const uint64_t a = read_array();
        if (!bad && a != len) fail("tuple arity mismatch");
    }
    uint64_t read_map() {
        int m; uint64_t a;
        head(m, a);
        if (bad) return 0;
        if (m != 5) { fail("expected map"); return 0; }
        return a;
    }
    bool is_null() { return peek() == 0xf6 && !bad; }
    void read_null() {
        if (peek() != 0xf6) { fail("expected null"); return; }
        if (!bad) ++pos;
    }
    // tag 42 link → raw CID bytes (without the 0x00 multibase prefix); structure validated by the caller
    void read_link(const uint8_t*& cid, size_t& len) {
        cid = p;
        len = 0;
        int m; uint64_t a;
        head(m, a);
        if (bad) return;
        if (m != 6 || a != 42) { fail("expected tag 42"); return; }
        const uint8_t* q; size_t l;
        read_bytes(q, l);
        if (bad) return;
        if (l < 1 || q[0] != 0x00) { fail("CID link must start with the identity multibase prefix"); return; }
        cid = q + 1;
        len = l - 1;
    }
    // IgnoredAny: skip exactly one well-formed item
    void skip() {
        uint64_t todo = 1;
        while (todo && !bad) {
            --todo;
            int m; uint64_t a;
            head(m, a);
            if (bad) return;
            switch (m) {
                case 0: case 1: break;
                case 2:
                    if (a > n - pos) { fail("byte string runs past the end"); return; }
                    pos += size_t(a);
                    break;
                case 3:
                    if (a > n - pos) { fail("text string runs past the end"); return; }
                    if (!utf8_valid(p + pos, size_t(a))) { fail("invalid UTF-8 in text string"); return; }
                    pos += size_t(a);
                    break;
                case 4:
                    if (a > n - pos) { fail("array longer than the input"); return; }
                    todo += a;
                    break;
                case 5:
                    // IgnoredAny over a map: 2·a items.  (Key typing is not enforced when
                    // skipping ⚠; typed map decodes in storage.cpp do enforce text keys.)
                    if (a > (n - pos) / 2) { fail("map longer than the input"); return; }
                    todo += 2 * a;
                    break;
                case 6: {
                    if (a != 42) { fail("unsupported tag"); return; }
                    const uint8_t* q; size_t l;
                    read_bytes(q, l);
                    if (bad) return;
                    if (l < 1 || q[0] != 0x00) { fail("CID link must start with 0x00"); return; }
                    CidParts parts;  // deserialize_any on tag 42 builds a Cid, so the bytes must parse
                    if (!cid_parse_binary(q + 1, l - 1, parts)) { fail("malformed CID in link"); return; }
                    break;
                }
                case 7: break;  // false/true/null/f64: validated by head()
            }
        }
    }
    void finish() {
        if (!bad && pos != n) fail("trailing bytes after the top-level item");
    }
};
using Reader = ReaderT<true>;
using TryReader = ReaderT<false>;

}  // namespace orc
