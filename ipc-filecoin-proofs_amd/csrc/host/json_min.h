// csrc/host/json_min.h — a pull parser for exactly the JSON the proof bundle needs.
//
// Mirrors what `serde_json::from_str` + `#[derive(Deserialize)]` accept for the bundle structs
// (src/proofs/common/bundle.rs:10-45, events/bundle.rs:5-30, storage/bundle.rs:4-14): RFC 8259 syntax,
// 128 levels of nesting, integers only where the field is u64/i64 (a float, an exponent form, "-0" or an
// out-of-range literal is an `invalid type` error), strings with escapes (lone surrogates rejected),
// unknown fields skipped, duplicate or missing fields rejected by the caller.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace ipcfp {

struct JsonCursor {
    const char* p;
    const char* end;
    const char* begin;
    std::string err;

    JsonCursor(const char* s, size_t n) : p(s), end(s + n), begin(s) {}
    bool failed() const { return !err.empty(); }
    bool fail(const char* what) {
        if (err.empty()) err = std::string(what) + " at byte " + std::to_string(size_t(p - begin));
        return false;
    }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool peek(char c) {
        ws();
        return p < end && *p == c;
    }
    bool consume(char c) {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    bool expect(char c, const char* what) { return consume(c) || fail(what); }
    bool at_end() {
        ws();
        return p == end;
    }

    // A string whose body is returned as a raw span (no unescaping); `plain` = no backslash inside.
    bool string_span(const char*& s, size_t& n, bool& plain) {
        ws();
        if (p >= end || *p != '"') return fail("expected a string");
        ++p;
        s = p;
        plain = true;
        for (;;) {
            // jump to the next quote; strings in a bundle are long (base64) and almost never escaped
            const char* q = static_cast<const char*>(std::memchr(p, '"', size_t(end - p)));
            if (!q) return fail("unterminated string");
            // a quote preceded by an odd number of backslashes is escaped
            const char* b = q;
            while (b > s && b[-1] == '\\') --b;
            if ((q - b) & 1) {
                plain = false;
                p = q + 1;
                continue;
            }
            if (q != s && std::memchr(s, '\\', size_t(q - s))) plain = false;
            n = size_t(q - s);
            p = q + 1;
            return true;
        }
    }

    static int hexval(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    static void put_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out.push_back(char(cp));
        else if (cp < 0x800) {
            out.push_back(char(0xC0 | (cp >> 6)));
            out.push_back(char(0x80 | (cp & 63)));
        } else if (cp < 0x10000) {
            out.push_back(char(0xE0 | (cp >> 12)));
            out.push_back(char(0x80 | ((cp >> 6) & 63)));
            out.push_back(char(0x80 | (cp & 63)));
        } else {
            out.push_back(char(0xF0 | (cp >> 18)));
            out.push_back(char(0x80 | ((cp >> 12) & 63)));
            out.push_back(char(0x80 | ((cp >> 6) & 63)));
            out.push_back(char(0x80 | (cp & 63)));
        }
    }
    // validate (control characters, escapes) and unescape a span produced by string_span
    bool unescape(const char* s, size_t n, std::string& out) {
        out.clear();
        out.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            const unsigned char c = static_cast<unsigned char>(s[i]);
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') {
                out.push_back(char(c));
                continue;
            }
            if (++i >= n) return fail("bad escape");
            switch (s[i]) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    auto hex4 = [&](size_t at, uint32_t& v) {
                        if (at + 4 > n) return false;
                        v = 0;
                        for (int k = 0; k < 4; ++k) {
                            const int h = hexval(s[at + k]);
                            if (h < 0) return false;
                            v = (v << 4) | uint32_t(h);
                        }
                        return true;
                    };
                    uint32_t cp;
                    if (!hex4(i + 1, cp)) return fail("bad \\u escape");
                    i += 4;
                    if (cp >= 0xDC00 && cp <= 0xDFFF) return fail("lone trailing surrogate");
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        uint32_t lo;
                        if (i + 2 >= n || s[i + 1] != '\\' || s[i + 2] != 'u' || !hex4(i + 3, lo) || lo < 0xDC00 || lo > 0xDFFF)
                            return fail("lone leading surrogate");
                        i += 6;
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    put_utf8(out, cp);
                    break;
                }
                default: return fail("bad escape");
            }
        }
        return true;
    }
    bool string(std::string& out) {
        const char* s;
        size_t n;
        bool plain;
        if (!string_span(s, n, plain)) return false;
        return unescape(s, n, out);
    }
    // validate a span as string content without materialising it
    bool check_plain(const char* s, size_t n) {
        for (size_t i = 0; i < n; ++i)
            if (static_cast<unsigned char>(s[i]) < 0x20) return fail("control character in string");
        return true;
    }

    // JSON number grammar; integer = no fraction and no exponent
    bool number_span(const char*& s, size_t& n, bool& integer, bool& negative) {
        ws();
        s = p;
        negative = false;
        integer = true;
        if (p < end && *p == '-') {
            negative = true;
            ++p;
        }
        if (p >= end) return fail("expected a number");
        if (*p == '0') ++p;
        else if (*p >= '1' && *p <= '9') {
            while (p < end && *p >= '0' && *p <= '9') ++p;
        } else return fail("expected a number");
        if (p < end && *p == '.') {
            integer = false;
            ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("bad number");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            integer = false;
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("bad number");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        n = size_t(p - s);
        return true;
    }
    bool u64(uint64_t& v) {
        const char* s;
        size_t n;
        bool integer, negative;
        if (!number_span(s, n, integer, negative)) return false;
        if (!integer || negative) return fail("invalid type: expected u64");
        v = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t d = uint64_t(s[i] - '0');
            if (v > (UINT64_MAX - d) / 10) return fail("invalid type: integer out of range for u64");
            v = v * 10 + d;
        }
        return true;
    }
    bool i64(int64_t& v) {
        const char* s;
        size_t n;
        bool integer, negative;
        if (!number_span(s, n, integer, negative)) return false;
        if (!integer) return fail("invalid type: expected i64");
        uint64_t m = 0;
        for (size_t i = negative ? 1 : 0; i < n; ++i) {
            const uint64_t d = uint64_t(s[i] - '0');
            if (m > (UINT64_MAX - d) / 10) return fail("invalid type: integer out of range for i64");
            m = m * 10 + d;
        }
        if (negative) {
            if (m == 0) return fail("invalid type: -0 is a float");  // serde_json parses "-0" as -0.0
            if (m > (1ULL << 63)) return fail("invalid type: integer out of range for i64");
            v = int64_t(~m + 1);
        } else {
            if (m > uint64_t(INT64_MAX)) return fail("invalid type: integer out of range for i64");
            v = int64_t(m);
        }
        return true;
    }

    // serde's IgnoredAny: any well-formed value
    bool skip_value(int depth = 0) {
        ws();
        if (p >= end) return fail("expected a value");
        const char c = *p;
        if (c == '"') {
            std::string tmp;
            return string(tmp);
        }
        // serde_json: 128 levels; entering the 128th container is the error
        if ((c == '{' || c == '[') && depth + 1 >= 128) return fail("recursion limit exceeded");
        if (c == '{') {
            ++p;
            if (consume('}')) return true;
            for (;;) {
                std::string k;
                if (!string(k)) return false;
                if (!expect(':', "expected ':'")) return false;
                if (!skip_value(depth + 1)) return false;
                if (consume(',')) continue;
                return expect('}', "expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p;
            if (consume(']')) return true;
            for (;;) {
                if (!skip_value(depth + 1)) return false;
                if (consume(',')) continue;
                return expect(']', "expected ',' or ']'");
            }
        }
        auto lit = [&](const char* w) {
            const size_t k = std::strlen(w);
            if (size_t(end - p) >= k && std::memcmp(p, w, k) == 0) {
                p += k;
                return true;
            }
            return fail("expected a value");
        };
        if (c == 't') return lit("true");
        if (c == 'f') return lit("false");
        if (c == 'n') return lit("null");
        const char* s;
        size_t n;
        bool integer, negative;
        return number_span(s, n, integer, negative);
    }
};

}  // namespace ipcfp
