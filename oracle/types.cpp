// oracle/types.cpp — TEST INFRASTRUCTURE (see types.hpp for provenance).
#include "types.hpp"

#include <array>
#include <map>

namespace orc {

HeaderLite decode_header(const Bytes& raw) {
    Reader r(raw);
    HeaderLite h;
    r.expect_array(16);
    for (int i = 0; i < 5; ++i) r.skip();  // miner, ticket, election proof, beacon entries, winpost proof
    const uint64_t np = r.read_array();    // 5: parents
    for (uint64_t i = 0; i < np; ++i) h.parents.push_back(read_cid(r));
    r.skip();                              // 6: parent weight
    h.height = r.read_int();               // 7
    h.parent_state_root = read_cid(r);     // 8
    h.parent_message_receipts = read_cid(r);  // 9
    h.messages = read_cid(r);              // 10
    r.skip();                              // 11: bls aggregate
    h.timestamp = r.read_uint();           // 12
    r.skip();                              // 13: block sig
    h.fork_signaling = r.read_uint();      // 14
    r.skip();                              // 15: parent base fee
    r.finish();
    return h;
}

void check_cid_value(Reader& r) { (void)read_cid(r); }

void check_receipt(Reader& r) {
    r.expect_array(4);
    if (r.read_uint() > 0xffffffffull) decode_err("exit code does not fit u32");
    const uint8_t* p; size_t n;
    r.read_bytes(p, n);
    (void)r.read_uint();
    if (r.is_null()) r.read_null();
    else (void)read_cid(r);
}

void check_stamped_event(Reader& r) {
    r.expect_array(2);
    (void)r.read_uint();
    const uint64_t ne = r.read_array();
    for (uint64_t i = 0; i < ne; ++i) {
        r.expect_array(4);
        (void)r.read_uint();
        (void)r.read_text();
        (void)r.read_uint();
        const uint8_t* p; size_t n;
        r.read_bytes(p, n);
    }
}

// Address::from_bytes (fvm_shared::address) ⚠ restated: protocol byte + payload shape
static void check_address_bytes(const uint8_t* p, size_t n) {
    if (n < 1) decode_err("empty address");
    auto leb = [&](size_t& pos) {
        for (int k = 0; k < 10; ++k) {
            if (pos >= n) decode_err("truncated leb128 in address");
            if (!(p[pos++] & 0x80)) return;
        }
        decode_err("leb128 too long in address");
    };
    switch (p[0]) {
        case 0: {
            size_t pos = 1;
            leb(pos);
            if (pos != n) decode_err("trailing bytes in ID address");
            break;
        }
        case 1: case 2:
            if (n != 21) decode_err("bad secp/actor address length");
            break;
        case 3:
            if (n != 49) decode_err("bad BLS address length");
            break;
        case 4: {
            size_t pos = 1;
            leb(pos);
            if (n - pos > 54) decode_err("delegated sub-address too long");
            break;
        }
        default:
            decode_err("unknown address protocol");
    }
}

void check_actor_state(Reader& r) {
    r.expect_array(5);
    (void)read_cid(r);
    (void)read_cid(r);
    (void)r.read_uint();
    const uint8_t* p; size_t n;
    r.read_bytes(p, n);  // TokenAmount: bigint bytes, sign byte 0|1 then magnitude
    if (n > 128) decode_err("bigint too long");
    if (n > 0 && p[0] > 1) decode_err("bigint sign byte must be 0 or 1");
    if (r.is_null()) r.read_null();
    else {
        r.read_bytes(p, n);
        check_address_bytes(p, n);
    }
}

void check_vec_u8(Reader& r) {
    const uint64_t n = r.read_array();
    for (uint64_t i = 0; i < n; ++i)
        if (r.read_uint() > 255) decode_err("Vec<u8> element does not fit u8");
}

Receipt decode_receipt(const ValueLoc& v) {
    Reader r(v.block->data() + v.off, v.len);
    Receipt out;
    r.expect_array(4);
    out.exit_code = r.read_uint();
    const uint8_t* p; size_t n;
    r.read_bytes(p, n);
    out.gas_used = r.read_uint();
    if (r.is_null()) r.read_null();
    else {
        out.has_events_root = true;
        out.events_root = read_cid(r);
    }
    return out;
}

StampedEvent decode_stamped_event(const ValueLoc& v) {
    Reader r(v.block->data() + v.off, v.len);
    StampedEvent ev;
    r.expect_array(2);
    ev.emitter = r.read_uint();
    const uint64_t ne = r.read_array();
    for (uint64_t i = 0; i < ne; ++i) {
        EventEntry e;
        r.expect_array(4);
        e.flags = r.read_uint();
        e.key = r.read_text();
        e.codec = r.read_uint();
        r.read_bytes(e.value, e.value_len);
        ev.entries.push_back(e);
    }
    return ev;
}

ActorState decode_actor_state(const ValueLoc& v) {
    Reader r(v.block->data() + v.off, v.len);
    ActorState a;
    r.expect_array(5);
    a.code = read_cid(r);
    a.state = read_cid(r);
    a.sequence = r.read_uint();
    return a;
}

bool extract_evm_log(const StampedEvent& ev, EvmLog& out) {
    // `HashMap<&str, &[u8]>`: a repeated key keeps the LAST value (src/proofs/common/evm.rs:14-17)
    std::map<std::string, std::pair<const uint8_t*, size_t>> m;
    for (const auto& e : ev.entries) m[e.key] = {e.value, e.value_len};
    out.topics.clear();
    out.data.clear();
    auto it = m.find("topics");
    if (it != m.end()) {  // Case A (evm.rs:19-30)
        const auto [p, n] = it->second;
        if (n % 32 != 0) return false;
        for (size_t o = 0; o < n; o += 32) {
            std::array<uint8_t, 32> t;
            std::memcpy(t.data(), p + o, 32);
            out.topics.push_back(t);
        }
        auto d = m.find("data");
        if (d != m.end()) out.data.assign(d->second.first, d->second.first + d->second.second);
        return true;
    }
    // Case B (evm.rs:32-58)
    static const char* keys[4] = {"t1", "t2", "t3", "t4"};
    for (int i = 0; i < 4; ++i) {
        auto t = m.find(keys[i]);
        if (t == m.end()) break;
        if (t->second.second != 32) return false;
        std::array<uint8_t, 32> a;
        std::memcpy(a.data(), t->second.first, 32);
        out.topics.push_back(a);
    }
    if (out.topics.empty()) return false;
    auto d = m.find("d");
    if (d != m.end()) out.data.assign(d->second.first, d->second.first + d->second.second);
    return true;
}

static bool try_evm_state(const Bytes& raw, int fields, Cid& contract_state) {
    try {
        Reader r(raw);
        r.expect_array(uint64_t(fields));
        (void)read_cid(r);  // bytecode
        const uint8_t* p; size_t n;
        r.read_bytes(p, n);  // BytecodeHash: strict_bytes, exactly 32
        if (n != 32) return false;
        contract_state = read_cid(r);
        if (fields == 6) {
            if (r.is_null()) r.read_null(); else r.skip();  // reserved: Option<IgnoredAny>
        }
        (void)r.read_uint();                                   // nonce
        if (r.is_null()) r.read_null(); else r.skip();         // tombstone: Option<IgnoredAny>
        r.finish();
        return true;
    } catch (const Err&) {
        return false;
    }
}

Cid parse_evm_state_contract(const Bytes& raw) {
    Cid c;
    if (try_evm_state(raw, 6, c)) return c;
    if (try_evm_state(raw, 5, c)) return c;
    throw Err(IPCFP_ST_ERR_DECODE, "decode EVM state (5-field)");
}

Cid decode_state_root_actors(const Bytes& raw) {
    Reader r(raw);
    r.expect_array(3);
    if (r.read_uint() > 5) decode_err("unknown StateTreeVersion");
    Cid actors = read_cid(r);
    (void)read_cid(r);
    r.finish();
    return actors;
}

}  // namespace orc
