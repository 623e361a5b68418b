// oracle/types.hpp — TEST INFRASTRUCTURE.  Typed DAG-CBOR decodes of the chain objects on the path.
//
// Restates the serde shapes of (fvm_shared 4.7 is NOT under /root/reference):
//   HeaderLite            src/proofs/common/decode.rs:100-118 (16-tuple, fields 5,7,8,9,10,12,14 typed)
//   TxMeta (Cid, Cid)     src/proofs/events/utils.rs:61
//   Receipt               fvm_shared::receipt::Receipt   [exit_code u32, return_data bytes, gas_used u64, events_root cid|null]
//   StampedEvent          fvm_shared::event              [emitter u64, [[flags u64, key text, codec u64, value bytes]…]]
//   StateRoot             fvm_shared::state              [version 0..5, actors cid, info cid]
//   ActorState            fvm_shared::state              [code cid, state cid, sequence u64, balance bigint-bytes, delegated addr|null]
//   EvmStateV6 / V5       src/proofs/common/decode.rs:49-67
//   Vec<u8>               serde's Vec<u8> = CBOR ARRAY of u8 (not a byte string) — the value type the
//                         reference opens the storage HAMT with (src/proofs/storage/decode.rs:79,86,92) ⚠
#pragma once
#include <array>

#include "amt.hpp"

namespace orc {

struct HeaderLite {
    std::vector<Cid> parents;
    int64_t height = 0;
    Cid parent_state_root, parent_message_receipts, messages;
    uint64_t timestamp = 0, fork_signaling = 0;
};
HeaderLite decode_header(const Bytes& raw);

void check_cid_value(Reader& r);       // V = Cid
void check_receipt(Reader& r);         // V = Receipt
void check_stamped_event(Reader& r);   // V = StampedEvent
void check_actor_state(Reader& r);     // V = ActorState
void check_vec_u8(Reader& r);          // V = Vec<u8>

struct Receipt {
    uint64_t exit_code = 0;
    uint64_t gas_used = 0;
    bool has_events_root = false;
    Cid events_root;
};
Receipt decode_receipt(const ValueLoc& v);

struct EventEntry {
    uint64_t flags = 0, codec = 0;
    std::string key;
    const uint8_t* value = nullptr;
    size_t value_len = 0;
};
struct StampedEvent {
    uint64_t emitter = 0;
    std::vector<EventEntry> entries;
};
StampedEvent decode_stamped_event(const ValueLoc& v);

struct ActorState {
    Cid code, state;
    uint64_t sequence = 0;
};
ActorState decode_actor_state(const ValueLoc& v);

struct EvmLog {
    std::vector<std::array<uint8_t, 32>> topics;
    Bytes data;
};
// extract_evm_log (src/proofs/common/evm.rs:13-59); false ⇒ None
bool extract_evm_log(const StampedEvent& ev, EvmLog& out);

// parse_evm_state (src/proofs/common/decode.rs:79-97): V6 then V5; returns contract_state
Cid parse_evm_state_contract(const Bytes& raw);

// StateRoot decode → actors cid (src/proofs/common/decode.rs:23-26)
Cid decode_state_root_actors(const Bytes& raw);

}  // namespace orc
