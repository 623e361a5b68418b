// csrc/kernels/header_dev.h — `HeaderLite` (src/proofs/common/decode.rs:100-124) on the device; needs the CBOR reader only.
#pragma once
#include "cbor_dev.h"

namespace ipcfp {

// HeaderLite (src/proofs/common/decode.rs:100-118): 16-tuple; fields 5,7,8,9,10,12,14 typed.
struct HeaderLite {
    uint32_t parents_off;  // offset of the first parent link item (after the array header)
    uint32_t n_parents;
    long long height;
    CidKey parent_state_root, parent_message_receipts, messages;
};

// from_slice::<HeaderLite>(raw): TRUE or ERR_DECODE
__device__ __forceinline__ uint32_t decode_header(Rd& r, HeaderLite& h) {
    r.expect_array(16);
    for (int i = 0; i < 5; ++i) r.skip();
    const uint64_t np = r.read_array();
    h.parents_off = r.pos;
    h.n_parents = np > 0xffffffffULL ? 0xffffffffu : uint32_t(np);
    for (uint64_t i = 0; i < np && r.ok(); ++i) {
        uint32_t o, l;
        r.read_link(o, l);
    }
    r.skip();
    h.height = r.read_int();
    r.read_link_key(h.parent_state_root);
    r.read_link_key(h.parent_message_receipts);
    r.read_link_key(h.messages);
    r.skip();
    (void)r.read_uint();
    r.skip();
    (void)r.read_uint();
    r.skip();
    r.finish();
    return r.ok() ? IPCFP_ST_TRUE : IPCFP_ST_ERR_DECODE;
}

}  // namespace ipcfp
