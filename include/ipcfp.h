/* ipcfp.h — C ABI of the MI355X-native batch AMT/HAMT Merkle-witness engine.
 *
 * This header is the drop-in boundary.  The reference
 * (consensus-shipyard/ipc-filecoin-proofs, Rust) has no FFI of its own: its seam
 * is the `fvm_ipld_blockstore::Blockstore` trait plus four free functions.  Each
 * entry point below names the reference interface it replaces (paths relative to
 * /root/reference/).  A Rust host binds these with `extern "C"`
 * (bindings/rust/ffi.rs).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch/Rust types cross the ABI.
 *   - every function returns 0 (IPCFP_OK) or a negative IPCFP_E_* code; the text of
 *     the last failure on a context is available from ipcfp_last_error().
 *   - "host" pointers are ordinary process memory, BORROWED for the duration of
 *     the call.  "device" pointers (suffix _d / functions named *_device) are HBM
 *     addresses on the context's GPU (e.g. a hipMalloc'ed buffer or
 *     torch.Tensor.data_ptr()).
 *   - a context is thread-compatible (external synchronisation), owns one HIP
 *     stream, and all calls are synchronous unless named *_async.
 *   - per-proof outcomes are STATUS BYTES (ipcfp_status_t), because the reference
 *     distinguishes Ok(true) / Ok(false) / Err.  The wrapper maps
 *     the lowest-index ERR_* of a batch to `Err(..)`, exactly as the reference's
 *     first-error-aborts loop does (src/proofs/events/verifier.rs:62-71,
 *     src/proofs/verifier.rs:19-28).
 */
#ifndef IPCFP_H
#define IPCFP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever the layout of a public struct or the meaning of an argument changes; a binding compares it with
 * ipcfp_abi_version() before its first call.  2: IPCFP_MAX_PARENTS 16 -> 32 (sizeof(ipcfp_tipset_ref_t) 648 -> 1328).
 * 3: ipcfp_tipset_ref_t.more_parents (tipset keys of ANY length: 1328 -> 1336); a CID longer than the slot crosses the ABI
 * FOLDED (ipcfp_cid_to_slot) instead of being refused.                                                                */
#define IPCFP_ABI_VERSION 3

/* ---- return codes ------------------------------------------------------- */
#define IPCFP_OK 0
#define IPCFP_E_INVALID (-1)     /* bad argument (null pointer, size mismatch …)      */
#define IPCFP_E_NO_DEVICE (-2)   /* no gfx950 device / HIP runtime unusable           */
#define IPCFP_E_HIP (-3)         /* a HIP call failed; see ipcfp_last_error()         */
#define IPCFP_E_NOMEM (-4)       /* host or device allocation failed                  */
#define IPCFP_E_UNSUPPORTED (-5) /* e.g. more than 2^32-2 blocks, a claim blob over 3.75 GB */
#define IPCFP_E_PARSE (-6)       /* a CID / hex string could not be parsed (host side)*/

/* ---- CIDs --------------------------------------------------------------- */
/* Binary CIDs cross the ABI in fixed 40-byte slots, zero padded on the right.
 * The Filecoin chain CID (CIDv1, dag-cbor 0x71, blake2b-256 0xb220, 32-byte
 * digest) is 38 bytes: 01 71 a0 e4 02 20 ‖ digest.  A binary CID is
 * self-delimiting, so zero padding is unambiguous.  Replaces `cid::Cid` values
 * (src/proofs/common/bundle.rs:12, src/proofs/common/witness.rs:60-72).
 * A CID of MORE than 40 bytes (a 64-byte digest: `cid` 0.11 takes multihashes of up to 64 bytes, so
 * `Cid::try_from` / serde accept it and src/proofs/common/witness.rs:60-72 stores a block under it) crosses FOLDED:
 *     ff | len | blake2b-256(the CID's bytes) | 00 …            (34 bytes; ipcfp_cid_to_slot)
 * 0xff starts no CID (a CIDv1 starts 01, a CIDv0 12 20), so a fold equals no short CID; two folds are equal iff the
 * CIDs are, short of a blake2b-256 collision — the assumption every chain CID rests on already.  The device folds a
 * long link it reads out of a block the same way, so long CIDs are found, compared and deduplicated like any other.
 * Where a CID comes BACK across the ABI in a slot (ipcfp_exec_order, the generator's message CIDs, touched sets) a long
 * one comes back as its fold: the caller that needs its bytes reads them from the block they stand in.           */
#define IPCFP_CID_SLOT 40
#define IPCFP_CID_BLAKE2B_LEN 38
#define IPCFP_CID_MAX_LEN 104 /* 1 + 9 + 9 + 1 + 64 rounded up: the longest binary CID `cid` 0.11 parses */

/* ---- per-item status bytes ---------------------------------------------- */
typedef uint8_t ipcfp_status_t;
enum {
    IPCFP_ST_FALSE = 0, /* Ok(false) — generic                                        */
    IPCFP_ST_TRUE = 1,  /* Ok(true)                                                   */
    /* Ok(false) with a reason (values 2..63)                                         */
    IPCFP_ST_FALSE_UNTRUSTED_PARENT = 2,   /* events/verifier.rs:134                  */
    IPCFP_ST_FALSE_UNTRUSTED_CHILD = 3,    /* events/verifier.rs:139, storage/verifier.rs:87 */
    IPCFP_ST_FALSE_PARENTS_MISMATCH = 4,   /* events/verifier.rs:161                  */
    IPCFP_ST_FALSE_CHILD_EPOCH = 5,        /* events/verifier.rs:166                  */
    IPCFP_ST_FALSE_PARENT_EPOCH = 6,       /* events/verifier.rs:176                  */
    IPCFP_ST_FALSE_MSG_NOT_IN_EXEC = 7,    /* events/verifier.rs:194                  */
    IPCFP_ST_FALSE_EXEC_INDEX = 8,         /* events/verifier.rs:199                  */
    IPCFP_ST_FALSE_NO_RECEIPT = 9,         /* events/verifier.rs:224                  */
    IPCFP_ST_FALSE_NO_EVENTS_ROOT = 10,    /* events/verifier.rs:229                  */
    IPCFP_ST_FALSE_NO_EVENT = 11,          /* events/verifier.rs:237                  */
    IPCFP_ST_FALSE_EMITTER = 12,           /* events/verifier.rs:262                  */
    IPCFP_ST_FALSE_NOT_EVM_LOG = 13,       /* events/verifier.rs:267                  */
    IPCFP_ST_FALSE_TOPIC_COUNT = 14,       /* events/verifier.rs:272                  */
    IPCFP_ST_FALSE_TOPIC = 15,             /* events/verifier.rs:276-281              */
    IPCFP_ST_FALSE_DATA = 16,              /* events/verifier.rs:284-287              */
    IPCFP_ST_FALSE_FILTER = 17,            /* events/verifier.rs:247-251              */
    IPCFP_ST_FALSE_STATE_ROOT = 18,        /* storage/verifier.rs:110                 */
    IPCFP_ST_FALSE_ACTOR_STATE = 19,       /* storage/verifier.rs:126                 */
    IPCFP_ST_FALSE_STORAGE_ROOT = 20,      /* storage/verifier.rs:144                 */
    IPCFP_ST_FALSE_VALUE = 21,             /* storage/verifier.rs:169                 */
    IPCFP_ST_NOT_FOUND = 32,               /* primitive gets: Ok(None)                */
    /* Err(..) (values >= 64)                                                         */
    IPCFP_ST_ERR = 64,                     /* generic Err                             */
    IPCFP_ST_ERR_MISSING_BLOCK = 65,       /* a CID the walk needs is not in the witness */
    IPCFP_ST_ERR_DECODE = 66,              /* DAG-CBOR shape the reference's serde rejects */
    IPCFP_ST_ERR_TXMETA_MISMATCH = 67,     /* events/utils.rs:66-72                   */
    IPCFP_ST_ERR_ACTOR_NOT_FOUND = 68,     /* common/decode.rs:39                     */
    IPCFP_ST_ERR_BAD_CLAIM = 69,           /* unparsable CID / hex string in the claim */
    IPCFP_ST_ERR_MAX_DEPTH = 70,           /* HAMT hash bits exhausted                */
    IPCFP_ST_ERR_EMPTY_PARENTS = 71        /* events/verifier.rs:172 `parent_cids[0]` panics */
};
#define IPCFP_ST_IS_ERR(s) ((s) >= 64)

/* per-block CID check outcome (ipcfp_witness_verify_cids) */
enum {
    IPCFP_CID_MISMATCH = 0,   /* bytes do not hash to the claimed CID                   */
    IPCFP_CID_OK = 1,         /* Blake2b-256(bytes) == digest carried by the CID        */
    IPCFP_CID_UNCHECKED = 2   /* CID is not (v1, *, blake2b-256, 32): not hashed here   */
};

/* Host-side string forms (no GPU involved; usable without a context).
 * ipcfp_cid_from_string: `Cid::try_from(&str)` (src/proofs/common/witness.rs:60-64) as the cid crate does it: the text
 *   up to and including the first "/ipfs/" is dropped; a 46-character "Qm…" is a CIDv0 (base58btc); else multibase,
 *   case-strict — b (base32 lower), B (base32 upper), f / F (base16 lower / upper), z (base58btc).  The other
 *   multibase alphabets (k, m, u, …) are an ENGINE LIMIT: such a string is reported like an unparsable one.  Returns
 *   the CID's byte length — the CID written zero-padded into the 40-byte slot, or FOLDED when it is longer than the slot
 *   (see "CIDs" above) — or IPCFP_E_PARSE if the parse is Err.
 * ipcfp_cid_to_slot: a binary CID of any length → its slot (zero padded, or folded); returns the CID's length,
 *   IPCFP_E_PARSE when the bytes are not exactly one well-formed CID.
 * ipcfp_cid_to_string: `Cid::to_string()` — "b" + base32-lower for CIDv1, base58btc for CIDv0; returns
 *   the string length (excluding NUL) or IPCFP_E_INVALID if cap is too small / the bytes are not a CID. */
int ipcfp_cid_from_string(const char* s, uint8_t out40[IPCFP_CID_SLOT]);
int ipcfp_cid_to_slot(const uint8_t* cid, uint32_t len, uint8_t out40[IPCFP_CID_SLOT]);
int ipcfp_cid_to_string(const uint8_t* cid, uint32_t len, char* out, uint32_t cap);

/* ---- context ------------------------------------------------------------ */
typedef struct ipcfp_ctx ipcfp_ctx_t;

/* Binds one GPU (HIP device ordinal) and creates the context's stream.
 * Fails with IPCFP_E_NO_DEVICE when no HIP device is usable: there is NO CPU
 * fallback anywhere behind this ABI. */
int ipcfp_ctx_create(int device, ipcfp_ctx_t** out);
void ipcfp_ctx_destroy(ipcfp_ctx_t* ctx);
const char* ipcfp_last_error(const ipcfp_ctx_t* ctx);
const char* ipcfp_strerror(int rc);
int ipcfp_abi_version(void);
/* The context's hipStream_t (as void*), so a host can record its own events on
 * the stream the kernels are launched on. */
void* ipcfp_ctx_stream(ipcfp_ctx_t* ctx);
int ipcfp_ctx_sync(ipcfp_ctx_t* ctx);
/* Route selection (A/B measurements; tests that drive both routes of an entry point over one corpus).  Outcomes never
 * depend on it — every fast route only ever answers what the general one would.  Keys:
 *   "hamt_levels"  -1 default (ipcfp_hamt_get*: level by level for batches of >= 1024 queries), 0 per-query walker only,
 *                  k > 0 exactly k levels whatever the batch size
 *   "hamt_coop"    0: the level path decodes ActorState nodes with one lane each instead of sixteen
 *   "hamt_table"   1: tabulate every block of the witness as a HAMT node first
 *   "fast_verify"  0: verify_event_proof never takes the route without mid-call synchronisation                     */
int ipcfp_ctx_set_tuning(ipcfp_ctx_t* ctx, const char* key, int64_t value);
/* Device properties used by the benchmarks: name (≤ 63 chars), CU count, HBM bytes. */
int ipcfp_ctx_device_info(ipcfp_ctx_t* ctx, char name[64], int* cu_count, uint64_t* hbm_bytes);

/* Per-kernel timing with HIP events on the context's stream.  While enabled,
 * every launch of a profiled kernel is bracketed by an event pair;
 * ipcfp_profile_read() synchronises and returns the launch count and the summed
 * duration for one kernel id since the last reset.  ipcfp_profile_enable: on = 0
 * off, 1 every profiled kernel, 2 + id that kernel id alone (an event pair is a
 * barrier packet on the stream; a timed region that wants ONE kernel's duration
 * keeps the others' launches back to back). */
enum {
    IPCFP_K_BLAKE2B_CID = 0,
    IPCFP_K_KECCAK256 = 1,
    IPCFP_K_SHA256 = 2,
    IPCFP_K_CID_INDEX = 3,
    IPCFP_K_AMT_GET = 4,
    IPCFP_K_EVENT_SCAN = 5,
    IPCFP_K_HAMT_GET = 6,
    IPCFP_K_REPLAY = 7,
    IPCFP_K_EVENT_VERIFY = 8,
    IPCFP_K_STORAGE_VERIFY = 9,
    IPCFP_K_EXEC_ORDER = 10,
    IPCFP_K_BLAKE2B_RAW = 11,
    IPCFP_K_BASE64 = 12,
    IPCFP_K_ALLGATHER = 13, /* ipcfp_allgather_segments: message packing + ncclAllGather */
    IPCFP_K_TIPSET_PROLOGUE = 14, /* header decodes, TxMeta re-hash, AMT roots (events/verifier.rs:147-181, utils.rs:48-72) */
    IPCFP_K_AMT_WALK = 15,        /* the dense Amt::for_each of message lists + receipts (levels and leaves) */
    IPCFP_K_COUNT = 16
};
int ipcfp_profile_enable(ipcfp_ctx_t* ctx, int on);
int ipcfp_profile_reset(ipcfp_ctx_t* ctx);
int ipcfp_profile_read(ipcfp_ctx_t* ctx, int kernel_id, uint64_t* launches, double* total_ms);

/* ---- witness store -------------------------------------------------------
 * Replaces `MemoryBlockstore` + `load_witness_store`
 * (src/proofs/events/verifier.rs:79-89, src/proofs/storage/verifier.rs:68-78): the
 * CID → bytes map every verifier call walks.  The engine keeps the whole witness
 * resident in HBM as a structure of arrays:
 *     bytes[]            all block payloads, each block 16-byte aligned
 *     off[n], len[n]     the coalesced offset/length table
 *     cids[n][40]        claimed CIDs
 *     open-addressing hash table  CID → block id  (duplicate CID: last one wins,
 *                        as HashMap::insert does)
 * `bytes`/`off`/`len`/`cids` are borrowed host memory; the witness owns its
 * device copy.  Block i is bytes[off[i] .. off[i]+len[i]).                      */
typedef struct ipcfp_witness ipcfp_witness_t;

int ipcfp_witness_create(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                         const uint32_t* len, const uint8_t* cids40, uint64_t n, ipcfp_witness_t** out);
/* Same, but the four arrays are already resident in HBM on the context's device
 * (the timed benchmarks use this: inputs resident before the clock starts).     */
int ipcfp_witness_create_device(ipcfp_ctx_t* ctx, const void* bytes_d, uint64_t nbytes, const void* off_d,
                                const void* len_d, const void* cids40_d, uint64_t n, ipcfp_witness_t** out);
/* The same witness with its TABLES in transport form — what has to cross PCIe when the bundle comes from host memory
 * (window T2 of the benchmarks; the witness blocks of a bundle are `(Cid, Vec<u8>)` pairs in `Cid: Ord` order,
 * src/proofs/common/bundle.rs:10-16, src/proofs/common/witness.rs:34-54, and nearly every CID of a Filecoin witness is
 * CIDv1 dag-cbor blake2b-256):
 *   bytes / len      the blocks back to back: block i starts at len[0] + ... + len[i-1]  (no offset table: 8 bytes per block)
 *   digests32        the multihash digest of block i's CID; the CID is cid_prefix ‖ digest  (32 instead of 40 bytes per block)
 *   cid_prefix       prefix_len ≤ 8 bytes, e.g. 01 71 a0 e4 02 20
 *   esc_index / esc_cids40   the blocks whose CID is NOT of that form (ascending ids) with their 40-byte slots; the
 *                    digest rows of those blocks are ignored
 * Offsets and CID slots are rebuilt on the device; everything after that is ipcfp_witness_create.                  */
int ipcfp_witness_create_packed(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint32_t* len,
                                const uint8_t* digests32, uint64_t n, const uint8_t* cid_prefix, uint32_t prefix_len,
                                const uint32_t* esc_index, const uint8_t* esc_cids40, uint64_t n_esc,
                                ipcfp_witness_t** out);
void ipcfp_witness_destroy(ipcfp_witness_t* w);
uint64_t ipcfp_witness_block_count(const ipcfp_witness_t* w);
uint64_t ipcfp_witness_byte_count(const ipcfp_witness_t* w);

/* K1 — Blake2b-256 CID check of every witness block.
 * Replaces `Block::cid` / `put_cbor(.., Code::Blake2b256)` (src/proofs/events/utils.rs:65-72;
 * README.md:401 "CID verification").  The reference's MemoryBlockstore never
 * hashes witness blocks, so this is reported BESIDE the verdicts:
 *   status[i] ∈ {IPCFP_CID_MISMATCH, IPCFP_CID_OK, IPCFP_CID_UNCHECKED}   (host, n bytes, nullable)
 *   *n_bad    = number of IPCFP_CID_MISMATCH blocks                         (nullable)            */
int ipcfp_witness_verify_cids(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, uint8_t* status, uint64_t* n_bad);
/* Launch-only form: results stay in HBM; read them with the accessors below
 * after ipcfp_ctx_sync().                                                      */
int ipcfp_witness_verify_cids_async(ipcfp_ctx_t* ctx, ipcfp_witness_t* w);
/* Results of the last ipcfp_witness_verify_cids_async on this witness (waits for it): same outputs as
 * ipcfp_witness_verify_cids, without launching K1 again.                          */
int ipcfp_witness_cid_results(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, uint8_t* status, uint64_t* n_bad);
/* Device pointers owned by the witness: the ⌈n/32⌉-word OK bitmap (bit i of
 * word i/32 set ⇔ block i is IPCFP_CID_OK) — the buffer a multi-GPU host
 * all-gathers — and the n status bytes.                                        */
void* ipcfp_witness_cid_bitmap_device(ipcfp_witness_t* w);
void* ipcfp_witness_cid_status_device(ipcfp_witness_t* w);

/* ---- the witness as a Blockstore -------------------------------------------------------------------------------
 * `fvm_ipld_blockstore::Blockstore { get, put_keyed, has }` — the trait the reference's stores implement
 * (src/proofs/common/blockstore.rs:26-39, src/client/blockstore.rs:20-37, src/client/cached_blockstore.rs:53-85) and
 * every fvm_ipld_amt / fvm_ipld_hamt call goes through.  bindings/rust/ffi.rs implements the trait over these, so
 * unmodified AMT/HAMT callers can sit on the HBM-resident store.
 *   has        has[i] = 1 iff cids40[i] is in the witness (block_ids[i] = its id, 0xffffffff if absent; both nullable)
 *   get        Ok(None): *found = 0.  Ok(Some(v)): *found = 1, *len = v.len(), the first min(*len, cap) bytes in out
 *              (an owned copy, as the trait returns; call with cap = 0 to size the buffer)
 *   put_keyed  n blocks at once; `MemoryBlockstore::put_keyed` semantics: the data is NOT hashed, an existing CID is
 *              replaced.  The witness is re-laid out (block ids of existing blocks are kept, new ones are appended).  */
int ipcfp_witness_has(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* cids40, uint64_t n, uint8_t* has,
                      uint32_t* block_ids);
int ipcfp_witness_get(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* cid40, uint8_t* out, uint64_t cap,
                      uint64_t* len, int* found);
int ipcfp_witness_put_keyed(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* cids40, const uint8_t* bytes,
                            const uint64_t* off, const uint32_t* len, uint64_t n);

/* ---- batch hashes (host buffers in, host digests out) --------------------
 * message i = bytes[off[i] .. off[i]+len[i]); out32 receives n × 32 bytes.
 *   blake2b256 : multihash-codetable Code::Blake2b256       (events/utils.rs:65)
 *   keccak256  : sha3::Keccak256 — hash_event_signature / keccak256
 *                (common/evm.rs:62-69, 81-88)
 *   sha256     : fvm_ipld_hamt's key hasher (common/decode.rs:29-39)            */
int ipcfp_blake2b256_batch(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                           const uint32_t* len, uint64_t n, uint8_t* out32);
int ipcfp_keccak256_batch(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                          const uint32_t* len, uint64_t n, uint8_t* out32);
int ipcfp_sha256_batch(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                       const uint32_t* len, uint64_t n, uint8_t* out32);

/* ---- path-walk primitives --------------------------------------------------
 * A located value: the CBOR item of block `block` at [off, off+len).  `block` indexes the
 * witness's blocks in the order they were passed to ipcfp_witness_create.            */
typedef struct ipcfp_value_loc {
    uint32_t block; /* 0xffffffff when no value was located */
    uint32_t off;
    uint32_t len;
} ipcfp_value_loc_t;

/* element type an AMT / HAMT is opened with (decides how every value of a visited node is
 * type-checked, as serde does when it decodes the node) */
enum {
    IPCFP_V_CID = 0,           /* Amtv0<Cid>            events/utils.rs:76,84            */
    IPCFP_V_RECEIPT = 1,       /* Amtv0<Receipt>        events/verifier.rs:220           */
    IPCFP_V_STAMPED_EVENT = 2, /* Amt<StampedEvent>     events/verifier.rs:234           */
    IPCFP_V_ACTOR_STATE = 3,   /* Hamt<_, ActorState>   common/decode.rs:29              */
    IPCFP_V_VEC_U8 = 4,        /* Hamt<_, Vec<u8>>      storage/decode.rs:79,86,92       */
    IPCFP_V_ANY = 5            /* any well-formed item                                    */
};

/* Bytes of located values (from the walk primitives, the scan's matches or ipcfp_verify_event_proofs_located): value i
 * is copied to out[i * stride ..), truncated to stride; entries with block == 0xffffffff are skipped.            */
int ipcfp_witness_read_values(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_value_loc_t* locs, uint64_t n,
                              uint8_t* out, uint64_t stride);

/* K5 — `Amt::load(root).get(index[i])` for a batch of indices (version 0 = Amtv0, 3 = Amt).
 * status[i] ∈ {IPCFP_ST_TRUE, IPCFP_ST_NOT_FOUND, IPCFP_ST_ERR_*}; loc nullable.
 * Replaces src/proofs/events/verifier.rs:220-226,234-239; src/proofs/events/generator.rs:249. */
int ipcfp_amt_get(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* root_cid40, int version, int value_kind,
                  const uint64_t* index, uint64_t n, ipcfp_status_t* status, ipcfp_value_loc_t* loc);

/* K7 — `Hamt::load_with_bit_width(root, bit_width).get(key_i)`; key i = keys[key_off[i] .. +key_len[i]).
 * The SHA-256 key hash (K3) is computed in the same kernel.
 * Replaces src/proofs/common/decode.rs:29-39; src/proofs/storage/decode.rs:79-96.          */
int ipcfp_hamt_get(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* root_cid40, uint32_t bit_width,
                   int value_kind, const uint8_t* keys, const uint32_t* key_off, const uint32_t* key_len, uint64_t n,
                   ipcfp_status_t* status, ipcfp_value_loc_t* loc);

/* ipcfp_hamt_get with every buffer already resident in HBM (keys_d readable up to the last key's end + 16 bytes;
 * key_off_d / key_len_d u32[n]; status_d u8[n]; loc_d ipcfp_value_loc_t[n] or NULL).  Asynchronous on the context's
 * stream: results are complete after ipcfp_ctx_sync.  A multi-GPU host calls it for its query-index range and
 * all-gathers the status bytes (ipcfp_allgather_device) — the loop it cuts: src/proofs/common/decode.rs:29-39.     */
int ipcfp_hamt_get_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* root_cid40, uint32_t bit_width,
                          int value_kind, const void* keys_d, const void* key_off_d, const void* key_len_d, uint64_t n,
                          void* status_d, void* loc_d);

/* `reconstruct_execution_order(bs, parent_hdr_cids)` (src/proofs/events/utils.rs:16-30 →
 * collect_exec_list(verify_txmeta = true), :48-94): per parent header the TxMeta is loaded and
 * re-hashed (Blake2b-256 of its canonical encoding must reproduce the header's `messages` CID),
 * then the BLS and secp message AMTs are walked; a CID seen before is skipped.
 *   *status_out  IPCFP_ST_TRUE, or the ERR_* the reference's `?` would surface first
 *   *count       number of messages in execution order
 *   out_cids40   receives min(*count, cap) message CIDs in execution order (nullable)        */
int ipcfp_exec_order(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                     ipcfp_status_t* status_out, uint8_t* out_cids40, uint64_t cap, uint64_t* count);

/* The closure `create_event_filter(event_sig, subnet_id)` returns
 * (src/proofs/events/verifier.rs:28-39): topics.len() >= 2 && t[0]==topic0 && t[1]==topic1. */
typedef struct ipcfp_event_filter {
    uint8_t topic0[32]; /* Keccak-256(event signature) */
    uint8_t topic1[32]; /* ascii_to_bytes32(subnet id)  */
} ipcfp_event_filter_t;

/* K6 + K8 — `find_matching_events` (src/proofs/events/generator.rs:180-307) over the resident tipset.
 * The receipt list the reference obtains over RPC is the receipts AMT (Amtv0<Receipt>) walked in
 * index order.  PASS 1 marks every receipt that has at least one event passing the emitter filter
 * (`actor_id_filter`) and `matches_log`; PASS 2 emits one match per such event in
 * (exec_index, event_index) order — the fields an `EventProof` is built from.
 *   *status_out        IPCFP_ST_TRUE or the first ERR_* in traversal order
 *   receipt_has_match  one byte per receipt index (nullable); *n_receipts = number of indices
 *   matches            (nullable) up to cap_matches records; *n_matches = total
 *   touched_bits       (nullable) ⌈block_count/32⌉ words: bit b set ⇔ witness block b is in the union
 *                      of the RecordingBlockStores generate_event_proof collects for this step
 *                      (`take_seen`, src/proofs/common/blockstore.rs:21-30): rec_receipts plus one
 *                      rec_events per matching receipt.                                          */
typedef struct ipcfp_event_match {
    uint64_t exec_index;
    uint64_t event_index;
    uint64_t emitter;
    ipcfp_value_loc_t event; /* the StampedEvent item */
    uint32_t reserved;
} ipcfp_event_match_t;

int ipcfp_scan_events(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* receipts_root40,
                      const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, ipcfp_status_t* status_out,
                      uint8_t* receipt_has_match, uint64_t cap_receipts, uint64_t* n_receipts,
                      ipcfp_event_match_t* matches, uint64_t cap_matches, uint64_t* n_matches, uint32_t* touched_bits);

/* ipcfp_scan_events with DEVICE outputs: receipt_has_match_d (cap_receipts bytes) and matches_d (cap_matches
 * ipcfp_event_match_t) are HBM buffers of the caller (nullable); status and the two counts come back to the host.
 * summary_d (nullable): device u64[2] that receives {status | phase << 8, n_matches}, stream-ordered after the scan
 * (phase: IPCFP_SCAN_PHASE_* below, 0 with status TRUE).
 * A multi-GPU host all-gathers the map without a round trip through host memory (ipcfp_allgather_segments).     */
int ipcfp_scan_events_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* receipts_root40,
                             const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, ipcfp_status_t* status_out,
                             void* receipt_has_match_d, uint64_t cap_receipts, uint64_t* n_receipts, void* matches_d,
                             uint64_t cap_matches, uint64_t* n_matches, void* summary_d);

/* Where the Err of the last ipcfp_scan_events* call on this witness arose (0: it returned TRUE).  The scan enumerates the
 * receipts of the WHOLE tipset before it opens any events AMT (the enumeration stands in for the reference's
 * ChainGetParentReceipts call, src/proofs/events/generator.rs:199-204), so an Err of the enumeration precedes every Err
 * of the events passes whatever the receipt index.  Receipt-range shards each see a part of the enumeration: their
 * merge takes the first shard (in range order) with a RECEIPTS-phase Err, and only without one the first shard with
 * an EVENTS-phase Err — the unsharded call's verdict (INTEGRATION.md "Multi-GPU"). */
#define IPCFP_SCAN_PHASE_RECEIPTS 1
#define IPCFP_SCAN_PHASE_EVENTS 2
int ipcfp_witness_last_scan_phase(const ipcfp_witness_t* w);

/* ---- proof claims (string form, exactly the reference's structs) ----------
 * CIDs and hex values are NUL-terminated strings, as in the reference's serde
 * structs; the host parses them once per batch (src/proofs/common/witness.rs:60-72).
 * An unparsable string is not an ABI error: it yields the per-proof status the
 * reference's `?` would produce at the point it parses that field.             */

/* `EventProof` + `EventData` (src/proofs/events/bundle.rs:5-23) */
typedef struct ipcfp_event_proof {
    int64_t parent_epoch;
    int64_t child_epoch;
    const char* const* parent_tipset_cids; /* ordered tipset key of H */
    uint32_t n_parent_tipset_cids;
    const char* child_block_cid;
    const char* message_cid;
    uint64_t exec_index;
    uint64_t event_index;
    /* event_data */
    uint64_t emitter;
    const char* const* topics; /* "0x…" hex strings */
    uint32_t n_topics;
    const char* data; /* "0x…" hex string */
} ipcfp_event_proof_t;

/* `StorageProof` (src/proofs/storage/bundle.rs:5-14) */
typedef struct ipcfp_storage_proof {
    int64_t child_epoch;
    const char* child_block_cid;
    const char* parent_state_root;
    uint64_t actor_id;
    const char* actor_state_cid;
    const char* storage_root;
    const char* slot;  /* "0x" + 64 hex */
    const char* value; /* "0x" + 64 hex */
} ipcfp_storage_proof_t;

/* topic0 = Keccak-256(event_sig) is computed ON THE DEVICE (K2); topic1 is the
 * zero-padded ASCII of subnet_id (src/proofs/common/evm.rs:62-78).               */
int ipcfp_create_event_filter(ipcfp_ctx_t* ctx, const char* event_sig, const char* subnet_id,
                              ipcfp_event_filter_t* out);

/* ---- generator side -------------------------------------------------
 * `generate_event_proof` (src/proofs/events/generator.rs:75-178) against the resident tipset: the
 * witness `w` plays the RPC blockstore (every block the generator may load must be in it) and a
 * bitmap over its blocks plays `RecordingBlockStore` (src/proofs/common/blockstore.rs:9-37).
 *   parent_cids40/n_parents, child_cid40   the tipset pair (header CIDs, 40-byte slots)
 *   filter, has_actor, actor               EventMatcher + optional emitter filter (generator.rs:26-40,187)
 *   *status_out      IPCFP_ST_TRUE, or the ERR_* the reference's `?` surfaces first
 *   matches[i]       (exec_index, event_index, emitter, event location) of proof i, receipt order
 *   message_cids40   message CID of proof i = execution_order[exec_index] (generator.rs:152-169)
 *   witness_block_ids / witness_cids40   the materialised witness: ids of the recorded blocks of
 *                    `w` in `Cid: Ord` order (collect_witness_blocks, common/witness.rs:34-54)
 * Outputs are truncated to cap_*; *n_proofs / *n_blocks always receive the full counts.      */
int ipcfp_generate_event_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                                const uint8_t* child_cid40, const ipcfp_event_filter_t* filter, int has_actor,
                                uint64_t actor, ipcfp_status_t* status_out, ipcfp_event_match_t* matches,
                                uint8_t* message_cids40, uint64_t cap_proofs, uint64_t* n_proofs,
                                uint32_t* witness_block_ids, uint8_t* witness_cids40, uint64_t cap_blocks,
                                uint64_t* n_blocks);

/* One generated StorageProof (the claim fields of src/proofs/storage/generator.rs:158-178). */
typedef struct ipcfp_generated_storage {
    uint8_t parent_state_root[IPCFP_CID_SLOT];
    uint8_t actor_state_cid[IPCFP_CID_SLOT];
    uint8_t storage_root[IPCFP_CID_SLOT];
    uint8_t value[32];  /* left-padded slot value; all zero when the slot is absent */
    uint32_t status;    /* IPCFP_ST_TRUE or the ERR_* generate_storage_proof returns */
    uint32_t reserved;
} ipcfp_generated_storage_t;

/* `generate_storage_proof` (src/proofs/storage/generator.rs:29-69) for n (actor_id, slot) specs of one
 * child block, one lane per spec; the witness ids are the union of the blocks all specs recorded
 * (`generate_proof_bundle` dedupes the union, src/proofs/generator.rs:52-62), in `Cid: Ord` order. */
int ipcfp_generate_storage_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* child_cid40,
                                  const uint64_t* actor_ids, const uint8_t* slots32, uint64_t n,
                                  ipcfp_generated_storage_t* out, uint32_t* witness_block_ids, uint8_t* witness_cids40,
                                  uint64_t cap_blocks, uint64_t* n_blocks);

/* `generate_proof_bundle` (src/proofs/generator.rs:25-95) against the resident tipset: every StorageProofSpec, then
 * every EventProofSpec (:12-22), one bundle.  The RPC block cache the reference shares between the specs is the
 * witness itself; what remains is the union of the recorded blocks, which the reference keeps in a
 * `BTreeSet<(Cid, Vec<u8>)>` (:34,52-54,75-77,85-88): ids in `Cid: Ord` order, each block once.
 *   storage_out[i]           claim fields + status of storage spec i (ERR_*: generate_proof_bundle returns that Err)
 *   event_status[j]          IPCFP_ST_TRUE or the ERR_* of event spec j
 *   matches / message_cids40 / match_spec   the EventProofs of all specs in spec order (match_spec[k] = j)
 *   *first_error             index of the first failing spec in the reference's order (storage specs first, then
 *                            n_storage + j), or UINT64_MAX: the reference aborts there and returns nothing        */
typedef struct ipcfp_storage_proof_spec {
    uint64_t actor_id;
    uint8_t slot[32];
} ipcfp_storage_proof_spec_t;
typedef struct ipcfp_event_proof_spec {
    const char* event_signature; /* e.g. "NewTopDownMessage(bytes32,uint256)" */
    const char* topic_1;         /* subnet id; ascii_to_bytes32 (src/proofs/common/evm.rs:72-78) */
    uint64_t actor_id_filter;
    uint8_t has_actor_id_filter;
} ipcfp_event_proof_spec_t;
int ipcfp_generate_proof_bundle(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                                const uint8_t* child_cid40, const ipcfp_storage_proof_spec_t* storage_specs,
                                uint64_t n_storage, const ipcfp_event_proof_spec_t* event_specs, uint64_t n_events,
                                ipcfp_generated_storage_t* storage_out, ipcfp_status_t* event_status,
                                ipcfp_event_match_t* matches, uint8_t* message_cids40, uint32_t* match_spec,
                                uint64_t cap_proofs, uint64_t* n_proofs, uint32_t* witness_block_ids,
                                uint8_t* witness_cids40, uint64_t cap_blocks, uint64_t* n_blocks, uint64_t* first_error);

/* `TrustPolicy` (src/proofs/trust/mod.rs:8-16,53-78): the predicate is evaluated on the
 * host, before the device call, because it is a pure function of (epoch, cid).  */
typedef struct ipcfp_trust_policy {
    int kind;           /* 0 = AcceptAll, 1 = F3Certificate epoch range (cert.rs:52-64) */
    int ec_chain_empty; /* kind 1: certificate with an empty EC chain ⇒ nothing is trusted */
    int64_t min_epoch;  /* kind 1: first EC-chain epoch */
    int64_t max_epoch;  /* kind 1: last EC-chain epoch  */
} ipcfp_trust_policy_t;

/* `verify_event_proof` (src/proofs/events/verifier.rs:51-74) over a batch.
 *   status[i]            per-proof ipcfp_status_t (n bytes, host)
 *   filter               nullable — the built-in `check_event` closure form
 * The reference aborts at the first Err; callers reproduce that by scanning status[]
 * for the lowest index with IPCFP_ST_IS_ERR (bindings/rust/ffi.rs does).          */
int ipcfp_verify_event_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs, uint64_t n,
                              const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                              ipcfp_status_t* status);

/* The same, also reporting WHERE each proof's StampedEvent lies (event_loc[i].block == 0xffffffff when proof i did
 * not get that far).  This is the door for an ARBITRARY host closure `check_event: &dyn Fn(&ActorEvent) -> bool`
 * (src/proofs/events/verifier.rs:51-56, applied at :247-251): call with filter = NULL, read the located events of the
 * proofs whose status is TRUE (ipcfp_witness_read_values), run the closure on the host, and turn a `false` into
 * IPCFP_ST_FALSE_FILTER — bindings/rust/ffi.rs::verify_event_proof_with does exactly that.                      */
int ipcfp_verify_event_proofs_located(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs,
                                      uint64_t n, const ipcfp_trust_policy_t* trust,
                                      const ipcfp_event_filter_t* filter, ipcfp_status_t* status,
                                      ipcfp_value_loc_t* event_loc);

/* `verify_event_proof(bundle, .., Some(check_event))` with an ARBITRARY host closure, end to end behind the ABI
 * (src/proofs/events/verifier.rs:51-56; the predicate is applied at :247-251, AFTER verify_event_data_matches, and
 * only to proofs that are still true).  The device verifies the batch and locates each proof's StampedEvent; the
 * events of the proofs whose status is TRUE are gathered by one kernel and copied back, and `check_event` runs on
 * the calling thread, in proof order, over the DAG-CBOR bytes of the StampedEvent item `[emitter, ActorEvent]`
 * (already validated on the device): a zero return turns the status into IPCFP_ST_FALSE_FILTER (`Ok(false)`).
 * A Rust wrapper passes a trampoline that decodes the item and calls the `&dyn Fn(&ActorEvent) -> bool`.
 * check_event == NULL: identical to ipcfp_verify_event_proofs.                                                    */
typedef int (*ipcfp_check_event_fn)(void* user, uint64_t proof_index, const uint8_t* stamped_event, uint64_t len);
int ipcfp_verify_event_proofs_with(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs, uint64_t n,
                                   const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                                   ipcfp_check_event_fn check_event, void* user, ipcfp_status_t* status);

/* `verify_storage_proof` (src/proofs/storage/verifier.rs:24-63) over a batch. */
int ipcfp_verify_storage_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_storage_proof_t* proofs,
                                uint64_t n, const ipcfp_trust_policy_t* trust, ipcfp_status_t* status);

/* ---- packed (binary) claims: the device-resident fast path -----------------------------
 * The string entry points above parse each claim once and lower it to these structs; a host that
 * already holds binary CIDs (or verifies the same bundle repeatedly) can build them itself and keep
 * them in HBM.  Semantics are identical: the flags record what `Cid::try_from(str)?` and the
 * hex compares of the reference would have observed for the original strings.               */
/* Parent blocks a tipset key holds INLINE (expected 5 per epoch; 17 happens every few weeks).  A longer key — the
 * reference takes any (src/proofs/events/verifier.rs:147-181, src/proofs/events/utils.rs:16-30) — carries the rest behind
 * `more_parents`; such a tipset is verified by the general route (one workgroup per parent block, the level-by-level
 * enumerator) instead of the single-launch prologue and the dense walk, with the same verdicts.  Up to
 * IPCFP_MAX_PARENTS_WIDE parents (the enumeration's error word orders 2^16 sequence numbers). */
#define IPCFP_MAX_PARENTS 32
#define IPCFP_MAX_PARENTS_WIDE 16000

#define IPCFP_TIPSET_PARENTS_PARSED 1u /* every parent_tipset_cids[i] parses (events/verifier.rs:130) */
#define IPCFP_TIPSET_CHILD_PARSED 2u   /* child_block_cid parses (:131)                               */
typedef struct ipcfp_tipset_ref {
    uint32_t flags;
    uint32_t n_parents;
    uint8_t child[IPCFP_CID_SLOT];
    uint8_t parents[IPCFP_MAX_PARENTS][IPCFP_CID_SLOT]; /* the first min(n_parents, IPCFP_MAX_PARENTS) */
    const uint8_t* more_parents; /* n_parents > IPCFP_MAX_PARENTS: parents[IPCFP_MAX_PARENTS ..] as 40-byte slots in HOST
                                    memory, valid for the call; ignored (may be NULL) otherwise */
} ipcfp_tipset_ref_t;

#define IPCFP_CLAIM_MSG_PARSED 1u     /* message_cid parses (events/verifier.rs:193)                   */
#define IPCFP_CLAIM_DATA_MATCHABLE 2u /* event_data.data is "0x" + an even number of hex digits        */
typedef struct ipcfp_event_claim {
    int64_t parent_epoch;
    int64_t child_epoch;
    uint64_t exec_index;
    uint64_t event_index;
    uint64_t emitter;
    uint8_t message_cid[IPCFP_CID_SLOT];
    uint32_t tipset;     /* index into the ipcfp_tipset_ref_t table                                    */
    uint32_t flags;      /* IPCFP_CLAIM_*                                                               */
    uint32_t n_topics;   /* claimed topic count                                                        */
    uint32_t topics_off; /* blob offset of n_topics × 33 bytes: [1 if the string was "0x"+64 hex, topic[32]] */
    uint32_t data_off;   /* blob offset of the claimed data bytes                                      */
    uint32_t data_len;
} ipcfp_event_claim_t;

/* verify_event_proof over packed claims resident in HBM.  `tipsets` is a small HOST table;
 * claims_d / blob_d / status_d are DEVICE pointers (n claims, blob_len bytes, n status bytes).
 * Synchronous: returns after the status bytes are written.  A claim whose `tipset` index or blob offsets
 * fall outside n_tipsets / blob_len gets IPCFP_ST_ERR_BAD_CLAIM (never followed).                */
int ipcfp_verify_event_claims_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets,
                                     uint32_t n_tipsets, const void* claims_d, uint64_t n, const void* blob_d,
                                     uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                                     const ipcfp_event_filter_t* filter, void* status_d);

/* ipcfp_verify_event_claims_device FOLLOWED BY ipcfp_scan_events_device of tipsets[0]'s child block — its receipts AMT,
 * `HeaderLite.parent_message_receipts` (src/proofs/common/decode.rs:100-118) — in ONE call: the scan's tail is queued
 * behind the verify kernel and both results come back with one synchronisation.  Outputs and statuses are exactly those of
 * the two calls in that order (verify_event_proof: src/proofs/events/verifier.rs:51-74; find_matching_events:
 * src/proofs/events/generator.rs:180-307); *scan_status is the child header's error when there is no receipts root.   */
int ipcfp_verify_and_scan_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets, uint32_t n_tipsets,
                                 const void* claims_d, uint64_t n, const void* blob_d, uint64_t blob_len,
                                 const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* check_filter, void* status_d,
                                 const ipcfp_event_filter_t* scan_filter, int has_actor, uint64_t actor,
                                 ipcfp_status_t* scan_status, void* receipt_has_match_d, uint64_t cap_receipts,
                                 uint64_t* n_receipts, void* matches_d, uint64_t cap_matches, uint64_t* n_matches);

/* ---- event claims in TRANSPORT form: what crosses PCIe when the claims come from host memory (window T2) -----------
 * An `EventProof` (src/proofs/events/bundle.rs:5-23) lowered to ipcfp_event_claim_t + blob is ≈ 200 bytes; most of it
 * says the same thing a million times: the epochs and the tipset of every claim of one bundle, the six prefix bytes of
 * the message CID, 33-byte framing for a 32-byte topic, u64 / u32 fields for indices and lengths that fit in half.
 * The compact record keeps what differs from claim to claim (56 bytes + 32 per topic + the data bytes); the device
 * rebuilds ipcfp_event_claim_t[n] + blob from it (ipcfp_expand_event_claims_device), byte for byte what
 * the plain lowering produces with the claims' blob segments laid out in claim order.
 * Representable: exec_index, event_index < 2^32; n_topics ≤ 8; data_len < 65536; at most 256 distinct
 * (parent_epoch, child_epoch, tipset) groups; message CID = 01 71 a0 e4 02 20 ‖ digest (or not parsed at all).
 * ipcfp_compact_event_claims returns IPCFP_E_UNSUPPORTED for a batch that is not: the caller keeps the plain form.  */
typedef struct ipcfp_event_claim_group {
    int64_t parent_epoch;
    int64_t child_epoch;
    uint32_t tipset; /* index into the ipcfp_tipset_ref_t table */
    uint32_t reserved;
} ipcfp_event_claim_group_t;
#define IPCFP_COMPACT_MAX_GROUPS 256u
#define IPCFP_COMPACT_MAX_TOPICS 8u
typedef struct ipcfp_event_claim_compact {
    uint64_t emitter;
    uint32_t exec_index;
    uint32_t event_index;
    uint8_t message_digest[32]; /* message_cid = 01 71 a0 e4 02 20 ‖ message_digest (ignored without IPCFP_CLAIM_MSG_PARSED) */
    uint16_t data_len;
    uint8_t n_topics;    /* ≤ IPCFP_COMPACT_MAX_TOPICS */
    uint8_t topic_flags; /* bit t: topic t was "0x" + 64 hex digits (the plain form's per-topic flag byte) */
    uint8_t flags;       /* IPCFP_CLAIM_* */
    uint8_t group;       /* index into the group table */
    uint16_t reserved;
} ipcfp_event_claim_compact_t; /* 56 bytes; blob: per claim, in claim order, n_topics × 32 topic bytes, then data_len bytes */

/* Host-side conversion plain → compact (a caller that builds the compact form directly needs none).  `groups` holds
 * IPCFP_COMPACT_MAX_GROUPS entries; out / out_blob hold n records / cap_blob bytes (blob_len is always enough).       */
int ipcfp_compact_event_claims(const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob, uint64_t blob_len,
                               ipcfp_event_claim_group_t* groups, uint32_t* n_groups, ipcfp_event_claim_compact_t* out,
                               uint8_t* out_blob, uint64_t cap_blob, uint64_t* out_blob_len);
/* compact → plain ON THE DEVICE (all pointers but `groups` are device pointers): claims_out_d receives n
 * ipcfp_event_claim_t, blob_out_d (cap_blob ≥ cblob_len + 8·n bytes) their blob; *blob_len_out (host, nullable: then the
 * call does not synchronise) the blob's length.  A record whose group, topic count or blob segment is out of range
 * becomes a claim with tipset = 0xffffffff: IPCFP_ST_ERR_BAD_CLAIM when verified.                                   */
int ipcfp_expand_event_claims_device(ipcfp_ctx_t* ctx, const ipcfp_event_claim_group_t* groups, uint32_t n_groups,
                                     const void* compact_d, uint64_t n, const void* cblob_d, uint64_t cblob_len,
                                     void* claims_out_d, void* blob_out_d, uint64_t cap_blob, uint64_t* blob_len_out);
/* ipcfp_verify_event_claims with the claims in compact form in HOST memory: they cross PCIe beside the tipset's AMT
 * walk, are expanded on the device, verified; status bytes back (host).                                              */
int ipcfp_verify_event_claims_compact(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets,
                                      uint32_t n_tipsets, const ipcfp_event_claim_group_t* groups, uint32_t n_groups,
                                      const ipcfp_event_claim_compact_t* claims, uint64_t n, const uint8_t* cblob,
                                      uint64_t cblob_len, const ipcfp_trust_policy_t* trust,
                                      const ipcfp_event_filter_t* filter, ipcfp_status_t* status);

/* The same over packed claims in HOST memory (tipsets, claims, blob, status: host): upload, verify, status bytes
 * back — the PCIe-inclusive form of verify_event_proof for callers that hold binary claims
 * (src/proofs/events/verifier.rs:51-74).                                                           */
int ipcfp_verify_event_claims(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets,
                              uint32_t n_tipsets, const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob,
                              uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                              const ipcfp_event_filter_t* filter, ipcfp_status_t* status);
/* The claims of the receipts [receipt_lo, receipt_hi) out of a batch that is in exec_index order — the order
 * generate_event_proof emits them in (src/proofs/events/generator.rs:242-301) — and stays where it is in host memory: a rank
 * of a receipt cut takes its share with two binary searches, NO pass over the batch.  The records cross PCIe as they are;
 * the window of the blob they point into is found on the device and uploaded beside the walk; the offsets are rebased in
 * front of the verify kernel (a record that points outside the batch's blob: IPCFP_ST_ERR_BAD_CLAIM).  last_shard != 0:
 * the claims with exec_index >= receipt_hi are this rank's too (every claim has one owner: ipcfp_route_event_claims).
 *   *first_out, *count_out   the records verified are claims[*first_out .. *first_out + *count_out)
 *   status                   receives *count_out bytes (room for n)
 * IPCFP_E_INVALID when the batch is not in exec_index order: at the slice's ends (seen on the host), or anywhere INSIDE
 * the slice — a record below its predecessor or outside the rank's range — which the device checks where the records are
 * anyway (no verdicts are returned then: the binary searches may have routed claims to the wrong rank).  A batch in another
 * order goes through ipcfp_route_event_claims.                                                                       */
int ipcfp_verify_event_claims_range(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets,
                                    uint32_t n_tipsets, const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob,
                                    uint64_t blob_len, uint64_t receipt_lo, uint64_t receipt_hi, int last_shard,
                                    const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                                    uint64_t* first_out, uint64_t* count_out, ipcfp_status_t* status);

/* Host-only lowering of the reference's structs to the packed form (no context, no device; parallel over
 * claims): what ipcfp_verify_event_proofs does before its upload, for callers that keep claims packed or
 * resident in HBM and call ipcfp_verify_event_claims_device.  The handle owns the three arrays. */
typedef struct ipcfp_packed_events ipcfp_packed_events_t;
int ipcfp_pack_event_proofs(const ipcfp_event_proof_t* proofs, uint64_t n, ipcfp_packed_events_t** out);
void ipcfp_packed_events_destroy(ipcfp_packed_events_t* p);
const ipcfp_tipset_ref_t* ipcfp_packed_events_tipsets(const ipcfp_packed_events_t* p, uint32_t* n);
const ipcfp_event_claim_t* ipcfp_packed_events_claims(const ipcfp_packed_events_t* p, uint64_t* n);
const uint8_t* ipcfp_packed_events_blob(const ipcfp_packed_events_t* p, uint64_t* len);

#define IPCFP_SCLAIM_CHILD_PARSED 1u        /* child_block_cid parses (storage/verifier.rs:85)              */
#define IPCFP_SCLAIM_STATE_ROOT_CANON 2u    /* parent_state_root parses and equals its Cid::to_string()     */
#define IPCFP_SCLAIM_ACTOR_STATE_CANON 4u
#define IPCFP_SCLAIM_STORAGE_ROOT_CANON 8u
#define IPCFP_SCLAIM_SLOT_PARSED 16u        /* slot hex decodes to exactly 32 bytes (:155-157)              */
#define IPCFP_SCLAIM_VALUE_MATCHABLE 32u    /* value is "0x" + 64 hex digits                                */
typedef struct ipcfp_storage_claim {
    int64_t child_epoch;
    uint64_t actor_id;
    uint8_t child[IPCFP_CID_SLOT];
    uint8_t state_root[IPCFP_CID_SLOT];
    uint8_t actor_state[IPCFP_CID_SLOT];
    uint8_t storage_root[IPCFP_CID_SLOT];
    uint8_t slot[32];
    uint8_t value[32];
    uint32_t flags; /* IPCFP_SCLAIM_* */
    uint32_t reserved;
} ipcfp_storage_claim_t;

/* Host-only, parallel lowering of n StorageProof structs to packed claims (what ipcfp_verify_storage_proofs does
 * before its upload), written to the caller's array of n ipcfp_storage_claim_t. */
int ipcfp_pack_storage_proofs(const ipcfp_storage_proof_t* proofs, uint64_t n, ipcfp_storage_claim_t* claims);

/* verify_storage_proof over packed claims resident in HBM (claims_d: n structs, status_d: n bytes). */
int ipcfp_verify_storage_claims_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const void* claims_d, uint64_t n,
                                       const ipcfp_trust_policy_t* trust, void* status_d);

/* Rebuild the CID → block index of an existing witness in place (K4), e.g. once per verification
 * pass when the index build is to be charged to that pass.  No allocation.                    */
int ipcfp_witness_rebuild_index(ipcfp_ctx_t* ctx, ipcfp_witness_t* w);

/* ---- one proof batch over the GPUs of a node ----------------------------------------------
 * The reference verifies a bundle sequentially, proof by proof (src/proofs/verifier.rs:19-28,49-54;
 * src/proofs/events/verifier.rs:62-71).  Given a read-only witness the proofs are independent: rank r of G takes a
 * contiguous range of the units (blocks / receipts / keys / claims), verifies it with the entry points above on
 * its own GPU, and ONE all-gather of the per-rank verdict bytes / bitmaps closes the step.                     */

/* [lo, hi) of shard `shard` when n units are cut into n_shards contiguous ranges (sizes differ by at most one). */
void ipcfp_shard_range(uint64_t n, uint32_t n_shards, uint32_t shard, uint64_t* lo, uint64_t* hi);

/* Which blocks of `w` — a witness holding a WHOLE tipset — does shard `shard` need to scan the receipts
 * [*receipt_lo, *receipt_hi) and to verify the EventProof claims about them?  Found like the reference's generator
 * finds a witness, with a recorder (src/proofs/common/blockstore.rs:26-30):
 *   every rank   child header, parent headers, their TxMeta and message AMTs (the execution order is global:
 *                reconstruct_execution_order, src/proofs/events/utils.rs:16-30), the receipts-AMT root;
 *   this rank    the receipts-AMT nodes on the paths to its receipts and those receipts' events AMTs.
 *   *status_out  IPCFP_ST_TRUE, or the ERR_* the traversal met first (nothing else is then written)
 *   *n_receipts  (nullable) the receipts AMT's count;  block_ids: ascending ids, truncated to cap_blocks
 * The ranges are cut on the root's count, which `Amtv0::load` checks against nothing (a root that says 572 over 700
 * receipts answers get(650)): whatever the tree holds beyond the count is the LAST shard's, as a claim beyond it is
 * (ipcfp_route_event_claims) — its plan and ipcfp_witness_create_shard_pull's pull run to the end of the tree, and its
 * witness is to be tagged [*receipt_lo, UINT64_MAX) (the pull does so itself).                                    */
int ipcfp_shard_plan_tipset(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                            const uint8_t* child_cid40, uint32_t n_shards, uint32_t shard, ipcfp_status_t* status_out,
                            uint64_t* receipt_lo, uint64_t* receipt_hi, uint64_t* n_receipts, uint32_t* block_ids,
                            uint64_t cap_blocks, uint64_t* n_blocks);

/* PLAN ONCE, SCATTER.  Every shard's plan in ONE call (one pass over the receipts instead of n_shards): run it where
 * the whole witness is resident — the bundle's producer, or one rank — and hand rank r the list
 * block_ids[shard_off[r] .. shard_off[r+1]) (ascending ids; the replicated blocks are in every list) and its receipts
 * [receipt_bounds[r], receipt_bounds[r+1]).  Rank r then uploads ONLY its blocks (ipcfp_witness_cut_host +
 * ipcfp_witness_create + ipcfp_witness_set_receipt_range) and its claims (ipcfp_route_event_claims): the G PCIe links
 * of a node carry G different shards instead of G copies of the bundle.  Cuts the same loops as above.
 *   receipt_bounds, shard_off   u64[n_shards + 1] each;  block_ids truncated to cap_ids, *n_ids = shard_off[n_shards]
 *   *status_out                 IPCFP_ST_TRUE, or the ERR_* the traversal met first (the lists are then empty)       */
#define IPCFP_MAX_SHARDS 64
int ipcfp_shard_plan_tipset_all(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                                const uint8_t* child_cid40, uint32_t n_shards, ipcfp_status_t* status_out,
                                uint64_t* n_receipts, uint64_t* receipt_bounds, uint64_t* shard_off, uint32_t* block_ids,
                                uint64_t cap_ids, uint64_t* n_ids);

/* SELF-PLANNED SHARD.  Rank `shard` of n_shards builds its shard of ONE tipset straight out of the bundle in ITS host
 * memory: no rank ever holds the whole witness in HBM, no plan is made elsewhere, the host cuts no block lists.  The
 * bundle is given in the transport form of ipcfp_witness_create_packed (blocks back to back, lengths, 32-byte digests +
 * the CID prefix they share, escapes for other CID forms).  What crosses PCIe: the tables (36 bytes per block of the
 * bundle) and the shard's own blocks — which the DEVICE reads out of `bytes` itself, level by level along the links
 * (child header → receipts AMT paths to the rank's receipts → their events AMTs; parent headers → TxMeta → message AMTs,
 * replicated: reconstruct_execution_order is global, src/proofs/events/utils.rs:16-30), so `bytes` must be host memory
 * the device can read: hipHostMalloc'd, or registered once (hipHostRegister / ipcfp_host_register below — an ingest buffer
 * is registered when it is made, not per bundle: 19 ms for 640 MB on the MI355X box).  `digests32` MAY be such memory too:
 * the table is then read where it lies instead of being copied first.  Cuts the loops of
 * src/proofs/verifier.rs:19-28,49-54 and src/proofs/events/verifier.rs:62-71; the shard equals what
 * ipcfp_shard_plan_tipset + ipcfp_witness_create_subset make of the whole witness.
 *   *status_out   IPCFP_ST_TRUE: *out is the shard, tagged [*receipt_lo, *receipt_hi) = ipcfp_shard_range(count);
 *                 IPCFP_ST_ERR_MISSING_BLOCK: the child header or the receipts root is not in the bundle or does not
 *                 decode — there is no receipt range to cut by (*out = NULL; verify on the whole bundle instead)
 *   stats         (nullable) what the call moved and how long its phases took on the host's clock               */
typedef struct ipcfp_shard_pull_stats {
    uint32_t rounds;      /* link levels followed                                          */
    uint32_t blocks;      /* blocks of the shard                                           */
    uint64_t table_bytes; /* bytes of the bundle's tables uploaded                         */
    uint64_t block_bytes; /* bytes of blocks read over PCIe (each block padded to 128)     */
    uint64_t payload_bytes; /* … of which the blocks themselves (the sum of their lengths) */
    double tables_ms;     /* tables up, CID slots, index                                   */
    double pull_ms;       /* the rounds                                                    */
    double create_ms;     /* the shard's own arena, schedule and index                     */
} ipcfp_shard_pull_stats_t;
int ipcfp_witness_create_shard_pull(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint32_t* len,
                                    const uint8_t* digests32, uint64_t n, const uint8_t* cid_prefix, uint32_t prefix_len,
                                    const uint32_t* esc_index, const uint8_t* esc_cids40, uint64_t n_esc,
                                    const uint8_t* parent_cids40, uint32_t n_parents, const uint8_t* child_cid40,
                                    uint32_t n_shards, uint32_t shard, ipcfp_status_t* status_out, uint64_t* receipt_lo,
                                    uint64_t* receipt_hi, uint64_t* n_receipts, ipcfp_shard_pull_stats_t* stats,
                                    ipcfp_witness_t** out);
/* hipHostRegister / hipHostUnregister for hosts that do not link HIP themselves (a Rust caller's ingest buffer).
 * The buffer must OWN its pages: `p` on a page boundary (IPCFP_E_INVALID otherwise) in a mapping of its own (mmap /
 * posix_memalign of whole pages), never a slice of the heap and never memory that is also handed to the runtime as the
 * pageable source of a copy.  Registration is by page: a page shared with another user (the runtime's own pin of a
 * pageable source among them) loses its device mapping for that user when this one is unregistered, and a later,
 * unrelated copy meets a GPU memory fault (seen in this repository's own suite: DESIGN.md §13).                     */
int ipcfp_host_register(void* p, uint64_t bytes);
int ipcfp_host_unregister(void* p);

/* Host only (no context, no device): blocks block_ids[0..n) of a witness that lies in HOST memory, packed back to back
 * as a witness of their own — out_off[i] / out_len[i] / out_cids40[i] describe block block_ids[i], its bytes at
 * out_bytes + out_off[i].  *nbytes_out = the payload size; call with every out pointer NULL to size the buffers.
 * IPCFP_E_INVALID (nothing written) when an id is >= n_src or a block lies outside [0, nbytes).                    */
int ipcfp_witness_cut_host(const uint8_t* bytes, uint64_t nbytes, const uint64_t* off, const uint32_t* len,
                           const uint8_t* cids40, uint64_t n_src, const uint32_t* block_ids, uint64_t n,
                           uint8_t* out_bytes, uint64_t cap_bytes, uint64_t* out_off, uint32_t* out_len,
                           uint8_t* out_cids40, uint64_t* nbytes_out);

/* A new witness made of blocks block_ids[0..n) of `src` (device-side copy; block i of the new witness is block
 * block_ids[i] of src), tagged as the receipt-range shard [receipt_lo, receipt_hi): ipcfp_scan_events walks only
 * those receipts (receipt_has_match[i - receipt_lo] is receipt i) and ipcfp_verify_event_* resolves them by table.
 * Pass (0, UINT64_MAX) for an untagged subset (e.g. a block-range shard of a CID batch), and UINT64_MAX as the LAST
 * receipt-range shard's receipt_hi (see ipcfp_shard_plan_tipset).                                               */
int ipcfp_witness_create_subset(ipcfp_ctx_t* ctx, ipcfp_witness_t* src, const uint32_t* block_ids, uint64_t n,
                                uint64_t receipt_lo, uint64_t receipt_hi, ipcfp_witness_t** out);
/* Tag / read the receipt range of a witness created by other means (drops its cached enumerations).             */
int ipcfp_witness_set_receipt_range(ipcfp_witness_t* w, uint64_t lo, uint64_t hi);
void ipcfp_witness_receipt_range(const ipcfp_witness_t* w, uint64_t* lo, uint64_t* hi);

/* Host only: the packed claims a receipt-range shard verifies — exec_index in [receipt_lo, receipt_hi); the LAST shard
 * (last_shard != 0) also owns every claim whose exec_index is >= receipt_hi, so that each claim has exactly one owner
 * (such a claim names no receipt: steps 1-3 of verify_single_proof, src/proofs/events/verifier.rs:92-204, settle it on
 * data every rank holds).  The shard's claims keep their order and get a blob of their own (topics_off / data_off
 * rewritten); positions[k] (nullable) = index of out_claims[k] in `claims`, which is where its status byte belongs
 * when the shards' verdicts are merged.  Call with every out pointer NULL to size (*n_out claims, *blob_out bytes).  */
int ipcfp_route_event_claims(const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob, uint64_t blob_len,
                             uint64_t receipt_lo, uint64_t receipt_hi, int last_shard, uint64_t* positions,
                             ipcfp_event_claim_t* out_claims, uint64_t cap_claims, uint8_t* out_blob, uint64_t cap_blob,
                             uint64_t* n_out, uint64_t* blob_out);

/* The collective: RCCL's ncclAllGather over xGMI, called directly (librccl.so.1 is resolved when the first
 * communicator is made; single-GPU hosts never need it).  One process per GPU; rank 0 makes the id
 * (ncclGetUniqueId) and the HOST application carries its 128 bytes to the other ranks over whatever channel it has. */
#define IPCFP_COMM_ID_BYTES 128
typedef struct ipcfp_comm ipcfp_comm_t;
int ipcfp_comm_unique_id(uint8_t id[IPCFP_COMM_ID_BYTES]);
int ipcfp_comm_create(ipcfp_ctx_t* ctx, const uint8_t id[IPCFP_COMM_ID_BYTES], int n_ranks, int rank,
                      ipcfp_comm_t** out);
void ipcfp_comm_destroy(ipcfp_comm_t* comm);
int ipcfp_comm_rank(const ipcfp_comm_t* comm);
int ipcfp_comm_size(const ipcfp_comm_t* comm);
/* recv_d[r * bytes_per_rank ..] = rank r's send_d[0 .. bytes_per_rank), on the context's stream (asynchronous:
 * ipcfp_ctx_sync or a later synchronous call completes it).  Waits for K1's stream first, so a CID bitmap written
 * by ipcfp_witness_verify_cids_async may be part of the payload.                                                  */
int ipcfp_allgather_device(ipcfp_ctx_t* ctx, ipcfp_comm_t* comm, const void* send_d, void* recv_d,
                           uint64_t bytes_per_rank);

/* The usual step payload lives in several buffers (status bytes, has-match map, CID bitmap): they are packed back to
 * back into staging_d (bytes_per_rank bytes, zero padded) on the context's stream and all-gathered in the same call —
 * still ONE collective.  comm may be null (or of size 1): the packed message is then also the result.            */
int ipcfp_allgather_segments(ipcfp_ctx_t* ctx, ipcfp_comm_t* comm, const void* const* seg_d, const uint64_t* seg_bytes,
                             uint32_t n_seg, void* staging_d, void* recv_d, uint64_t bytes_per_rank);

/* ---- bundle wire format ----------------------------------------------
 * `UnifiedProofBundle` (src/proofs/common/bundle.rs:39-45) as the JSON serde_json writes for the
 * derived structs: {"storage_proofs":[StorageProof…],"event_proofs":[EventProof…],"blocks":[{"cid":
 * [bytes…],"data":"<base64>"}…]}.  Parsing follows `#[derive(Deserialize)]`: unknown fields are
 * skipped, a missing or duplicate field, a wrong type or invalid base64 is IPCFP_E_PARSE (the
 * reference's `Err` before any proof is looked at).  The host locates the strings; every block's
 * base64 (`deserialize_base64`, bundle.rs:30-37) is decoded on the device into the witness arena.  */
typedef struct ipcfp_bundle ipcfp_bundle_t;
#define IPCFP_BUNDLE_CID_STRINGS 1u /* also accept "cid":"bafy…" (extension; serde_json writes a byte array) */
int ipcfp_bundle_parse_json(ipcfp_ctx_t* ctx, const char* json, uint64_t len, uint32_t flags, ipcfp_bundle_t** out);
/* Host half of the parse only (no context, no device): structure, field types, string escapes and base64
 * lengths; the CONTENT of `cid` arrays and `data` strings is checked on the device by the call above.
 * err_out (nullable) receives the first problem as text. */
int ipcfp_bundle_check_json(const char* json, uint64_t len, uint32_t flags, uint64_t* n_storage, uint64_t* n_events,
                            uint64_t* n_blocks, char* err_out, uint32_t err_cap);
void ipcfp_bundle_destroy(ipcfp_bundle_t* b);
ipcfp_witness_t* ipcfp_bundle_witness(ipcfp_bundle_t* b); /* owned by the bundle */
uint64_t ipcfp_bundle_block_count(const ipcfp_bundle_t* b);
uint64_t ipcfp_bundle_event_count(const ipcfp_bundle_t* b);
uint64_t ipcfp_bundle_storage_count(const ipcfp_bundle_t* b);
const ipcfp_event_proof_t* ipcfp_bundle_event_proofs(const ipcfp_bundle_t* b);     /* strings owned by the bundle */
const ipcfp_storage_proof_t* ipcfp_bundle_storage_proofs(const ipcfp_bundle_t* b);
/* `verify_proof_bundle` (src/proofs/verifier.rs:12-62): every storage proof, then every event proof,
 * against the bundle's own blocks.  One status per proof; the reference returns Err at the first
 * status >= 64 in that order (storage first) and otherwise the two Vec<bool> (status == 1). */
int ipcfp_verify_proof_bundle(ipcfp_ctx_t* ctx, ipcfp_bundle_t* b, const ipcfp_trust_policy_t* trust,
                              const ipcfp_event_filter_t* filter, ipcfp_status_t* storage_status,
                              ipcfp_status_t* event_status);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* IPCFP_H */
