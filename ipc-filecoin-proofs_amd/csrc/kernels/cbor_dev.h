// csrc/kernels/cbor_dev.h — strict DAG-CBOR reader for gfx950 device code.
//
// Device counterpart of every `serde_ipld_dagcbor::from_slice` /
// `fvm_ipld_encoding::from_slice` the reference performs on witness blocks
// (src/proofs/common/decode.rs:26,81,90,122; src/proofs/storage/decode.rs:46-85;
// src/proofs/events/utils.rs:25,61; src/proofs/events/verifier.rs:158,174,217) and of
// the typed values fvm_ipld_amt / fvm_ipld_hamt decode inside nodes.
//
// Rules (identical to oracle/cbor.hpp so parity is well defined):
// definite lengths only; tag 42 only, payload a byte string starting 0x00 holding ONE
// well-formed CID; text must be UTF-8; simple values false/true/null; floats 64-bit only;
// non-minimal integer encodings accepted; trailing bytes after the top-level item are an
// error (finish()).
//
// Error model: the reader is STICKY — the first violation stores a status code in
// `err` (IPCFP_ST_ERR_DECODE) and every later call is a no-op returning 0, so walk code
// reads straight-line and checks `err` where the reference's `?` would return.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ipcfp.h"
#include "witness_dev.h"

namespace ipcfp {

// UTF-8 validity of a short text item.  A FREE function, deliberately NOT inlined: with this loop
// inlined into Rd::read_text, hipcc 7.2 (AMD clang 22.0.0git) mis-structurises the caller's control
// flow on gfx950 — the success path of read_text picks up the zero the failure paths assign to `off`
// (reproduced stand-alone; the LLVM IR is correct, the emitted ISA is not).  Text items on this path
// are short map / event-entry keys.
__device__ __noinline__ bool utf8_ok_bytes(const uint8_t* s, uint32_t len) {
    uint32_t i = 0;
    while (i < len) {
        const uint32_t c = s[i];
        if (c < 0x80) {
            ++i;
            continue;
        }
        uint32_t need, cp;
        if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1F; }
        else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07; }
        else return false;
        if (need > len - i - 1) return false;
        for (uint32_t k = 1; k <= need; ++k) {
            const uint32_t cc = s[i + k];
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

// Per-lane line staging through LDS (see Rd::staged_chunk) is a per-translation-unit choice: it pays where
// lanes parse whole blocks front to back (k_verify_events: 0.58 → 0.46 ms) and costs where access is sparse or
// occupancy matters more (the enumeration levels, scan pass 1: +8 %), so only verify_events.hip turns it on.
#ifndef IPCFP_LINE_STAGE
#define IPCFP_LINE_STAGE 0
#endif

// IPCFP_RD_LDS = 1 (a per-translation-unit choice as well): every reader of that unit parses bytes that a wavefront
// staged in LDS (block_events.hip).  The pointers are typed into the LDS address space, so a chunk load is ONE
// ds_read_b128 (≈64 cycles) instead of a flat / global load; nothing else in the reader changes.
#ifndef IPCFP_RD_LDS
#define IPCFP_RD_LDS 0
#endif
#if IPCFP_RD_LDS
#define IPCFP_RD_AS __attribute__((address_space(3)))
#if IPCFP_LINE_STAGE
#error "IPCFP_LINE_STAGE stages global lines in LDS; a reader that already sits in LDS has nothing to stage"
#endif
#else
#define IPCFP_RD_AS
#endif

typedef unsigned long long rd_chunk_t __attribute__((ext_vector_type(2)));  // one 16-byte window chunk

// IPCFP_RD_RING = G (a mode of the LDS reader, per translation unit; G = 8: hamt_table.hip): a GROUP of G lanes drives ONE
// reader in lockstep over an item that stays in HBM.  The group streams it through a 1 KB ring in LDS: every step the G
// lanes fetch the next 2·G chunks (256 bytes for G = 8) with one coalesced load each and hold them in registers while
// the parser works on what is already in the ring; when the parser runs off the end of the ring the registers are
// written to LDS and the step after is requested.  So a sequential parse of a block of ANY size costs one memory
// latency per 256 bytes, most of it hidden behind the parse of the previous 256 — where one lane with a 16-byte window
// pays a latency per 16 bytes — for 1 KB of LDS per query.  The ring keeps the last ≥ 768 bytes: a reader may look back
// that far (a bucket's key after its value, a link's bytes after its header); further back is an error of the CALLER,
// reported as err = kRdRingLost so that it can never pass for a decode result.
#ifndef IPCFP_RD_RING
#define IPCFP_RD_RING 0
#endif
#if IPCFP_RD_RING && !IPCFP_RD_LDS
#error "IPCFP_RD_RING is a mode of the LDS reader"
#endif
constexpr uint32_t kRdRingLost = 0xfdu;  // not an ipcfp_status_t

// A CID LONGER than the 40-byte slot as a witness key: the fold  ff | len | blake2b-256(the CID's bytes) | 00 …  of
// include/ipcfp.h ("CIDs").  The reference stores and finds blocks under CIDs of any length
// (src/proofs/common/witness.rs:60-72: `Cid::try_from`, digests of up to 64 bytes) and compares message CIDs of any length
// (src/proofs/events/utils.rs:76-90, src/proofs/events/verifier.rs:193-201); folded, a long CID is found, compared and
// deduplicated through the same five words as a short one.  m[]: the CID's bytes as little-endian words, zero padded
// (len ≤ 128: one compression).  NOT inlined: one copy per kernel, however many readers it has (no Filecoin chain has such a CID).
static __device__ __noinline__ CidKey long_cid_fold(const uint64_t* __restrict__ m_in, uint32_t len) {
    // A ROLLED Blake2b-256 compression with its state in arrays that are indexed at run time — i.e. in scratch memory, on
    // purpose: this is the coldest path of the library, and a kernel's register allocation is the maximum over everything
    // it can call.  The unrolled compression of blake2b_dev.h here cost every caller ≈ 100 VGPRs whether or not it ever met
    // a long link (k_receipt_events 31 → 100, k_hamt_lv_advance 57 → 100, k_verify_storage_table 83 → 108: one wavefront
    // per SIMD less in each; profiles/r06_experiments.md); this form needs about two dozen.
    const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                            0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    // sigma rows as sixteen nibbles each, entry 0 in the low nibble (RFC 7693 §2.7; rounds 10, 11 repeat rows 0, 1)
    const uint64_t sigma[12] = {0xfedcba9876543210ULL, 0x357b20c16df984aeULL, 0x491763eadf250c8bULL, 0x8f04a562ebcd1397ULL,
                                0xd386cb1efa427509ULL, 0x91ef57d438b0a6c2ULL, 0xb8293670a4def15cULL, 0xa2684f05931ce7bdULL,
                                0x5a417d2c803b9ef6ULL, 0x0dc3e9bf5167482aULL, 0xfedcba9876543210ULL, 0x357b20c16df984aeULL};
    uint64_t v[16], m[16];
#pragma nounroll
    for (uint32_t i = 0; i < 16; ++i) m[i] = m_in[i];
#pragma nounroll
    for (uint32_t i = 0; i < 8; ++i) {
        v[i] = iv[i];
        v[8 + i] = iv[i];
    }
    v[0] ^= 0x01010020ULL;  // digest length 32, no key, fanout 1, depth 1
    v[12] ^= uint64_t(len);  // t0 = bytes compressed; one block: the last
    v[14] = ~v[14];
    auto rotr = [](uint64_t x, uint32_t n) { return (x >> n) | (x << (64u - n)); };
#pragma nounroll
    for (uint32_t r = 0; r < 12; ++r) {
        const uint64_t sg = sigma[r];
#pragma nounroll
        for (uint32_t g = 0; g < 8; ++g) {
            const uint32_t i = g & 3u, diag = g >> 2;  // columns, then diagonals
            const uint32_t a = i, b = 4u + ((i + diag) & 3u), c = 8u + ((i + 2u * diag) & 3u), d = 12u + ((i + 3u * diag) & 3u);
            const uint64_t x = m[(sg >> (8u * g)) & 15u], y = m[(sg >> (8u * g + 4u)) & 15u];
            uint64_t va = v[a], vb = v[b], vc = v[c], vd = v[d];
            va = va + vb + x;
            vd = rotr(vd ^ va, 32);
            vc = vc + vd;
            vb = rotr(vb ^ vc, 24);
            va = va + vb + y;
            vd = rotr(vd ^ va, 16);
            vc = vc + vd;
            vb = rotr(vb ^ vc, 63);
            v[a] = va;
            v[b] = vb;
            v[c] = vc;
            v[d] = vd;
        }
    }
    uint64_t h[4];
#pragma nounroll
    for (uint32_t i = 0; i < 4; ++i) h[i] = iv[i] ^ v[i] ^ v[8 + i];
    h[0] ^= 0x01010020ULL;
    CidKey k;
    k.w[0] = 0xffULL | (uint64_t(len & 0xffu) << 8) | (h[0] << 16);
    k.w[1] = (h[0] >> 48) | (h[1] << 16);
    k.w[2] = (h[1] >> 48) | (h[2] << 16);
    k.w[3] = (h[2] >> 48) | (h[3] << 16);
    k.w[4] = h[3] >> 48;
    return k;
}

struct Rd {
    const IPCFP_RD_AS uint8_t* p;
    uint32_t n;
    uint32_t pos;
    uint32_t err;
    // Byte access goes through a window that is filled 16 aligned bytes at a time.  Parsing is a chain of
    // dependent loads in which every lane of a wavefront reads a DIFFERENT block, so each load instruction
    // is 64 separate line requests to the texture-address unit: what the walk kernels pay for is the NUMBER
    // of load instructions, not the bytes.  One 16-byte load per chunk is half the instructions of a window
    // fed 8 bytes at a time (and ≈12× fewer than single bytes).  The window is the current chunk (lo, hi)
    // plus the high word of the chunk before it (ph): a parser only moves forward, so when an 8-byte peek
    // straddles into the next chunk the low word of the old chunk is never needed again.  Blocks sit
    // line-aligned in the arena with tail slack, so the chunk after the one that holds an item's last
    // byte never leaves the arena.
    const IPCFP_RD_AS rd_chunk_t* base16;  // p rounded down to 16 bytes
    uint32_t bias;             // p - base16
    uint32_t cwi;              // index of the current chunk (0xfffffff0: none)
    uint32_t phi;              // index of the chunk whose high word is in `ph` (0xffffffff: none)
    uint64_t lo, hi, ph;
    bool stage;                // chunk loads go through the lane's LDS line slot (IPCFP_LINE_STAGE builds only)

    __device__ __forceinline__ void init(const IPCFP_RD_AS uint8_t* data, uint32_t len) {
        p = data;
        n = len;
        pos = 0;
        err = 0;
        const uintptr_t a = (uintptr_t)data;
        base16 = (const IPCFP_RD_AS rd_chunk_t*)(a & ~uintptr_t(15));
        bias = uint32_t(a & 15);
        cwi = 0xfffffff0u;
        phi = 0xffffffffu;
        lo = hi = ph = 0;
        stage = IPCFP_LINE_STAGE != 0;
        if (IPCFP_LINE_STAGE) stage_invalidate();  // LDS holds garbage (or another workgroup's lines) until written
    }
    // The window state as plain values.  at()/peek64() copy the members into one of these on entry and
    // store them back on exit, so that every member is read and written UNCONDITIONALLY: always_inline
    // functions are optimised on their own before they are inlined, and there LLVM merges `hi` loaded on
    // one path with `lo` loaded on another into a single load from a phi of two addresses — after
    // inlining that variable offset keeps the whole reader (and everything behind it) in scratch memory.
    struct Win {
        uint32_t cwi, phi;
        uint64_t lo, hi, ph;
        bool stage;
    };
    // make chunk ci the current one (moving forward by one chunk keeps the old high word)
    __device__ __forceinline__ static void slide(Win& w, const IPCFP_RD_AS rd_chunk_t* base16, uint32_t ci) {
        if (ci == w.cwi) return;
        const bool next = ci == w.cwi + 1;
        w.ph = next ? w.hi : w.ph;
        w.phi = next ? w.cwi : 0xffffffffu;
#if IPCFP_LINE_STAGE
        if (w.stage) {
            const ulonglong2 v = staged_chunk(reinterpret_cast<const ulonglong2*>(base16 + ci));
            w.lo = v.x;
            w.hi = v.y;
        } else
#endif
        {
            const rd_chunk_t v = base16[ci];
            w.lo = v.x;
            w.hi = v.y;
        }
        w.cwi = ci;
    }
    // ---- per-lane line staging (IPCFP_LINE_STAGE) -------------------------------------------------------
    // A lane walks its block 16 bytes at a time, and with ~1300 lanes per CU sharing a 16 KB L1 the line is
    // gone again before the lane asks for its next chunk: every chunk is an L2 (or HBM) round trip and the
    // same 128-byte line crosses L2→L1 up to eight times (k_verify_events: FETCH_SIZE 1.94 GB for a 0.44 GB
    // witness).  With staging, the first touch of a line fetches the REST of the line (chunks k..7, issued
    // back to back: one latency, one L2→L1 transfer) into the lane's LDS slot, and the following chunks are
    // LDS reads.  Layout [chunk][lane] — a wavefront reading the same chunk index is conflict-free.  The
    // slot's tag (line address | first valid chunk) lives in LDS as well, so any number of readers on one lane
    // (a node reader, an item reader, a copy rewound to an earlier offset) share the slot safely: a reader that
    // finds another line there simply refills.
    struct StageLds {
        ulonglong2 chunk[8][256];
        unsigned long long tag[256];
    };
    __device__ __forceinline__ static StageLds& stage_lds() {
        __shared__ StageLds s;
        return s;
    }
    __device__ __forceinline__ static void stage_invalidate() { stage_lds().tag[threadIdx.x & 255u] = ~0ull; }
    // slow path, deliberately NOT inlined: it is reached once per half line, and inlined at every peek it costs
    // each walk kernel ≈30 VGPRs (an occupancy step)
    __device__ __attribute__((noinline)) static void stage_refill(unsigned long long line, uint32_t k, uint32_t lane) {
        StageLds& s = stage_lds();
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(line);
        // the rest of the line in half-line bursts (four 16-byte loads in flight): the upper half always, the
        // lower half only when the reader starts inside it
        if (k < 4) {
            const ulonglong2 t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
            s.chunk[0][lane] = t0;
            s.chunk[1][lane] = t1;
            s.chunk[2][lane] = t2;
            s.chunk[3][lane] = t3;
        }
        {
            const ulonglong2 t4 = src[4], t5 = src[5], t6 = src[6], t7 = src[7];
            s.chunk[4][lane] = t4;
            s.chunk[5][lane] = t5;
            s.chunk[6][lane] = t6;
            s.chunk[7][lane] = t7;
        }
        s.tag[lane] = line | (k < 4 ? 0ull : 4ull);
    }
#if IPCFP_LINE_STAGE == 2
    // ---- IPCFP_LINE_STAGE == 2: the slot is a WINDOW — the eight chunks from any 16-byte boundary on (tag = that address;
    // chunk c sits in slot c mod 8) — and the parser says where it is about to read (ensure_span).  With line-aligned slots
    // a lane meets the end of its line somewhere inside an element loop, and of the 64 lanes of a wavefront SOME lane does
    // at nearly every fetch: a wavefront of k_hamt_node_table_lane made ≈ 80 refills per node, each a round trip to the L2
    // with the other lanes waiting, and lived 260 µs whatever its instruction count was (profiles/r06_experiments.md).
    // One call per pointer / per bucket entry takes all lanes' refills at the same place: ≈ 10 per node.
    __device__ __attribute__((noinline)) static void stage_refill_window(unsigned long long a, uint32_t lane) {
        StageLds& s = stage_lds();
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(a);
        const uint32_t k = uint32_t(a >> 4);
        const ulonglong2 t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3], t4 = src[4], t5 = src[5], t6 = src[6], t7 = src[7];
        s.chunk[(k + 0u) & 7u][lane] = t0;
        s.chunk[(k + 1u) & 7u][lane] = t1;
        s.chunk[(k + 2u) & 7u][lane] = t2;
        s.chunk[(k + 3u) & 7u][lane] = t3;
        s.chunk[(k + 4u) & 7u][lane] = t4;
        s.chunk[(k + 5u) & 7u][lane] = t5;
        s.chunk[(k + 6u) & 7u][lane] = t6;
        s.chunk[(k + 7u) & 7u][lane] = t7;
        s.tag[lane] = a;
    }
    __device__ __forceinline__ static ulonglong2 staged_chunk(const ulonglong2* addr) {
        StageLds& s = stage_lds();
        const uint32_t lane = threadIdx.x & 255u;
        const unsigned long long a = reinterpret_cast<unsigned long long>(addr);
        if (a - s.tag[lane] >= 128ull) stage_refill_window(a, lane);  // (an invalid tag is ~0: a + 1)
        return s.chunk[uint32_t(a >> 4) & 7u][lane];
    }
    // the next `span` bytes from the reader's position are about to be read: have them in the window (span ≤ 113)
    __device__ __forceinline__ void ensure_span(uint32_t span) {
        const uint32_t lane = threadIdx.x & 255u;
        const unsigned long long a = reinterpret_cast<unsigned long long>(base16 + ((pos + bias) >> 4));
        const unsigned long long tag = stage_lds().tag[lane];
        const unsigned long long end = reinterpret_cast<unsigned long long>(base16) + bias + pos + span;
        if (pos < n && (a - tag >= 128ull || end > tag + 128ull)) stage_refill_window(a, lane);
    }
#else
    __device__ __forceinline__ void ensure_span(uint32_t) {}
    __device__ __forceinline__ static ulonglong2 staged_chunk(const ulonglong2* addr) {
        StageLds& s = stage_lds();
        const uint32_t lane = threadIdx.x & 255u;
        const unsigned long long a = reinterpret_cast<unsigned long long>(addr);
        const unsigned long long line = a & ~127ull;
        const uint32_t k = uint32_t(a >> 4) & 7u;
        const unsigned long long tag = s.tag[lane];
        if ((tag & ~127ull) != line || k < uint32_t(tag & 7ull)) stage_refill(line, k, lane);
        return s.chunk[k][lane];
    }
#endif
    __device__ __forceinline__ Win win() const { return Win{cwi, phi, lo, hi, ph, stage}; }
    __device__ __forceinline__ void keep(const Win& w) {
        cwi = w.cwi;
        phi = w.phi;
        lo = w.lo;
        hi = w.hi;
        ph = w.ph;
    }
#if IPCFP_RD_RING
    // ---- the ring (see IPCFP_RD_RING above).  Coordinates are BIASED: byte i of the item is ring byte i + bias ----
    static constexpr uint32_t kRingBytes = 1024u, kRingStep = uint32_t(IPCFP_RD_RING) * 32u;
    static_assert(kRingBytes % kRingStep == 0 && kRingStep >= 128u, "a step must divide the ring");
    const rd_chunk_t* g16;    // HBM: the 16-byte chunk that holds the item's first byte
    uint32_t rhi;             // biased bytes committed so far: the ring holds [rhi - kRingBytes, rhi)
    uint32_t rlo;             // … of which the bytes from rlo on are really there (a forward jump leaves a hole behind it)
    uint32_t rlim;            // biased bytes worth fetching (the item, the reader's 16 bytes of slack, rounded up to a step)
    rd_chunk_t pend0, pend1;  // this lane's two chunks of the step [rhi, rhi + kRingStep), in flight
    struct RingState {
        uint32_t rhi, rlo;
        rd_chunk_t pend0, pend1;
    };
    // The slow path — commit the step in flight, request the next, until byte e is in — is deliberately NOT inlined
    // (by value in, by value out): inlined at every at()/peek64() it costs the walk kernels ≈70 VGPRs, i.e. one or two
    // of the three or four wavefronts per SIMD that hide its own memory latency.
    __device__ __attribute__((noinline)) static RingState ring_fill(IPCFP_RD_AS uint8_t* ring, const rd_chunk_t* g16, uint32_t rlim,
                                                                    uint32_t e, RingState st) {
        const uint32_t sub = threadIdx.x & (uint32_t(IPCFP_RD_RING) - 1u);
        IPCFP_RD_AS rd_chunk_t* ring16 = (IPCFP_RD_AS rd_chunk_t*)(uintptr_t)(ring);
        if (e > st.rhi + 4u * kRingStep && st.rhi < rlim) {  // a long jump forward (a skipped byte string): do not stream what nobody reads
            const uint32_t target = (e - 1u) / kRingStep * kRingStep;
            st.rhi = target < rlim ? target : rlim - kRingStep;
            st.rlo = st.rhi;
            st.pend0 = g16[(st.rhi >> 4) + sub];
            st.pend1 = g16[(st.rhi >> 4) + uint32_t(IPCFP_RD_RING) + sub];
        }
        while (st.rhi < e && st.rhi < rlim) {
            const uint32_t c = (st.rhi >> 4) & (kRingBytes / 16u - 1u);  // a multiple of 2·G: no wrap inside a step
            __builtin_amdgcn_wave_barrier();
            ring16[c + sub] = st.pend0;
            ring16[c + uint32_t(IPCFP_RD_RING) + sub] = st.pend1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            st.rhi += kRingStep;
            if (st.rhi > kRingBytes && st.rlo < st.rhi - kRingBytes) st.rlo = st.rhi - kRingBytes;
            if (st.rhi < rlim) {
                st.pend0 = g16[(st.rhi >> 4) + sub];
                st.pend1 = g16[(st.rhi >> 4) + uint32_t(IPCFP_RD_RING) + sub];
            }
        }
        return st;
    }
    // eight bytes at BIASED offset j of the ring, any lane its own j (the caller has ring_ensure'd, group-uniformly)
    __device__ __forceinline__ uint64_t ring_raw64(uint32_t j) const {
        const IPCFP_RD_AS uint64_t* q = (const IPCFP_RD_AS uint64_t*)(uintptr_t)(p);
        const uint32_t wi = (j >> 3) & (kRingBytes / 8u - 1u);
        const uint64_t w0 = q[wi], w1 = q[(wi + 1u) & (kRingBytes / 8u - 1u)];
        const uint32_t sh = (j & 7u) * 8u;
        return (w0 >> sh) | ((w1 << 1) << (63u - sh));
    }
    // bytes below biased offset e are in the ring (or e lies beyond everything worth fetching)
    __device__ __forceinline__ void ring_ensure(uint32_t e) {
        if (rhi < e && rhi < rlim) {
            const RingState st = ring_fill((IPCFP_RD_AS uint8_t*)(uintptr_t)(p), g16, rlim, e, RingState{rhi, rlo, pend0, pend1});
            rhi = st.rhi;
            rlo = st.rlo;
            pend0 = st.pend0;
            pend1 = st.pend1;
        }
    }
    // every lane of the group with the same arguments.  `ring`: kRingBytes of LDS, 16-byte aligned, the group's own.
    __device__ __forceinline__ void init_ring(const uint8_t* item, uint32_t len, IPCFP_RD_AS uint8_t* ring) {
        p = ring;
        n = len;
        pos = 0;
        err = 0;
        const uintptr_t a = (uintptr_t)item;
        g16 = (const rd_chunk_t*)(a & ~uintptr_t(15));
        bias = uint32_t(a & 15);
        base16 = nullptr;
        cwi = phi = 0;
        lo = hi = ph = 0;
        stage = false;
        rlim = (bias + len + 16u + kRingStep - 1u) / kRingStep * kRingStep;
        rhi = rlo = 0;
        {
            const uint32_t sub = threadIdx.x & (uint32_t(IPCFP_RD_RING) - 1u);
            pend0 = g16[sub];
            pend1 = g16[uint32_t(IPCFP_RD_RING) + sub];
        }
        ring_ensure(1u);
    }
    __device__ __forceinline__ uint32_t at(uint32_t i) {
        const uint32_t j = i + bias;
        ring_ensure(j + 1u);
        if (j < rlo) err = err ? err : kRdRingLost;
        return p[j & (kRingBytes - 1u)];
    }
    __device__ __forceinline__ uint64_t peek64(uint32_t i) {
        const uint32_t j = i + bias;
        ring_ensure(j + 16u);
        if ((j & ~7u) < rlo) err = err ? err : kRdRingLost;
        return ring_raw64(j);
    }
    __device__ __forceinline__ void peek128(uint32_t i, uint64_t& w0, uint64_t& w1) {
        w0 = peek64(i);
        w1 = peek64(i + 8u);
    }
#elif IPCFP_RD_LDS
    // A reader that sits in LDS needs no window: a byte is one ds_read_u8, eight unaligned bytes one ds_read_b64
    // (gfx950 serves unaligned LDS accesses).  The window above exists to cut the NUMBER of global load instructions;
    // what it costs is instructions — ≈10 k VALU per wavefront of 64 receipts in k_scan_pass1 / k_event_table, which
    // is what those kernels were bound by (profiles/r01_final_pmc.txt: 15 % of the wave cycles wait for memory).
    __device__ __forceinline__ uint32_t at(uint32_t i) { return p[i]; }
    // Eight bytes at any offset = the two ALIGNED words around them (one ds_read2_b64), funnel-shifted.  gfx950 does
    // serve an unaligned ds_read_b64, but by replaying it: SQ_LDS_UNALIGNED_STALL was 6x the LDS instruction cycles
    // of k_block_events with the one-instruction form (profiles/r02_pmc_block_events.txt).
    __device__ __forceinline__ uint64_t peek64(uint32_t i) {
        const uint32_t a = uint32_t(uintptr_t(p + i));
        const IPCFP_RD_AS uint64_t* q = (const IPCFP_RD_AS uint64_t*)(uintptr_t(a & ~7u));
        const uint64_t lo = q[0], hi = q[1];
        const uint32_t sh = (a & 7u) * 8u;
        return (lo >> sh) | ((hi << 1) << (63u - sh));
    }
    // sixteen bytes at any offset as two words: three aligned words
    __device__ __forceinline__ void peek128(uint32_t i, uint64_t& w0, uint64_t& w1) {
        const uint32_t a = uint32_t(uintptr_t(p + i));
        const IPCFP_RD_AS uint64_t* q = (const IPCFP_RD_AS uint64_t*)(uintptr_t(a & ~7u));
        const uint64_t q0 = q[0], q1 = q[1], q2 = q[2];
        const uint32_t sh = (a & 7u) * 8u;
        w0 = (q0 >> sh) | ((q1 << 1) << (63u - sh));
        w1 = (q1 >> sh) | ((q2 << 1) << (63u - sh));
    }
#else
    // byte i of the item (i < n, or inside the block's padded tail)
    __device__ __forceinline__ uint32_t at(uint32_t i) {
        const uint32_t j = i + bias;
        const uint32_t ci = j >> 4;
        const bool high = (j & 8u) != 0;
        Win w = win();
        const bool from_prev = ci == w.phi && high;
        if (!from_prev) slide(w, base16, ci);
        const uint64_t m = high ? ~0ull : 0ull;
        const uint64_t cur = (w.hi & m) | (w.lo & ~m);
        const uint64_t word = from_prev ? w.ph : cur;
        keep(w);
        return uint32_t(word >> ((j & 7u) * 8u)) & 0xffu;
    }
    // the 8 bytes at [i, i+8) as a little-endian u64 (unaligned)
    __device__ __forceinline__ uint64_t peek64(uint32_t i) {
        const uint32_t j = i + bias;
        const uint32_t ci = j >> 4;
        const uint32_t sh = (j & 7u) * 8u;
        Win w = win();
        uint64_t first, second;
        if ((j & 8u) == 0) {  // both words inside chunk ci
            slide(w, base16, ci);
            first = w.lo;
            second = w.hi;
        } else {              // high word of chunk ci, low word of chunk ci + 1
            if (ci == w.phi) {
                first = w.ph;  // the current chunk is ci + 1
            } else {
                slide(w, base16, ci);
                first = w.hi;
                slide(w, base16, ci + 1);
            }
            second = w.lo;
        }
        keep(w);
        // (second << 1) << (63 - sh) is second << (64 - sh), and 0 for sh == 0: no branch on sh
        return (first >> sh) | ((second << 1) << (63u - sh));
    }
    __device__ __forceinline__ void peek128(uint32_t i, uint64_t& w0, uint64_t& w1) {
        w0 = peek64(i);
        w1 = peek64(i + 8u);
    }
#endif
    // the CID bytes [off, off+len) as a witness key (len ≤ 40): five unaligned words, tail masked
    __device__ __forceinline__ CidKey key_at(uint32_t off, uint32_t len) {
        CidKey k;
#pragma unroll
        for (int w = 0; w < 5; ++w) {
            const uint32_t lo = 8u * w;
            uint64_t v = 0;
            if (lo < len) {
                v = peek64(off + lo);
                const uint32_t valid = len - lo;  // bytes of this word that belong to the CID
                if (valid < 8) v &= (1ULL << (8u * valid)) - 1ULL;
            }
            k.w[w] = v;
        }
        return k;
    }
    // … and of a CID of 41 .. IPCFP_CID_MAX_LEN bytes: its fold (long_cid_fold above)
    __device__ __forceinline__ CidKey key_long(uint32_t off, uint32_t len) {
        uint64_t m[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t lo = 8u * w;
            uint64_t v = 0;
            if (lo < len) {
                v = peek64(off + lo);
                const uint32_t valid = len - lo;
                if (valid < 8) v &= (1ULL << (8u * valid)) - 1ULL;
            }
            m[w] = v;
        }
        return long_cid_fold(m, len);
    }
    // the CID bytes [off, off+len) of a link that read_link accepted, as a witness key
    __device__ __forceinline__ CidKey key_any(uint32_t off, uint32_t len) {
        if (len <= 40u) return key_at(off, len);
        if (len > 128u) return CidKey{{~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL}};  // (cid_ok admits no such CID)
        return key_long(off, len);
    }
    // 32 bytes at `off` equal to q[0..32)?  (q: any alignment)
    __device__ __forceinline__ bool equal32(uint32_t off, const uint8_t* q) {
        uint64_t diff = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint64_t e;
            __builtin_memcpy(&e, q + 8 * w, 8);  // one unaligned 8-byte load (gfx950 runs in unaligned-access mode)
            diff |= peek64(off + 8u * w) ^ e;
        }
        return diff == 0;
    }
    // n bytes at `off` equal to q[0..n)?  (q: any alignment)
    __device__ __forceinline__ bool equal_bytes(uint32_t off, const uint8_t* q, uint32_t n) {
        uint64_t diff = 0;
        for (uint32_t i = 0; i < n; i += 8) {
            const uint32_t valid = n - i;
            uint64_t e = 0;
            if (valid >= 8) __builtin_memcpy(&e, q + i, 8);  // one unaligned 8-byte load
            else
                for (uint32_t k = 0; k < valid; ++k) e |= uint64_t(q[i + k]) << (8u * k);  // never read past q + n
            uint64_t d = peek64(off + i) ^ e;
            if (valid < 8) d &= (1ull << (8u * valid)) - 1ull;
            diff |= d;
        }
        return diff == 0;
    }
    __device__ __forceinline__ void fail() {
        if (!err) err = IPCFP_ST_ERR_DECODE;
    }
    __device__ __forceinline__ bool ok() const { return err == 0; }

    // next byte without consuming; 0xff (never a valid start we rely on) when failed / at end
    __device__ __forceinline__ uint32_t peek() {
        if (err) return 0xff;
        if (pos >= n) {
            fail();
            return 0xff;
        }
        return at(pos);
    }

    // item header → major type, argument
#if IPCFP_RD_LDS
    // In LDS there is nothing to protect a failed reader from: the fetch is clamped into the item and the outcome is
    // selected, so that the whole header decode is straight-line code — a lane whose reader has failed, or sits on
    // another kind of item, costs its wavefront no branch.  Same results as the version below.
    __device__ __forceinline__ void head(uint32_t& major, uint64_t& arg) {
        const bool in = pos < n;
        const uint32_t at_pos = in ? pos : 0u;
        const uint64_t raw = peek64(at_pos);
        const uint32_t b = uint32_t(raw) & 0xffu;
        const uint32_t m = b >> 5, ai = b & 31u;
        const bool imm = ai < 24;
        const uint32_t nb = imm ? 0u : (1u << ((ai - 24u) & 3u));
        bool bad = !in || ai > 27;
        bad |= m == 7 && (imm ? !(ai >= 20 && ai <= 22) : ai != 27);
        bad |= in && nb > n - at_pos - 1;
        uint64_t be = __builtin_bswap64(raw >> 8);
        be |= nb == 8 ? uint64_t(at(at_pos + 8)) : 0ull;  // (reads inside the stage: the item plus its slack)
        const uint64_t v = imm ? uint64_t(ai) : (nb == 8 ? be : (be >> ((64u - 8u * nb) & 63u)));
        const bool good = !err && !bad;
        err = err ? err : (bad ? uint32_t(IPCFP_ST_ERR_DECODE) : 0u);
        pos += good ? 1u + nb : 0u;
        major = good ? m : 8u;
        arg = good ? v : 0ull;
    }
#else
    __device__ __forceinline__ void head(uint32_t& major, uint64_t& arg) {
        major = 8;  // invalid
        arg = 0;
        if (err) return;
        if (pos >= n) return fail();
        // one unaligned fetch covers the initial byte and up to 8 argument bytes minus one; the
        // argument is big-endian.  Straight-line on purpose: lanes of a wavefront sit on different
        // items, and a branchy decoder makes the wave execute every path.
        const uint64_t raw = peek64(pos);
        const uint32_t b = uint32_t(raw) & 0xffu;
        const uint32_t m = b >> 5, ai = b & 31u;
        const bool imm = ai < 24;
        const uint32_t nb = imm ? 0u : (1u << ((ai - 24u) & 3u));  // 1, 2, 4, 8 argument bytes
        bool bad = ai > 27;                                          // indefinite length / reserved
        bad |= m == 7 && (imm ? !(ai >= 20 && ai <= 22) : ai != 27);
        bad |= nb > n - pos - 1;
        if (bad) return fail();
        uint64_t v = ai;
        if (!imm) {
            // argument bytes are raw bytes 1..nb (little-endian positions) → big-endian value
            uint64_t be = __builtin_bswap64(raw >> 8);            // bytes 1..7 → top of the word
            if (nb == 8) be |= uint64_t(at(pos + 8));             // the 8th argument byte lies beyond the fetch
            v = nb == 8 ? be : (be >> (64u - 8u * nb));
        }
        pos += 1 + nb;
        major = m;
        arg = v;
    }
#endif

    __device__ __forceinline__ uint64_t read_uint() {
        uint32_t m;
        uint64_t a;
        head(m, a);
        if (err) return 0;
        if (m != 0) {
            fail();
            return 0;
        }
        return a;
    }
    // i64 (major 0 or 1)
    __device__ __forceinline__ long long read_int() {
        uint32_t m;
        uint64_t a;
        head(m, a);
        if (err) return 0;
        if ((m != 0 && m != 1) || a > 0x7fffffffffffffffULL) {
            fail();
            return 0;
        }
        return m == 0 ? (long long)a : -1 - (long long)a;
    }
    // byte string → offset of its first byte (relative to p) and length
    __device__ __forceinline__ void read_bytes(uint32_t& off, uint32_t& len) {
        off = len = 0;
        uint32_t m;
        uint64_t a;
        head(m, a);
        if (err) return;
        if (m != 2 || a > uint64_t(n - pos)) return fail();
        off = pos;
        len = uint32_t(a);
        pos += len;
    }
    // UTF-8 validity of the text at [off, off+len): short ASCII keys ("t1", "d", "topics", "root" …)
    // are settled from the window; anything else takes the out-of-line validator.
    __device__ __forceinline__ bool text_ok(uint32_t off, uint32_t len) {
        if (len <= 8) {  // all ASCII?
#if IPCFP_RD_LDS
            const uint64_t m = len == 8 ? ~0ull : ((1ull << (8u * len)) - 1ull);  // one fetch, the rest masked off
            if (len == 0 || (peek64(off) & m & 0x8080808080808080ull) == 0) return true;
#else
            uint32_t hi = 0;
            for (uint32_t i = 0; i < len; ++i) hi |= at(off + i);
            if (hi < 0x80) return true;
#endif
        }
#if IPCFP_RD_RING
        if (!err) err = kRdRingLost;  // (a long or non-ASCII text is not linear in the ring: the caller's one-lane path decides)
        return false;
#else
        return utf8_ok_bytes((const uint8_t*)(p + off), len);  // (an LDS reader hands out the generic address)
#endif
    }
    __device__ __forceinline__ void read_text(uint32_t& off, uint32_t& len) {
        off = len = 0;
        uint32_t m;
        uint64_t a;
        head(m, a);
        if (err) return;
        if (m != 3 || a > uint64_t(n - pos)) return fail();
        if (!text_ok(pos, uint32_t(a))) return fail();
        off = pos;
        len = uint32_t(a);
        pos += len;
    }
    __device__ __forceinline__ uint64_t read_array() {
        uint32_t m;
        uint64_t a;
        head(m, a);
        if (err) return 0;
        if (m != 4) {
            fail();
            return 0;
        }
        return a;
    }
    __device__ __forceinline__ void expect_array(uint64_t len) {
        const uint64_t a = read_array();
        if (!err && a != len) fail();
    }
    __device__ __forceinline__ uint64_t read_map() {
        uint32_t m;
        uint64_t a;
        head(m, a);
        if (err) return 0;
        if (m != 5) {
            fail();
            return 0;
        }
        return a;
    }
    __device__ __forceinline__ bool at_null() { return !err && pos < n && at(pos) == 0xf6; }
    __device__ __forceinline__ void read_null() {
        if (err) return;
        if (pos >= n || at(pos) != 0xf6) return fail();
        ++pos;
    }

    // one well-formed binary CID in p[off, off+len)?  (oracle/cid.hpp cid_parse_binary)
    __device__ __forceinline__ bool cid_ok(uint32_t off, uint32_t len) {
        if (len == 34 && at(off) == 0x12 && at(off + 1) == 0x20) return true;  // CIDv0
        uint32_t q = off;
        const uint32_t end = off + len;
        uint64_t field[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            uint64_t v = 0;
            bool done = false;
            for (int shift = 0; shift < 63; shift += 7) {
                if (q >= end) return false;
                const uint32_t c = at(q);
                ++q;
                v |= uint64_t(c & 0x7f) << shift;
                if (!(c & 0x80)) {
                    if (c == 0 && shift > 0) return false;  // non-minimal varint
                    done = true;
                    break;
                }
            }
            if (!done) return false;
            field[f] = v;
        }
        if (field[0] != 1) return false;        // version
        if (field[3] > 64) return false;        // multihash size
        return uint64_t(end - q) == field[3];   // digest fills the rest exactly
    }
    // tag-42 link → offset/len of the CID bytes (without the 0x00 prefix)
    __device__ __forceinline__ void read_link(uint32_t& off, uint32_t& len) {
        off = len = 0;
        // The standard 43-byte link — d8 2a | 58 27 | 00 | 01 71 a0 e4 02 20 | digest[32]: tag 42, a 39-byte string, the
        // identity multibase byte, CIDv1 dag-cbor blake2b-256 — passes every check below; seeing it is two compares
        // instead of two item headers and four varints (≈40 instructions instead of ≈250).  Nearly every link of a
        // Filecoin witness has this form: a 32-link HAMT node is 8 k instructions the long way.
        if (!err && pos + 43u <= n && peek64(pos) == 0xa071010027582ad8ull && (peek64(pos + 8u) & 0xffffffull) == 0x2002e4ull) {
            off = pos + 5u;
            len = 38u;
            pos += 43u;
            return;
        }
        uint32_t m;
        uint64_t a;
        head(m, a);
        if (err) return;
        if (m != 6 || a != 42) return fail();
        uint32_t bo, bl;
        read_bytes(bo, bl);
        if (err) return;
        if (bl < 1 || at(bo) != 0x00) return fail();
        if (!cid_ok(bo + 1, bl - 1)) return fail();
        off = bo + 1;
        len = bl - 1;
    }
    // the link as a witness key (a CID longer than the 40-byte slot: its fold — key_long)
    __device__ __forceinline__ bool read_link_key(CidKey& key) {
        uint32_t off, len;
        read_link(off, len);
        if (err) return false;
        key = key_any(off, len);
        return true;
    }

    // IgnoredAny: skip exactly one well-formed item (iterative; maps count 2 items per pair)
    __device__ __forceinline__ void skip() {
        uint64_t todo = 1;
        while (todo && !err) {
            --todo;
            uint32_t m;
            uint64_t a;
            head(m, a);
            if (err) return;
            switch (m) {
                case 0:
                case 1:
                case 7:
                    break;
                case 2:
                    if (a > uint64_t(n - pos)) return fail();
                    pos += uint32_t(a);
                    break;
                case 3:
                    if (a > uint64_t(n - pos)) return fail();
                    if (!text_ok(pos, uint32_t(a))) return fail();
                    pos += uint32_t(a);
                    break;
                case 4:
                    if (a > uint64_t(n - pos)) return fail();
                    todo += a;
                    break;
                case 5:
                    if (a > uint64_t(n - pos) / 2) return fail();
                    todo += 2 * a;
                    break;
                case 6: {
                    if (a != 42) return fail();
                    uint32_t bo, bl;
                    read_bytes(bo, bl);
                    if (err) return;
                    if (bl < 1 || at(bo) != 0x00 || !cid_ok(bo + 1, bl - 1)) return fail();
                    break;
                }
                default:
                    return fail();
            }
        }
    }
    __device__ __forceinline__ void finish() {
        if (!err && pos != n) fail();
    }
};

// ---- typed value checks (what serde does when it decodes Vec<V> / bucket values) ----
enum ValueKind : int { VK_CID = 0, VK_RECEIPT = 1, VK_STAMPED_EVENT = 2, VK_ACTOR_STATE = 3, VK_VEC_U8 = 4, VK_ANY = 5 };

__device__ __forceinline__ void check_receipt(Rd& r) {  // fvm_shared Receipt
    r.expect_array(4);
    if (r.read_uint() > 0xffffffffULL) r.fail();  // exit_code: u32
    uint32_t o, l;
    r.read_bytes(o, l);
    (void)r.read_uint();
    if (r.at_null()) r.read_null();
    else r.read_link(o, l);
}

__device__ __forceinline__ void check_stamped_event(Rd& r) {
    r.expect_array(2);
    (void)r.read_uint();
    const uint64_t ne = r.read_array();
    for (uint64_t i = 0; i < ne && r.ok(); ++i) {
        uint32_t o, l;
        r.expect_array(4);
        (void)r.read_uint();
        r.read_text(o, l);
        (void)r.read_uint();
        r.read_bytes(o, l);
    }
}

__device__ __forceinline__ void check_address(Rd& r, uint32_t off, uint32_t n) {  // Address::from_bytes shape
    if (n < 1) return r.fail();
    const uint32_t proto = r.at(off);
    if (proto == 0 || proto == 4) {
        uint32_t pos = 1;
        bool term = false;
        for (int k = 0; k < 10; ++k) {
            if (pos >= n) return r.fail();
            if (!(r.at(off + pos++) & 0x80)) {
                term = true;
                break;
            }
        }
        if (!term) return r.fail();
        if (proto == 0) {
            if (pos != n) r.fail();
        } else if (n - pos > 54) {
            r.fail();
        }
    } else if (proto == 1 || proto == 2) {
        if (n != 21) r.fail();
    } else if (proto == 3) {
        if (n != 49) r.fail();
    } else {
        r.fail();
    }
}

__device__ __forceinline__ void check_actor_state(Rd& r) {
    uint32_t o, l;
    r.expect_array(5);
    r.read_link(o, l);
    r.read_link(o, l);
    (void)r.read_uint();
    r.read_bytes(o, l);  // TokenAmount: sign byte 0|1 + magnitude, ≤ 128 bytes
    if (r.ok() && (l > 128 || (l > 0 && r.at(o) > 1))) r.fail();
    if (r.at_null()) r.read_null();
    else {
        r.read_bytes(o, l);
        if (r.ok()) check_address(r, o, l);
    }
}

// A tag-42 link at `at` in one of the two spellings encoders write, from two fetches: → its encoded length, 0 for any
// other spelling (the caller takes read_link, which also decides what is an error).
//   standard   d8 2a | 58 27 | 00 | 01 71 a0 e4 02 20 | digest[32]          43 bytes (CIDv1 dag-cbor blake2b-256)
//   short      d8 2a | 4l / 58 l | 00 | 01 | codec < 80 | hash code < 80 | n ≤ 40 | digest[n],  l = 5 + n
//              — e.g. the builtin actors' code CIDs: raw (0x55), identity hash (0x00), the actor's name as "digest"
// Both pass every check of read_link / cid_ok (version 1, minimal one-byte varints, the digest fills the string).
#if !IPCFP_RD_RING
__device__ __forceinline__ uint32_t link_fast_len(Rd& r, uint32_t at) {
    const uint64_t w0 = r.peek64(at);
    if (w0 == 0xa071010027582ad8ull) return (r.peek64(at + 8u) & 0xffffffull) == 0x2002e4ull ? 43u : 0u;
    if ((w0 & 0xffffull) != 0x2ad8ull) return 0u;
    const uint32_t hb = uint32_t(w0 >> 16) & 0xffu;
    uint32_t l, h;  // string length, header bytes before the string
    uint64_t body;  // the string's first bytes: 00 01 codec code n
    if (hb >= 0x45u && hb <= 0x57u) {
        l = hb - 0x40u;
        h = 3u;
        body = w0 >> 24;
    } else if (hb == 0x58u) {
        l = uint32_t(w0 >> 24) & 0xffu;
        h = 4u;
        body = w0 >> 32;
    } else {
        return 0u;
    }
    const uint32_t n = h == 3u ? uint32_t(body >> 32) & 0xffu : uint32_t(r.at(at + 8u));
    if ((body & 0xffffull) != 0x0100ull || (body & 0x80800000ull) != 0ull || n > 40u || l != 5u + n) return 0u;
    return h + l;
}

// One HAMT bucket entry `[key bytes, ActorState]` in the spelling every encoder writes — settled from a handful of
// fetches instead of eight generic item headers and eight varints (a state-tree node holds ≈ 45 entries and a walk
// decodes every entry of every node it visits: src/proofs/common/decode.rs:29-39):
//   82 | 4k key[k ≤ 6] | 85 | link | link | sequence uint | balance 4l [sign ≤ 1, …] (l ≤ 23) | f6        (links: link_fast_len)
// true: exactly that; r.pos is past the entry, (ko, kl) is the key, vstart the ActorState item.  Every check
// check_actor_state makes holds for this form (the links are well-formed CIDv1s, the TokenAmount is short with a 0/1
// sign byte, delegated_address is None).  false: some other spelling — r is UNTOUCHED (pos, err) and the caller decodes
// the entry item by item, which also decides what is an error.
__device__ __forceinline__ bool actor_entry_fast(Rd& r, uint32_t& ko, uint32_t& kl, uint32_t& vstart) {
    if (r.err) return false;
    const uint32_t at = r.pos;
    if (at + 48u > r.n) return false;  // (far shorter than any such entry: spares the peeks near the end)
    const uint64_t w0 = r.peek64(at);
    const uint32_t k = (uint32_t(w0 >> 8) & 0xffu) - 0x40u;
    if ((uint32_t(w0) & 0xffu) != 0x82u || k > 6u) return false;
    const uint32_t p2 = at + 2u + k;
    if (r.at(p2) != 0x85u) return false;
    const uint32_t l1 = link_fast_len(r, p2 + 1u);
    if (!l1) return false;
    const uint32_t l2 = link_fast_len(r, p2 + 1u + l1);
    if (!l2) return false;
    const uint32_t p3 = p2 + 1u + l1 + l2;
    const uint64_t w3 = r.peek64(p3);
    const uint32_t b3 = uint32_t(w3) & 0xffu;
    if (b3 > 0x1bu) return false;                                // major 0, 1/2/3/5/9 bytes
    const uint32_t p4 = p3 + 1u + (b3 < 0x18u ? 0u : (1u << (b3 - 0x18u)));
    const uint64_t w4 = r.peek64(p4);
    const uint32_t l = (uint32_t(w4) & 0xffu) - 0x40u;           // TokenAmount bytes, immediate length
    if (l > 23u || (l > 0u && (uint32_t(w4 >> 8) & 0xffu) > 1u)) return false;
    const uint32_t p5 = p4 + 1u + l;
    if (p5 >= r.n || r.at(p5) != 0xf6u) return false;
    ko = at + 2u;
    kl = k;
    vstart = p2;
    r.pos = p5 + 1u;
    return true;
}
#endif

// One element of a serde Vec<u8> (a CBOR array of small unsigned integers) out of the 8 bytes `w` fetched at the reader's
// position, `used` bytes of which are consumed already: 00..17 is the value in one byte, 18 xx the value in two — what
// every encoder writes.  false: the element is spelled some other way (or may straddle the fetch): take the general path.
__device__ __forceinline__ bool vec_u8_step(uint64_t w, uint32_t& used, uint32_t& value) {
    if (used > 6u) return false;
    const uint32_t b = uint32_t(w >> (8u * used)) & 0xffu;
    if (b < 0x18u) {
        value = b;
        used += 1u;
        return true;
    }
    if (b == 0x18u) {
        value = uint32_t(w >> (8u * used + 8u)) & 0xffu;
        used += 2u;
        return true;
    }
    return false;
}

// Four elements of a serde Vec<u8> out of the eight bytes {hi, lo} at the reader's position (one or two bytes
// each, so eight bytes always hold four): the elements packed big-endian into `cur` (first element highest), the bytes they
// took added to `pos`.  `bad`: a byte that is neither 00..17 nor 18 where an element starts.  COUNT ≤ 4.
template <int COUNT>
__device__ __forceinline__ void vec_u8_take(uint32_t lo, uint32_t hi, uint32_t& cur, uint32_t& pos, uint32_t& bad) {
#pragma unroll
    for (int e = 0; e < COUNT; ++e) {
        const uint32_t b = lo & 0xffu;
        const bool two = b == 0x18u;
        bad |= uint32_t(b > 0x18u);
        const uint32_t x = two ? (lo >> 8) & 0xffu : b;
        const uint32_t sh = two ? 16u : 8u;
        lo = __builtin_amdgcn_alignbit(hi, lo, sh);
        hi >>= sh;
        cur = (cur << 8) | x;
        pos += sh >> 3;
    }
}

// A serde Vec<u8> in its usual spelling (`8n` | `98 nn`, then n elements of one or two bytes) at offset `at` of the reader's
// item, four elements per fetch: → the offset behind it, or 0 when it is spelled some other way (or is no such thing): the
// caller then reads it item by item.  The reader's position is left alone.
__device__ __forceinline__ uint32_t vec_u8_end(Rd& r, uint32_t at) {
    if (at >= r.n) return 0;
    const uint64_t v8 = r.peek64(at);
    const uint32_t hv = uint32_t(v8) & 0xffu;
    uint32_t n;
    if (hv >= 0x80u && hv < 0x98u) {
        n = hv - 0x80u;
        at += 1u;
    } else if (hv == 0x98u) {
        n = uint32_t(v8 >> 8) & 0xffu;
        at += 2u;
    } else {
        return 0;
    }
    uint32_t cur = 0, bad = 0, q = n >> 2;
    for (; q && at < r.n; --q) {
        const uint64_t w8 = r.peek64(at);
        vec_u8_take<4>(uint32_t(w8), uint32_t(w8 >> 32), cur, at, bad);
    }
    if (q) return 0;
    const uint32_t rest = n & 3u;
    if (rest) {
        if (at >= r.n) return 0;
        const uint64_t w8 = r.peek64(at);
        if (rest == 1) vec_u8_take<1>(uint32_t(w8), uint32_t(w8 >> 32), cur, at, bad);
        else if (rest == 2) vec_u8_take<2>(uint32_t(w8), uint32_t(w8 >> 32), cur, at, bad);
        else vec_u8_take<3>(uint32_t(w8), uint32_t(w8 >> 32), cur, at, bad);
    }
    return bad || at > r.n || !r.ok() ? 0u : at;
}

__device__ __forceinline__ void check_vec_u8(Rd& r) {  // serde Vec<u8> = array of u8
    const uint64_t n = r.read_array();
    uint64_t i = 0;
    // A storage value is up to 32 such elements and a storage node holds hundreds: one generic item header per element
    // (≈80 instructions) was 50 instructions per byte of witness (profiles/r03_experiments.md).  Eight bytes per fetch,
    // a handful of instructions per element; anything unusual falls through to read_uint.
    while (i < n && r.ok() && r.pos + 8u <= r.n) {
        const uint64_t w = r.peek64(r.pos);
        uint32_t used = 0, v;
        while (i < n && vec_u8_step(w, used, v)) ++i;
        r.pos += used;
        if (i < n && used <= 6u) {  // the element at the reader's position is not in the short form
            if (r.read_uint() > 255) r.fail();
            ++i;
        }
    }
    for (; i < n && r.ok(); ++i)
        if (r.read_uint() > 255) r.fail();
}

__device__ __forceinline__ void check_value(Rd& r, int kind) {
    uint32_t o, l;
    switch (kind) {
        case VK_CID: r.read_link(o, l); break;
        case VK_RECEIPT: check_receipt(r); break;
        case VK_STAMPED_EVENT: check_stamped_event(r); break;
        case VK_ACTOR_STATE: check_actor_state(r); break;
        case VK_VEC_U8: check_vec_u8(r); break;
        default: r.skip(); break;
    }
}

}  // namespace ipcfp
