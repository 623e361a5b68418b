//! bindings/rust/ffi.rs — the Rust side of the drop-in boundary.
//!
//! NOT compiled in this repository (the build image has no Rust toolchain): this is the module a maintainer of
//! consensus-shipyard/ipc-filecoin-proofs adds as `src/gpu/mod.rs` (with `ffi_sys.rs` beside it as
//! `src/gpu/ffi_sys.rs`) and `println!("cargo:rustc-link-lib=dylib=ipcfp")` in build.rs.
//!
//! * `ffi_sys.rs` — GENERATED from `include/ipcfp.h` (tools/gen_rust_ffi.py): the raw declaration of every symbol.
//! * this file — the `#[repr(C)]` mirrors of the header's structs and the safe wrappers, which keep the reference's
//!   names, argument meaning and error behaviour:
//!     `load_witness_store`            src/proofs/events/verifier.rs:79-89
//!     `verify_event_proof`            src/proofs/events/verifier.rs:51-74 (built-in filter) — `verify_event_proof_with`
//!                                     for an arbitrary `check_event` closure (:247-251)
//!     `verify_storage_proof`          src/proofs/storage/verifier.rs:24-63
//!     `verify_proof_bundle[_json]`    src/proofs/verifier.rs:12-62
//!     `generate_proof_bundle`         src/proofs/generator.rs:25-95
//!     `impl Blockstore for Witness`   src/proofs/common/blockstore.rs:26-39 (get / put_keyed / has)
//!   A string with an interior NUL cannot cross a C ABI; the reference would fail to parse it (CID fields) or find it
//!   unequal (compared fields), so the wrappers replace NUL by 0x01 — a byte that is in no multibase / hex alphabet
//!   either — instead of panicking (`c_string`).
#![allow(non_camel_case_types)]
use std::cell::RefCell;
use std::ffi::{c_char, c_int, CStr, CString};

use anyhow::{anyhow, Result};
use cid::Cid;
use fvm_ipld_blockstore::Blockstore;

pub mod ffi_sys;
pub use ffi_sys::*;

use crate::proofs::common::bundle::{ProofBlock, UnifiedProofBundle};
use crate::proofs::events::bundle::{EventData, EventProof, EventProofBundle};
use crate::proofs::generator::{EventProofSpec, StorageProofSpec};
use crate::proofs::storage::bundle::StorageProof;
use crate::proofs::trust::TrustPolicy;

// ---- opaque handles -----------------------------------------------------------------------------------------------
#[repr(C)] pub struct ipcfp_ctx_t { _p: [u8; 0] }
#[repr(C)] pub struct ipcfp_witness_t { _p: [u8; 0] }
#[repr(C)] pub struct ipcfp_bundle_t { _p: [u8; 0] }
#[repr(C)] pub struct ipcfp_packed_events_t { _p: [u8; 0] }
#[repr(C)] pub struct ipcfp_comm_t { _p: [u8; 0] }

// ---- PODs of include/ipcfp.h --------------------------------------------------------------------------------------
pub const IPCFP_CID_SLOT: usize = 40;
pub const IPCFP_MAX_PARENTS: usize = 32;
/// == `IPCFP_ABI_VERSION` of include/ipcfp.h: the struct layouts below are this version's (2: IPCFP_MAX_PARENTS 16 -> 32;
/// 3: `ipcfp_tipset_ref_t.more_parents`, long CIDs cross folded)
pub const IPCFP_ABI_VERSION: c_int = 3;
pub const IPCFP_SCAN_PHASE_RECEIPTS: u32 = 1;
pub const IPCFP_SCAN_PHASE_EVENTS: u32 = 2;
pub const IPCFP_ST_TRUE: u8 = 1;
pub const IPCFP_ST_FALSE_FILTER: u8 = 17;
/// `ipcfp_check_event_fn` of the header: the host predicate of ipcfp_verify_event_proofs_with
pub type ipcfp_check_event_fn = Option<unsafe extern "C" fn(user: *mut std::ffi::c_void, proof_index: u64, stamped_event: *const u8, len: u64) -> c_int>;

#[repr(C)]
pub struct ipcfp_event_proof_t {
    pub parent_epoch: i64,
    pub child_epoch: i64,
    pub parent_tipset_cids: *const *const c_char,
    pub n_parent_tipset_cids: u32,
    pub child_block_cid: *const c_char,
    pub message_cid: *const c_char,
    pub exec_index: u64,
    pub event_index: u64,
    pub emitter: u64,
    pub topics: *const *const c_char,
    pub n_topics: u32,
    pub data: *const c_char,
}

#[repr(C)]
pub struct ipcfp_storage_proof_t {
    pub child_epoch: i64,
    pub child_block_cid: *const c_char,
    pub parent_state_root: *const c_char,
    pub actor_id: u64,
    pub actor_state_cid: *const c_char,
    pub storage_root: *const c_char,
    pub slot: *const c_char,
    pub value: *const c_char,
}

#[repr(C)] #[derive(Clone, Copy)] pub struct ipcfp_event_filter_t { pub topic0: [u8; 32], pub topic1: [u8; 32] }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct ipcfp_value_loc_t { pub block: u32, pub off: u32, pub len: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct ipcfp_event_match_t { pub exec_index: u64, pub event_index: u64, pub emitter: u64,
                                                                   pub event: ipcfp_value_loc_t, pub reserved: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct ipcfp_generated_storage_t { pub parent_state_root: [u8; 40], pub actor_state_cid: [u8; 40],
                                                                         pub storage_root: [u8; 40], pub value: [u8; 32],
                                                                         pub status: u32, pub reserved: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct ipcfp_trust_policy_t { pub kind: c_int, pub ec_chain_empty: c_int, pub min_epoch: i64, pub max_epoch: i64 }
/// `more_parents`: the slots of parents[IPCFP_MAX_PARENTS ..] of a tipset key wider than the inline form (host memory, valid for the call)
#[repr(C)] pub struct ipcfp_tipset_ref_t { pub flags: u32, pub n_parents: u32, pub child: [u8; 40], pub parents: [[u8; 40]; IPCFP_MAX_PARENTS],
                                           pub more_parents: *const u8 }
#[repr(C)] pub struct ipcfp_event_claim_t { pub parent_epoch: i64, pub child_epoch: i64, pub exec_index: u64, pub event_index: u64,
                                            pub emitter: u64, pub message_cid: [u8; 40], pub tipset: u32, pub flags: u32,
                                            pub n_topics: u32, pub topics_off: u32, pub data_off: u32, pub data_len: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct ipcfp_event_claim_group_t { pub parent_epoch: i64, pub child_epoch: i64, pub tipset: u32, pub reserved: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct ipcfp_event_claim_compact_t { pub emitter: u64, pub exec_index: u32, pub event_index: u32,
                                                                          pub message_digest: [u8; 32], pub data_len: u16, pub n_topics: u8,
                                                                          pub topic_flags: u8, pub flags: u8, pub group: u8, pub reserved: u16 }
#[repr(C)] pub struct ipcfp_storage_claim_t { pub child_epoch: i64, pub actor_id: u64, pub child: [u8; 40], pub state_root: [u8; 40],
                                              pub actor_state: [u8; 40], pub storage_root: [u8; 40], pub slot: [u8; 32],
                                              pub value: [u8; 32], pub flags: u32, pub reserved: u32 }
#[repr(C)] pub struct ipcfp_storage_proof_spec_t { pub actor_id: u64, pub slot: [u8; 32] }
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct ipcfp_shard_pull_stats_t { pub rounds: u32, pub blocks: u32, pub table_bytes: u64, pub block_bytes: u64, pub payload_bytes: u64,
                                      pub tables_ms: f64, pub pull_ms: f64, pub create_ms: f64 }
#[repr(C)] pub struct ipcfp_event_proof_spec_t { pub event_signature: *const c_char, pub topic_1: *const c_char,
                                                 pub actor_id_filter: u64, pub has_actor_id_filter: u8 }

// ---- helpers --------------------------------------------------------------------------------------------------------
/// A Rust string as a C string.  An interior NUL (legal in a JSON claim: "\u0000") becomes 0x01: like NUL it is in no
/// multibase / hex alphabet and equals no character of a canonical form, so every parse and compare keeps its outcome.
fn c_string(s: &str) -> CString {
    let bytes: Vec<u8> = s.bytes().map(|b| if b == 0 { 1 } else { b }).collect();
    CString::new(bytes).expect("no interior NUL left")
}

/// A `Cid` as its 40-byte ABI slot: zero padded, or — longer than the slot (a 64-byte digest) — FOLDED to
/// `ff | len | blake2b-256(bytes)` by `ipcfp_cid_to_slot` (include/ipcfp.h "CIDs"); the device folds the long links it
/// reads out of blocks the same way, so such a CID is found and compared like any other.
fn cid_slot(c: &Cid) -> Result<[u8; 40]> {
    let b = c.to_bytes();
    let mut slot = [0u8; 40];
    match unsafe { ipcfp_cid_to_slot(b.as_ptr(), b.len() as u32, slot.as_mut_ptr()) } {
        n if n > 0 => Ok(slot),
        rc => Err(anyhow!("ipcfp_cid_to_slot({c}): {rc}")),
    }
}

fn cid_from_slot(slot: &[u8; 40]) -> Result<Cid> {
    // (a CID longer than the slot comes back as its fold: its bytes stand in the block that holds it — include/ipcfp.h "CIDs")
    if slot[0] == 0xff { return Err(anyhow!("a CID of {} bytes came back folded; read it from its block", slot[1])); }
    // a binary CID is self-delimiting: Cid::read_bytes stops after the multihash
    Ok(Cid::read_bytes(&slot[..])?)
}

/// `status >= 64` is `Err`; the reference aborts at the first one in proof order
/// (src/proofs/events/verifier.rs:62-71, src/proofs/verifier.rs:19-28).
fn statuses_to_result(st: &[u8]) -> Result<Vec<bool>> {
    if let Some((i, s)) = st.iter().enumerate().find(|(_, s)| **s >= 64) {
        return Err(anyhow!("proof {i}: verification error (ipcfp status {s})"));
    }
    Ok(st.iter().map(|s| *s == IPCFP_ST_TRUE).collect())
}

/// `TrustPolicy` (src/proofs/trust/mod.rs:8-16,53-78) as the POD the engine evaluates: both variants are pure
/// functions of the epoch (AcceptAll; F3Certificate = the EC chain's epoch range, src/cert.rs:52-64).
pub fn trust_pod(p: &TrustPolicy) -> ipcfp_trust_policy_t {
    match p {
        TrustPolicy::AcceptAll => ipcfp_trust_policy_t { kind: 0, ec_chain_empty: 0, min_epoch: 0, max_epoch: 0 },
        TrustPolicy::F3Certificate(cert) => match (cert.ec_chain.first(), cert.ec_chain.last()) {
            (Some(a), Some(b)) => ipcfp_trust_policy_t { kind: 1, ec_chain_empty: 0, min_epoch: a.epoch, max_epoch: b.epoch },
            _ => ipcfp_trust_policy_t { kind: 1, ec_chain_empty: 1, min_epoch: 0, max_epoch: 0 },
        },
    }
}

// ---- the bundle in transport form; a rank's shard -----------------------------------------------------------------------
/// `Vec<ProofBlock>` (src/proofs/common/bundle.rs:10-15) as the tables of `ipcfp_witness_create_packed` /
/// `ipcfp_witness_create_shard_pull`: blocks back to back, lengths, 32-byte digests + the chain's CID prefix, escapes for
/// every other CID form.  `register = true` maps `bytes` for device reads (`ipcfp_host_register`; undone on drop): what a
/// self-planned shard needs — an ingest buffer is registered once, when it is made, not per call.
/// A byte buffer that OWNS its pages (page-aligned, whole pages): what `ipcfp_host_register` takes (include/ipcfp.h — a
/// slice of the heap shares its first and last page with other allocations, and registration is by page).
pub struct PageBuf { ptr: *mut u8, len: usize, cap: usize }
impl PageBuf {
    const PAGE: usize = 4096;
    pub fn from_slice(src: &[u8]) -> Self {
        let cap = std::cmp::max((src.len() + Self::PAGE - 1) / Self::PAGE * Self::PAGE, Self::PAGE);
        let layout = std::alloc::Layout::from_size_align(cap, Self::PAGE).expect("page layout");
        let ptr = unsafe { std::alloc::alloc_zeroed(layout) };
        if ptr.is_null() { std::alloc::handle_alloc_error(layout); }
        unsafe { std::ptr::copy_nonoverlapping(src.as_ptr(), ptr, src.len()); }
        Self { ptr, len: src.len(), cap }
    }
    pub fn as_ptr(&self) -> *const u8 { self.ptr }
    pub fn as_mut_ptr(&mut self) -> *mut u8 { self.ptr }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn as_slice(&self) -> &[u8] { unsafe { std::slice::from_raw_parts(self.ptr, self.len) } }
}
impl Drop for PageBuf {
    fn drop(&mut self) { unsafe { std::alloc::dealloc(self.ptr, std::alloc::Layout::from_size_align_unchecked(self.cap, Self::PAGE)); } }
}

pub struct PackedBundle { pub bytes: PageBuf, pub len: Vec<u32>, pub digests: Vec<u8>, pub esc_index: Vec<u32>, pub esc_cids: Vec<u8>,
                          registered: bool }
impl PackedBundle {
    pub const STD: [u8; 6] = [0x01, 0x71, 0xa0, 0xe4, 0x02, 0x20];  // CIDv1, dag-cbor, blake2b-256, 32-byte digest
    pub fn new(blocks: &[ProofBlock], register: bool) -> Result<Self> {
        let (mut bytes, mut len, mut digests) = (Vec::new(), Vec::new(), Vec::new());
        let (mut esc_index, mut esc_cids) = (Vec::<u32>::new(), Vec::<u8>::new());
        for (i, b) in blocks.iter().enumerate() {
            len.push(u32::try_from(b.data.len())?);
            bytes.extend_from_slice(&b.data);
            let slot = cid_slot(&b.cid)?;
            if slot[..6] == Self::STD && slot[38..] == [0, 0] {
                digests.extend_from_slice(&slot[6..38]);
            } else {
                digests.extend_from_slice(&[0u8; 32]);
                esc_index.push(i as u32);
                esc_cids.extend_from_slice(&slot);
            }
        }
        let mut t = Self { bytes: PageBuf::from_slice(&bytes), len, digests, esc_index, esc_cids, registered: false };
        if register && !t.bytes.is_empty() {
            match unsafe { ipcfp_host_register(t.bytes.as_mut_ptr() as *mut std::ffi::c_void, t.bytes.len() as u64) } {
                0 => t.registered = true,
                rc => return Err(anyhow!("ipcfp_host_register: {rc}")),
            }
        }
        Ok(t)
    }
}
impl Drop for PackedBundle {
    fn drop(&mut self) { if self.registered { unsafe { ipcfp_host_unregister(self.bytes.as_mut_ptr() as *mut std::ffi::c_void); } } }
}

/// Rank `shard`'s part of one tipset: the witness of the receipts `receipts` (of `n_receipts`), and what the pull moved.
pub struct WitnessShard<'e> { pub witness: Witness<'e>, pub receipts: std::ops::Range<u64>, pub n_receipts: u64, pub n_shards: u32,
                              pub shard: u32, pub stats: ipcfp_shard_pull_stats_t }

/// The event proofs of a bundle lowered ONCE to the binary claims every rank slices (`ipcfp_pack_event_proofs`): the
/// reference's strings parsed (CIDs, hex) exactly as `verify_event_proof` would (src/proofs/events/verifier.rs:93-181).
pub struct PackedEvents { p: *mut ipcfp_packed_events_t }
impl PackedEvents {
    pub fn new(proofs: &[EventProof]) -> Result<Self> {
        let keep: Vec<_> = proofs.iter().map(CEventProof::new).collect();
        let raw: Vec<ipcfp_event_proof_t> = keep.iter().map(|k| k.raw()).collect();
        let mut p = std::ptr::null_mut();
        match unsafe { ipcfp_pack_event_proofs(raw.as_ptr(), raw.len() as u64, &mut p) } { 0 => Ok(Self { p }), rc => Err(anyhow!("ipcfp_pack_event_proofs: {rc}")) }
    }
    pub fn len(&self) -> usize { let mut n = 0u64; unsafe { ipcfp_packed_events_claims(self.p, &mut n) }; n as usize }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
}
impl Drop for PackedEvents { fn drop(&mut self) { unsafe { ipcfp_packed_events_destroy(self.p) } } }

impl WitnessShard<'_> {
    /// `verify_event_proof` (src/proofs/events/verifier.rs:51-74) for THIS rank's share of the batch: the proofs whose
    /// `exec_index` lies in `self.receipts` (the last rank also takes everything beyond the last receipt, so that every
    /// proof has one owner), found by two binary searches in `events` — which must be in `exec_index` order, the order
    /// `generate_event_proof` emits (src/proofs/events/generator.rs:242-301); the device checks it and the call is an
    /// `Err` otherwise.  Returns the index of the first proof verified and the verdicts from there on.  A multi-GPU host
    /// gathers every rank's `(first, statuses)` (one all-gather of status bytes) and only THEN applies the reference's
    /// "first Err aborts" rule over the whole batch: `merge_event_statuses`.
    pub fn verify_event_proof_range(&self, events: &PackedEvents, trust: &ipcfp_trust_policy_t,
                                    filter: Option<&ipcfp_event_filter_t>) -> Result<(usize, Vec<u8>)> {
        let (mut nt, mut n, mut bl) = (0u32, 0u64, 0u64);
        let (ts, cl, blob) = unsafe { (ipcfp_packed_events_tipsets(events.p, &mut nt), ipcfp_packed_events_claims(events.p, &mut n),
                                       ipcfp_packed_events_blob(events.p, &mut bl)) };
        let mut st = vec![0u8; (n as usize).max(1)];
        let (mut first, mut count) = (0u64, 0u64);
        let w = &self.witness;
        let rc = unsafe { ipcfp_verify_event_claims_range(w.eng.ctx, w.raw(), ts, nt, cl, n, blob, bl, self.receipts.start, self.receipts.end,
                                                          (self.shard + 1 == self.n_shards) as c_int, trust,
                                                          filter.map_or(std::ptr::null(), |f| f as *const _), &mut first, &mut count,
                                                          st.as_mut_ptr()) };
        if rc != 0 { return Err(w.eng.err("ipcfp_verify_event_claims_range", rc)); }
        st.truncate(count as usize);
        Ok((first as usize, st))
    }
}

/// The verdicts of a whole batch from every rank's `(first, statuses)`: the reference's `Result<Vec<bool>>` — the Err of
/// the LOWEST proof index aborts (src/proofs/events/verifier.rs:62-71), whichever rank met it.
pub fn merge_event_statuses(n_proofs: usize, per_rank: &[(usize, Vec<u8>)]) -> Result<Vec<bool>> {
    let mut all = vec![0u8; n_proofs];
    let mut seen = vec![false; n_proofs];
    for (first, st) in per_rank {
        for (k, s) in st.iter().enumerate() {
            let i = first + k;
            if i >= n_proofs || seen[i] { return Err(anyhow!("merge_event_statuses: proof {i} has no single owner")); }
            all[i] = *s;
            seen[i] = true;
        }
    }
    if let Some(i) = seen.iter().position(|s| !*s) { return Err(anyhow!("merge_event_statuses: proof {i} was verified by no rank")); }
    statuses_to_result(&all)
}

// ---- engine / witness -------------------------------------------------------------------------------------------------
pub struct Engine { ctx: *mut ipcfp_ctx_t }
/// The HBM-resident witness store.  `&self` methods with interior state, like the trait it implements
/// (`fvm_ipld_blockstore::Blockstore` takes `&self`); not `Sync` — one thread at a time, as the C ABI requires.
pub struct Witness<'e> { eng: &'e Engine, w: RefCell<*mut ipcfp_witness_t> }

impl Engine {
    pub fn new(device: i32) -> Result<Self> {
        // a library built from another header strides arrays of these structs wrongly: refuse it before the first call
        let abi = unsafe { ipcfp_abi_version() };
        if abi != IPCFP_ABI_VERSION { return Err(anyhow!("libipcfp speaks ABI {abi}, this binding ABI {IPCFP_ABI_VERSION}")); }
        let mut ctx = std::ptr::null_mut();
        match unsafe { ipcfp_ctx_create(device, &mut ctx) } { 0 => Ok(Self { ctx }), rc => Err(anyhow!("ipcfp_ctx_create: {rc}")) }
    }
    fn err(&self, what: &str, rc: c_int) -> anyhow::Error {
        let msg = unsafe { CStr::from_ptr(ipcfp_last_error(self.ctx)) }.to_string_lossy().into_owned();
        anyhow!("{what}: {rc} {msg}")
    }

    /// replaces `load_witness_store(blocks)` (src/proofs/events/verifier.rs:79-89, src/proofs/storage/verifier.rs:68-78).
    /// The blocks are laid back to back as they come, so the tables cross PCIe in transport form
    /// (`ipcfp_witness_create_packed`): lengths instead of offsets, 32-byte digests + the chain's CID prefix instead of
    /// 40-byte slots, and the few CIDs of another form as escapes.
    pub fn load_witness_store(&self, blocks: &[ProofBlock]) -> Result<Witness<'_>> {
        let t = PackedBundle::new(blocks, false)?;
        let mut w = std::ptr::null_mut();
        let rc = unsafe { ipcfp_witness_create_packed(self.ctx, t.bytes.as_ptr(), t.bytes.len() as u64, t.len.as_ptr(), t.digests.as_ptr(),
                                                      t.len.len() as u64, PackedBundle::STD.as_ptr(), PackedBundle::STD.len() as u32,
                                                      t.esc_index.as_ptr(), t.esc_cids.as_ptr(), t.esc_index.len() as u64, &mut w) };
        if rc != 0 { return Err(self.err("ipcfp_witness_create_packed", rc)); }
        Ok(Witness { eng: self, w: RefCell::new(w) })
    }

    /// MULTI-GPU, rank `shard` of `n_shards` (one process per GPU; INTEGRATION.md "Multi-GPU"): the receipt-range shard
    /// of ONE tipset, built by the device straight out of the bundle in THIS host's memory — no rank holds the whole
    /// witness, nobody plans for anybody (`ipcfp_witness_create_shard_pull`).  Cuts the sequential loops of
    /// src/proofs/verifier.rs:19-28,49-54 and src/proofs/events/verifier.rs:62-71 by receipt index.
    /// `Ok(None)`: the child header or the receipts root is not in the bundle — there is no range to cut by, verify on
    /// the whole bundle (`load_witness_store`).
    pub fn shard_pull<'e>(&'e self, bundle: &PackedBundle, parent_cids: &[Cid], child_cid: &Cid, n_shards: u32, shard: u32)
                          -> Result<Option<WitnessShard<'e>>> {
        if !bundle.registered { return Err(anyhow!("shard_pull: the bundle's bytes must be device-readable (PackedBundle::new(blocks, true))")); }
        let mut pc = Vec::new();
        for c in parent_cids { pc.extend_from_slice(&cid_slot(c)?); }
        let child = cid_slot(child_cid)?;
        let (mut st, mut lo, mut hi, mut n_receipts) = (0u8, 0u64, 0u64, 0u64);
        let mut stats = ipcfp_shard_pull_stats_t::default();
        let mut w = std::ptr::null_mut();
        let rc = unsafe { ipcfp_witness_create_shard_pull(self.ctx, bundle.bytes.as_ptr(), bundle.bytes.len() as u64, bundle.len.as_ptr(),
                                                          bundle.digests.as_ptr(), bundle.len.len() as u64, PackedBundle::STD.as_ptr(),
                                                          PackedBundle::STD.len() as u32, bundle.esc_index.as_ptr(), bundle.esc_cids.as_ptr(),
                                                          bundle.esc_index.len() as u64, pc.as_ptr(), parent_cids.len() as u32, child.as_ptr(),
                                                          n_shards, shard, &mut st, &mut lo, &mut hi, &mut n_receipts, &mut stats, &mut w) };
        if rc != 0 { return Err(self.err("ipcfp_witness_create_shard_pull", rc)); }
        if st != IPCFP_ST_TRUE || w.is_null() { return Ok(None); }
        Ok(Some(WitnessShard { witness: Witness { eng: self, w: RefCell::new(w) }, receipts: lo..hi, n_receipts, n_shards, shard, stats }))
    }

    /// `create_event_filter(event_sig, subnet_id)` (src/proofs/events/verifier.rs:28-39) as the POD the device applies
    pub fn create_event_filter(&self, event_sig: &str, subnet_id: &str) -> Result<ipcfp_event_filter_t> {
        let (s, t) = (c_string(event_sig), c_string(subnet_id));
        let mut f = ipcfp_event_filter_t { topic0: [0; 32], topic1: [0; 32] };
        match unsafe { ipcfp_create_event_filter(self.ctx, s.as_ptr(), t.as_ptr(), &mut f) } { 0 => Ok(f), rc => Err(self.err("ipcfp_create_event_filter", rc)) }
    }

    /// drop-in for `verify_proof_bundle(&bundle, policy, filter)` (src/proofs/verifier.rs:12-62): one witness per
    /// bundle (the reference rebuilds it per storage proof), storage proofs first, the first `Err` aborts.
    pub fn verify_proof_bundle(&self, bundle: &UnifiedProofBundle, policy: &TrustPolicy,
                               filter: Option<&ipcfp_event_filter_t>) -> Result<(Vec<bool>, Vec<bool>)> {
        let witness = self.load_witness_store(&bundle.blocks)?;
        let trust = trust_pod(policy);
        let storage = witness.verify_storage_proof(&bundle.storage_proofs, &trust)?;
        let events = witness.verify_event_proof(&EventProofBundle { proofs: bundle.event_proofs.clone(), blocks: vec![] },
                                                &trust, filter)?;
        Ok((storage, events))
    }

    /// `serde_json::from_str::<UnifiedProofBundle>(text)` + `verify_proof_bundle` without materialising the blocks on
    /// the host: the base64 of every `ProofBlock.data` is decoded on the device straight into the witness arena.
    pub fn verify_proof_bundle_json(&self, text: &str, policy: &TrustPolicy,
                                    filter: Option<&ipcfp_event_filter_t>) -> Result<(Vec<bool>, Vec<bool>)> {
        let trust = trust_pod(policy);
        let mut b = std::ptr::null_mut();
        let rc = unsafe { ipcfp_bundle_parse_json(self.ctx, text.as_ptr() as *const c_char, text.len() as u64, 0, &mut b) };
        if rc != 0 { return Err(self.err("bundle JSON", rc)); }
        let (ns, ne) = unsafe { (ipcfp_bundle_storage_count(b) as usize, ipcfp_bundle_event_count(b) as usize) };
        let (mut ss, mut es) = (vec![0u8; ns.max(1)], vec![0u8; ne.max(1)]);
        let rc = unsafe { ipcfp_verify_proof_bundle(self.ctx, b, &trust, filter.map_or(std::ptr::null(), |f| f as *const _),
                                                    ss.as_mut_ptr(), es.as_mut_ptr()) };
        unsafe { ipcfp_bundle_destroy(b) };
        if rc != 0 { return Err(self.err("ipcfp_verify_proof_bundle", rc)); }
        Ok((statuses_to_result(&ss[..ns])?, statuses_to_result(&es[..ne])?))
    }
}
impl Drop for Engine { fn drop(&mut self) { unsafe { ipcfp_ctx_destroy(self.ctx) } } }
impl Drop for Witness<'_> { fn drop(&mut self) { unsafe { ipcfp_witness_destroy(*self.w.borrow()) } } }

/// The scan status of ONE tipset from its receipt-range shards' `(status, phase)` pairs in range order (a multi-GPU host:
/// INTEGRATION.md "Multi-GPU").  The unsharded scan enumerates every receipt before it opens an events AMT, so the first
/// shard with an Err of the receipts phase decides, and only without one the first shard with any Err.
pub fn merge_scan_status(per_shard: &[(u8, u32)]) -> u8 {
    per_shard.iter().find(|(st, ph)| *st != 1 && *ph == IPCFP_SCAN_PHASE_RECEIPTS).or_else(|| per_shard.iter().find(|(st, _)| *st != 1)).map(|(st, _)| *st).unwrap_or(1)
}

impl Witness<'_> {
    fn raw(&self) -> *mut ipcfp_witness_t { *self.w.borrow() }

    /// where the Err of the last scan on this witness arose (0: it returned TRUE; IPCFP_SCAN_PHASE_RECEIPTS / _EVENTS)
    pub fn last_scan_phase(&self) -> u32 { unsafe { ipcfp_witness_last_scan_phase(self.raw()) as u32 } }

    /// K1: Blake2b-256 of every block against its CID — the check `MemoryBlockstore` never makes
    pub fn verify_cids(&self) -> Result<u64> {
        let mut bad = 0u64;
        match unsafe { ipcfp_witness_verify_cids(self.eng.ctx, self.raw(), std::ptr::null_mut(), &mut bad) } { 0 => Ok(bad), rc => Err(self.eng.err("ipcfp_witness_verify_cids", rc)) }
    }

    /// drop-in for `verify_event_proof` with the built-in `create_event_filter` closure (or none)
    /// (src/proofs/events/verifier.rs:51-74).
    pub fn verify_event_proof(&self, bundle: &EventProofBundle, trust: &ipcfp_trust_policy_t,
                              filter: Option<&ipcfp_event_filter_t>) -> Result<Vec<bool>> {
        let keep: Vec<_> = bundle.proofs.iter().map(CEventProof::new).collect();  // owns the C strings
        let raw: Vec<ipcfp_event_proof_t> = keep.iter().map(|k| k.raw()).collect();
        let mut st = vec![0u8; raw.len()];
        let rc = unsafe { ipcfp_verify_event_proofs(self.eng.ctx, self.raw(), raw.as_ptr(), raw.len() as u64, trust,
                                                    filter.map_or(std::ptr::null(), |f| f as *const _), st.as_mut_ptr()) };
        if rc != 0 { return Err(self.eng.err("ipcfp_verify_event_proofs", rc)); }
        statuses_to_result(&st)
    }

    /// `verify_event_proof(.., Some(&check_event))` for an ARBITRARY closure (src/proofs/events/verifier.rs:51-56,
    /// applied at :247-251): the device verifies everything up to and including `verify_event_data_matches`, reports
    /// where each proof's `StampedEvent` lies, and the closure runs here over the decoded events of the proofs that
    /// are still true — `Ok(false)` where it declines, exactly where the reference calls it.
    pub fn verify_event_proof_with(&self, bundle: &EventProofBundle, trust: &ipcfp_trust_policy_t,
                                   check_event: &dyn Fn(&fvm_shared::event::ActorEvent) -> bool) -> Result<Vec<bool>> {
        // the trampoline libipcfp.so calls for every proof that is still true (ipcfp_verify_event_proofs_with does the
        // "predicate false => FALSE_FILTER" fold behind the ABI): decode the located StampedEvent, hand its ActorEvent on
        struct Env<'a> { f: &'a dyn Fn(&fvm_shared::event::ActorEvent) -> bool }
        unsafe extern "C" fn tramp(user: *mut std::ffi::c_void, _proof: u64, ev: *const u8, len: u64) -> c_int {
            let env = &*(user as *const Env);
            let raw = std::slice::from_raw_parts(ev, len as usize);
            match fvm_ipld_encoding::from_slice::<fvm_shared::event::StampedEvent>(raw) {  // validated on the device already
                Ok(stamped) => (env.f)(&stamped.event) as c_int,
                Err(_) => 0,
            }
        }
        let keep: Vec<_> = bundle.proofs.iter().map(CEventProof::new).collect();
        let raw: Vec<ipcfp_event_proof_t> = keep.iter().map(|k| k.raw()).collect();
        let mut st = vec![0u8; raw.len()];
        let env = Env { f: check_event };
        let rc = unsafe { ipcfp_verify_event_proofs_with(self.eng.ctx, self.raw(), raw.as_ptr(), raw.len() as u64, trust, std::ptr::null(),
                                                         Some(tramp), &env as *const Env as *mut std::ffi::c_void, st.as_mut_ptr()) };
        if rc != 0 { return Err(self.eng.err("ipcfp_verify_event_proofs_with", rc)); }
        statuses_to_result(&st)
    }

    /// drop-in for `verify_storage_proof` over all storage proofs of a bundle (src/proofs/storage/verifier.rs:24-63;
    /// the loop of src/proofs/verifier.rs:19-28 — the witness is NOT rebuilt per proof).
    pub fn verify_storage_proof(&self, proofs: &[StorageProof], trust: &ipcfp_trust_policy_t) -> Result<Vec<bool>> {
        let keep: Vec<_> = proofs.iter().map(CStorageProof::new).collect();
        let raw: Vec<ipcfp_storage_proof_t> = keep.iter().map(|k| k.raw()).collect();
        let mut st = vec![0u8; raw.len()];
        let rc = unsafe { ipcfp_verify_storage_proofs(self.eng.ctx, self.raw(), raw.as_ptr(), raw.len() as u64, trust, st.as_mut_ptr()) };
        if rc != 0 { return Err(self.eng.err("ipcfp_verify_storage_proofs", rc)); }
        statuses_to_result(&st)
    }

    /// `generate_proof_bundle` (src/proofs/generator.rs:25-95) with this witness in the role of the RPC block store:
    /// every block the generators may load must be resident.  Returns the bundle with `blocks` in
    /// `BTreeSet<(Cid, Vec<u8>)>` order.
    pub fn generate_proof_bundle(&self, parent_cids: &[Cid], parent_epoch: i64, child_cid: &Cid, child_epoch: i64,
                                 storage_specs: &[StorageProofSpec], event_specs: &[EventProofSpec]) -> Result<UnifiedProofBundle> {
        let mut pc = Vec::new();
        for c in parent_cids { pc.extend_from_slice(&cid_slot(c)?); }
        let child = cid_slot(child_cid)?;
        let ss: Vec<ipcfp_storage_proof_spec_t> = storage_specs.iter().map(|s| ipcfp_storage_proof_spec_t { actor_id: s.actor_id, slot: s.slot.0 }).collect();
        let keep: Vec<(CString, CString)> = event_specs.iter().map(|e| (c_string(&e.event_signature), c_string(&e.topic_1))).collect();
        let es: Vec<ipcfp_event_proof_spec_t> = event_specs.iter().zip(&keep).map(|(e, k)| ipcfp_event_proof_spec_t {
            event_signature: k.0.as_ptr(), topic_1: k.1.as_ptr(), actor_id_filter: e.actor_id_filter.unwrap_or(0),
            has_actor_id_filter: e.actor_id_filter.is_some() as u8 }).collect();
        let n_blocks_max = unsafe { ipcfp_witness_block_count(self.raw()) } as usize;
        let cap_p = 1usize << 16;
        let mut sout = vec![unsafe { std::mem::zeroed::<ipcfp_generated_storage_t>() }; ss.len().max(1)];
        let mut est = vec![0u8; es.len().max(1)];
        let mut matches = vec![unsafe { std::mem::zeroed::<ipcfp_event_match_t>() }; cap_p];
        let (mut msg, mut spec_of) = (vec![0u8; cap_p * 40], vec![0u32; cap_p]);
        let (mut ids, mut wcids) = (vec![0u32; n_blocks_max.max(1)], vec![0u8; n_blocks_max.max(1) * 40]);
        let (mut n_p, mut n_b, mut first_err) = (0u64, 0u64, 0u64);
        let rc = unsafe { ipcfp_generate_proof_bundle(self.eng.ctx, self.raw(), pc.as_ptr(), parent_cids.len() as u32, child.as_ptr(),
                                                      ss.as_ptr(), ss.len() as u64, es.as_ptr(), es.len() as u64, sout.as_mut_ptr(),
                                                      est.as_mut_ptr(), matches.as_mut_ptr(), msg.as_mut_ptr(), spec_of.as_mut_ptr(),
                                                      cap_p as u64, &mut n_p, ids.as_mut_ptr(), wcids.as_mut_ptr(), ids.len() as u64,
                                                      &mut n_b, &mut first_err) };
        if rc != 0 { return Err(self.eng.err("ipcfp_generate_proof_bundle", rc)); }
        if first_err != u64::MAX { return Err(anyhow!("proof spec {first_err} failed (the reference's `?` aborts the bundle there)")); }
        if n_p as usize > cap_p { return Err(anyhow!("{n_p} event proofs exceed the wrapper's buffer")); }
        // claim strings exactly as the generators format them (storage/generator.rs:158-178, events/generator.rs:274-293)
        let hex0x = |b: &[u8]| format!("0x{}", hex::encode(b));
        let mut storage_proofs = Vec::new();
        for (s, o) in storage_specs.iter().zip(&sout) {
            storage_proofs.push(StorageProof {
                child_epoch, child_block_cid: child_cid.to_string(), parent_state_root: cid_from_slot(&o.parent_state_root)?.to_string(),
                actor_id: s.actor_id, actor_state_cid: cid_from_slot(&o.actor_state_cid)?.to_string(),
                storage_root: cid_from_slot(&o.storage_root)?.to_string(), slot: hex0x(&s.slot.0), value: hex0x(&o.value) });
        }
        let locs: Vec<ipcfp_value_loc_t> = matches[..n_p as usize].iter().map(|m| m.event).collect();
        let stride = locs.iter().map(|l| l.len as usize).max().unwrap_or(0).max(1);
        let mut ev_bytes = vec![0u8; locs.len().max(1) * stride];
        if !locs.is_empty() {
            let rc = unsafe { ipcfp_witness_read_values(self.eng.ctx, self.raw(), locs.as_ptr(), locs.len() as u64, ev_bytes.as_mut_ptr(), stride as u64) };
            if rc != 0 { return Err(self.eng.err("ipcfp_witness_read_values", rc)); }
        }
        let mut event_proofs = Vec::new();
        for (k, m) in matches[..n_p as usize].iter().enumerate() {
            let stamped: fvm_shared::event::StampedEvent = fvm_ipld_encoding::from_slice(&ev_bytes[k * stride..k * stride + m.event.len as usize])?;
            let log = crate::proofs::common::evm::extract_evm_log(&stamped.event).ok_or_else(|| anyhow!("matched event is not an EVM log"))?;
            let mut slot = [0u8; 40];
            slot.copy_from_slice(&msg[k * 40..k * 40 + 40]);
            event_proofs.push(EventProof {
                parent_epoch, child_epoch, parent_tipset_cids: parent_cids.iter().map(|c| c.to_string()).collect(),
                child_block_cid: child_cid.to_string(), message_cid: cid_from_slot(&slot)?.to_string(),
                exec_index: m.exec_index, event_index: m.event_index,
                event_data: EventData { emitter: m.emitter, topics: log.topics.iter().map(|t| hex0x(t)).collect(), data: hex0x(&log.data) } });
        }
        let mut blocks = Vec::new();
        for k in 0..n_b as usize {
            let mut slot = [0u8; 40];
            slot.copy_from_slice(&wcids[k * 40..k * 40 + 40]);
            let cid = cid_from_slot(&slot)?;
            let data = self.get(&cid)?.ok_or_else(|| anyhow!("materialize: block {cid} vanished"))?;
            blocks.push(ProofBlock { cid, data });
        }
        Ok(UnifiedProofBundle { storage_proofs, event_proofs, blocks })
    }
}

/// The witness as the trait every fvm_ipld_amt / fvm_ipld_hamt call goes through
/// (src/proofs/common/blockstore.rs:26-39): unmodified AMT / HAMT callers can sit on the HBM-resident store.
impl Blockstore for Witness<'_> {
    fn get(&self, k: &Cid) -> Result<Option<Vec<u8>>> {
        let slot = cid_slot(k)?;
        let (mut len, mut found) = (0u64, 0 as c_int);
        let rc = unsafe { ipcfp_witness_get(self.eng.ctx, self.raw(), slot.as_ptr(), std::ptr::null_mut(), 0, &mut len, &mut found) };
        if rc != 0 { return Err(self.eng.err("ipcfp_witness_get", rc)); }
        if found == 0 { return Ok(None); }
        let mut out = vec![0u8; len as usize];
        let rc = unsafe { ipcfp_witness_get(self.eng.ctx, self.raw(), slot.as_ptr(), out.as_mut_ptr(), len, &mut len, &mut found) };
        if rc != 0 { return Err(self.eng.err("ipcfp_witness_get", rc)); }
        Ok(Some(out))
    }
    fn put_keyed(&self, k: &Cid, v: &[u8]) -> Result<()> {
        let slot = cid_slot(k)?;
        let (off, len) = ([0u64], [u32::try_from(v.len())?]);
        match unsafe { ipcfp_witness_put_keyed(self.eng.ctx, self.raw(), slot.as_ptr(), v.as_ptr(), off.as_ptr(), len.as_ptr(), 1) } { 0 => Ok(()), rc => Err(self.eng.err("ipcfp_witness_put_keyed", rc)) }
    }
    fn has(&self, k: &Cid) -> Result<bool> {
        let slot = cid_slot(k)?;
        let mut has = 0u8;
        match unsafe { ipcfp_witness_has(self.eng.ctx, self.raw(), slot.as_ptr(), 1, &mut has, std::ptr::null_mut()) } { 0 => Ok(has != 0), rc => Err(self.eng.err("ipcfp_witness_has", rc)) }
    }
}

// ---- owned C views of the claim structs -----------------------------------------------------------------------------
/// Owns the NUL-terminated copies of one EventProof's strings (src/proofs/events/bundle.rs:5-23).
struct CEventProof { parents: Vec<CString>, parent_ptrs: Vec<*const c_char>, child: CString, msg: CString,
                     topics: Vec<CString>, topic_ptrs: Vec<*const c_char>, data: CString,
                     epochs: (i64, i64), idx: (u64, u64), emitter: u64 }
impl CEventProof {
    fn new(p: &EventProof) -> Self {
        let parents: Vec<CString> = p.parent_tipset_cids.iter().map(|s| c_string(s)).collect();
        let topics: Vec<CString> = p.event_data.topics.iter().map(|s| c_string(s)).collect();
        Self { parent_ptrs: parents.iter().map(|c| c.as_ptr()).collect(), parents,
               child: c_string(&p.child_block_cid), msg: c_string(&p.message_cid),
               topic_ptrs: topics.iter().map(|c| c.as_ptr()).collect(), topics, data: c_string(&p.event_data.data),
               epochs: (p.parent_epoch, p.child_epoch), idx: (p.exec_index, p.event_index), emitter: p.event_data.emitter }
    }
    fn raw(&self) -> ipcfp_event_proof_t {
        ipcfp_event_proof_t { parent_epoch: self.epochs.0, child_epoch: self.epochs.1,
            parent_tipset_cids: self.parent_ptrs.as_ptr(), n_parent_tipset_cids: self.parent_ptrs.len() as u32,
            child_block_cid: self.child.as_ptr(), message_cid: self.msg.as_ptr(), exec_index: self.idx.0,
            event_index: self.idx.1, emitter: self.emitter, topics: self.topic_ptrs.as_ptr(),
            n_topics: self.topic_ptrs.len() as u32, data: self.data.as_ptr() }
    }
}

/// Owns the C strings of one StorageProof (src/proofs/storage/bundle.rs:5-14).
struct CStorageProof { s: [CString; 6], child_epoch: i64, actor_id: u64 }
impl CStorageProof {
    fn new(p: &StorageProof) -> Self {
        Self { s: [c_string(&p.child_block_cid), c_string(&p.parent_state_root), c_string(&p.actor_state_cid),
                   c_string(&p.storage_root), c_string(&p.slot), c_string(&p.value)],
               child_epoch: p.child_epoch, actor_id: p.actor_id }
    }
    fn raw(&self) -> ipcfp_storage_proof_t {
        ipcfp_storage_proof_t { child_epoch: self.child_epoch, child_block_cid: self.s[0].as_ptr(), parent_state_root: self.s[1].as_ptr(),
            actor_id: self.actor_id, actor_state_cid: self.s[2].as_ptr(), storage_root: self.s[3].as_ptr(),
            slot: self.s[4].as_ptr(), value: self.s[5].as_ptr() }
    }
}
