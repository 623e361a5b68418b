//! bindings/rust/ffi.rs — the Rust side of the drop-in boundary (NOT compiled in this repository:
//! the build image has no Rust toolchain; this is the file a maintainer of
//! consensus-shipyard/ipc-filecoin-proofs adds as `src/gpu/ffi.rs`, with `libipcfp.so` on the link
//! path: `println!("cargo:rustc-link-lib=dylib=ipcfp")` in build.rs).
//!
//! Declarations mirror `include/ipcfp.h` one to one; the safe wrappers keep the reference's names and
//! signatures (`verify_event_proof`, `verify_storage_proof`), so `src/proofs/verifier.rs` only swaps
//! which function it calls.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CString};

use anyhow::{anyhow, Result};

#[repr(C)] pub struct ipcfp_ctx_t { _p: [u8; 0] }
#[repr(C)] pub struct ipcfp_witness_t { _p: [u8; 0] }
#[repr(C)] pub struct ipcfp_bundle_t { _p: [u8; 0] }
#[repr(C)] pub struct ipcfp_packed_events_t { _p: [u8; 0] }

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

#[repr(C)] pub struct ipcfp_event_filter_t { pub topic0: [u8; 32], pub topic1: [u8; 32] }
#[repr(C)] pub struct ipcfp_value_loc_t { pub block: u32, pub off: u32, pub len: u32 }
#[repr(C)] pub struct ipcfp_event_match_t { pub exec_index: u64, pub event_index: u64, pub emitter: u64,
                                            pub event: ipcfp_value_loc_t, pub reserved: u32 }
#[repr(C)] pub struct ipcfp_generated_storage_t { pub parent_state_root: [u8; 40], pub actor_state_cid: [u8; 40],
                                                  pub storage_root: [u8; 40], pub value: [u8; 32], pub status: u32,
                                                  pub reserved: u32 }
#[repr(C)] pub struct ipcfp_trust_policy_t { pub kind: c_int, pub ec_chain_empty: c_int, pub min_epoch: i64, pub max_epoch: i64 }

extern "C" {
    pub fn ipcfp_ctx_create(device: c_int, out: *mut *mut ipcfp_ctx_t) -> c_int;
    pub fn ipcfp_ctx_destroy(ctx: *mut ipcfp_ctx_t);
    pub fn ipcfp_last_error(ctx: *const ipcfp_ctx_t) -> *const c_char;
    pub fn ipcfp_witness_create(ctx: *mut ipcfp_ctx_t, bytes: *const u8, nbytes: u64, off: *const u64, len: *const u32,
                                cids40: *const u8, n: u64, out: *mut *mut ipcfp_witness_t) -> c_int;
    pub fn ipcfp_witness_destroy(w: *mut ipcfp_witness_t);
    pub fn ipcfp_witness_verify_cids(ctx: *mut ipcfp_ctx_t, w: *mut ipcfp_witness_t, status: *mut u8, n_bad: *mut u64) -> c_int;
    pub fn ipcfp_create_event_filter(ctx: *mut ipcfp_ctx_t, sig: *const c_char, subnet: *const c_char,
                                     out: *mut ipcfp_event_filter_t) -> c_int;
    pub fn ipcfp_verify_event_proofs(ctx: *mut ipcfp_ctx_t, w: *mut ipcfp_witness_t, proofs: *const ipcfp_event_proof_t,
                                     n: u64, trust: *const ipcfp_trust_policy_t, filter: *const ipcfp_event_filter_t,
                                     status: *mut u8) -> c_int;
    pub fn ipcfp_verify_storage_proofs(ctx: *mut ipcfp_ctx_t, w: *mut ipcfp_witness_t, proofs: *const ipcfp_storage_proof_t,
                                       n: u64, trust: *const ipcfp_trust_policy_t, status: *mut u8) -> c_int;
    pub fn ipcfp_generate_event_proofs(ctx: *mut ipcfp_ctx_t, w: *mut ipcfp_witness_t, parent_cids40: *const u8, n_parents: u32,
                                       child_cid40: *const u8, filter: *const ipcfp_event_filter_t, has_actor: c_int, actor: u64,
                                       status_out: *mut u8, matches: *mut ipcfp_event_match_t, message_cids40: *mut u8,
                                       cap_proofs: u64, n_proofs: *mut u64, witness_block_ids: *mut u32, witness_cids40: *mut u8,
                                       cap_blocks: u64, n_blocks: *mut u64) -> c_int;
    pub fn ipcfp_generate_storage_proofs(ctx: *mut ipcfp_ctx_t, w: *mut ipcfp_witness_t, child_cid40: *const u8,
                                         actor_ids: *const u64, slots32: *const u8, n: u64, out: *mut ipcfp_generated_storage_t,
                                         witness_block_ids: *mut u32, witness_cids40: *mut u8, cap_blocks: u64,
                                         n_blocks: *mut u64) -> c_int;
    pub fn ipcfp_bundle_parse_json(ctx: *mut ipcfp_ctx_t, json: *const c_char, len: u64, flags: u32,
                                   out: *mut *mut ipcfp_bundle_t) -> c_int;
    pub fn ipcfp_bundle_destroy(b: *mut ipcfp_bundle_t);
    pub fn ipcfp_bundle_event_count(b: *const ipcfp_bundle_t) -> u64;
    pub fn ipcfp_bundle_storage_count(b: *const ipcfp_bundle_t) -> u64;
    pub fn ipcfp_verify_proof_bundle(ctx: *mut ipcfp_ctx_t, b: *mut ipcfp_bundle_t, trust: *const ipcfp_trust_policy_t,
                                     filter: *const ipcfp_event_filter_t, storage_status: *mut u8,
                                     event_status: *mut u8) -> c_int;
    // host-only, parallel lowering of the reference's structs to the packed claim form (no device involved)
    pub fn ipcfp_pack_event_proofs(proofs: *const ipcfp_event_proof_t, n: u64, out: *mut *mut ipcfp_packed_events_t) -> c_int;
    pub fn ipcfp_packed_events_destroy(p: *mut ipcfp_packed_events_t);
    pub fn ipcfp_pack_storage_proofs(proofs: *const ipcfp_storage_proof_t, n: u64, claims: *mut u8 /* n × ipcfp_storage_claim_t */) -> c_int;
    // … the remaining primitives (ipcfp_amt_get, ipcfp_hamt_get, ipcfp_scan_events, ipcfp_exec_order,
    //   ipcfp_*_batch, ipcfp_verify_event_claims_device, profiling) bind the same way.
}

/// `status >= 64` is `Err`; the reference aborts the whole bundle at the first one
/// (src/proofs/events/verifier.rs:62-71, src/proofs/verifier.rs:19-28).
fn statuses_to_result(st: &[u8]) -> Result<Vec<bool>> {
    if let Some((i, s)) = st.iter().enumerate().find(|(_, s)| **s >= 64) {
        return Err(anyhow!("proof {i}: verification error (ipcfp status {s})"));
    }
    Ok(st.iter().map(|s| *s == 1).collect())
}

pub struct Engine { ctx: *mut ipcfp_ctx_t }
pub struct Witness<'e> { eng: &'e Engine, w: *mut ipcfp_witness_t }

impl Engine {
    pub fn new(device: i32) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        match unsafe { ipcfp_ctx_create(device, &mut ctx) } { 0 => Ok(Self { ctx }), rc => Err(anyhow!("ipcfp_ctx_create: {rc}")) }
    }
    /// replaces `load_witness_store(blocks)` (src/proofs/events/verifier.rs:79-89)
    pub fn load_witness_store(&self, blocks: &[crate::proofs::common::bundle::ProofBlock]) -> Result<Witness<'_>> {
        let (mut bytes, mut off, mut len, mut cids) = (Vec::new(), Vec::new(), Vec::new(), Vec::new());
        for b in blocks {
            off.push(bytes.len() as u64); len.push(b.data.len() as u32); bytes.extend_from_slice(&b.data);
            let mut slot = [0u8; 40]; let c = b.cid.to_bytes();
            if c.len() > 40 { return Err(anyhow!("CID longer than 40 bytes")); }
            slot[..c.len()].copy_from_slice(&c); cids.extend_from_slice(&slot);
        }
        let mut w = std::ptr::null_mut();
        let rc = unsafe { ipcfp_witness_create(self.ctx, bytes.as_ptr(), bytes.len() as u64, off.as_ptr(), len.as_ptr(),
                                               cids.as_ptr(), blocks.len() as u64, &mut w) };
        if rc != 0 { return Err(anyhow!("ipcfp_witness_create: {rc}")); }
        Ok(Witness { eng: self, w })
    }
}
impl Engine {
    /// `serde_json::from_str::<UnifiedProofBundle>(text)` + `verify_proof_bundle(&bundle, policy, filter)`
    /// (src/proofs/verifier.rs:12-62) without materialising the blocks on the host: the base64 of every
    /// `ProofBlock.data` is decoded on the device straight into the witness arena.
    pub fn verify_proof_bundle_json(&self, text: &str, trust: &ipcfp_trust_policy_t,
                                    filter: Option<&ipcfp_event_filter_t>) -> Result<(Vec<bool>, Vec<bool>)> {
        let mut b = std::ptr::null_mut();
        let rc = unsafe { ipcfp_bundle_parse_json(self.ctx, text.as_ptr() as *const c_char, text.len() as u64, 0, &mut b) };
        if rc != 0 { return Err(anyhow!("bundle JSON: {rc}")); }
        let (ns, ne) = unsafe { (ipcfp_bundle_storage_count(b) as usize, ipcfp_bundle_event_count(b) as usize) };
        let (mut ss, mut es) = (vec![0u8; ns.max(1)], vec![0u8; ne.max(1)]);
        let rc = unsafe { ipcfp_verify_proof_bundle(self.ctx, b, trust, filter.map_or(std::ptr::null(), |f| f as *const _),
                                                    ss.as_mut_ptr(), es.as_mut_ptr()) };
        unsafe { ipcfp_bundle_destroy(b) };
        if rc != 0 { return Err(anyhow!("ipcfp_verify_proof_bundle: {rc}")); }
        // storage proofs are checked first and the first Err aborts (verifier.rs:19-28)
        Ok((statuses_to_result(&ss[..ns])?, statuses_to_result(&es[..ne])?))
    }
}
impl Drop for Engine { fn drop(&mut self) { unsafe { ipcfp_ctx_destroy(self.ctx) } } }
impl Drop for Witness<'_> { fn drop(&mut self) { unsafe { ipcfp_witness_destroy(self.w) } } }

impl Witness<'_> {
    /// drop-in for `verify_event_proof` (src/proofs/events/verifier.rs:51-56).  The trust closures are
    /// pure in (epoch, cid) for `TrustPolicy::{AcceptAll, F3Certificate}` and travel as a POD; an
    /// arbitrary `check_event` closure still runs on the host over the statuses that come back TRUE.
    pub fn verify_event_proof(&self, bundle: &crate::proofs::events::bundle::EventProofBundle,
                              trust: &ipcfp_trust_policy_t, filter: Option<&ipcfp_event_filter_t>) -> Result<Vec<bool>> {
        let keep: Vec<_> = bundle.proofs.iter().map(CProof::new).collect();   // owns the CStrings
        let raw: Vec<ipcfp_event_proof_t> = keep.iter().map(|k| k.raw()).collect();
        let mut st = vec![0u8; raw.len()];
        let rc = unsafe { ipcfp_verify_event_proofs(self.eng.ctx, self.w, raw.as_ptr(), raw.len() as u64, trust,
                                                    filter.map_or(std::ptr::null(), |f| f as *const _), st.as_mut_ptr()) };
        if rc != 0 { return Err(anyhow!("ipcfp_verify_event_proofs: {rc}")); }
        statuses_to_result(&st)
    }
}

/// Owns the NUL-terminated copies of one EventProof's strings.
struct CProof { parents: Vec<CString>, parent_ptrs: Vec<*const c_char>, child: CString, msg: CString,
                topics: Vec<CString>, topic_ptrs: Vec<*const c_char>, data: CString,
                epochs: (i64, i64), idx: (u64, u64), emitter: u64 }
impl CProof {
    fn new(p: &crate::proofs::events::bundle::EventProof) -> Self {
        let parents: Vec<CString> = p.parent_tipset_cids.iter().map(|s| CString::new(s.as_str()).unwrap()).collect();
        let topics: Vec<CString> = p.event_data.topics.iter().map(|s| CString::new(s.as_str()).unwrap()).collect();
        Self { parent_ptrs: parents.iter().map(|c| c.as_ptr()).collect(), parents,
               child: CString::new(p.child_block_cid.as_str()).unwrap(), msg: CString::new(p.message_cid.as_str()).unwrap(),
               topic_ptrs: topics.iter().map(|c| c.as_ptr()).collect(), topics,
               data: CString::new(p.event_data.data.as_str()).unwrap(),
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
