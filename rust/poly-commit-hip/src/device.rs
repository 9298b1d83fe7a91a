//! Process-wide device state behind the trait impls: one `pc_ctx` per process (device `PC_HIP_DEVICE`, default 0), the
//! committer keys that are resident in HBM, and the device copies of polynomials that `commit` leaves behind for `open`.
//!
//! The reference's associated types carry no device handles (`marlin_pc::CommitterKey<E>` is a `Vec<G1Affine>`,
//! `marlin_pc/data_structures.rs:26-44`; the commitment state is the hiding randomness), and this crate keeps those
//! types unchanged so that keys / states stay interchangeable with the reference.  Residency is therefore tracked on the
//! side, keyed by WHERE the host data lives and WHAT it looks like:
//!
//! * [`resident`] maps a slice of bases (`ck.powers`, a `shifted_powers(bound)` tail, `ck.comm_key`) to a resident
//!   `pc_srs` + `base_offset`.  The first time a key allocation is seen it is uploaded (the `trim` hook the survey asks
//!   for, taken lazily: `marlin_pc/mod.rs:80-169` returns a plain struct) and its window table is built
//!   (`pc_hip_srs_precompute`); later calls find it by address range, length and a fingerprint of sampled points.
//!   Resident keys are bounded by a BYTE budget (`PC_HIP_KEY_BUDGET_MB`, default 65536: bases + window tables + fold tables +
//!   pipelines as `pc_hip_srs_bytes_resident` reports them) and by `PC_HIP_MAX_KEYS`; least recently used keys go first, and
//!   [`release`] drops a key explicitly (`HipMarlinKZG10::release(&ck)`).
//! * [`device_poly`] does the same for coefficient vectors.  It is OPT-IN (`PC_HIP_POLY_CACHE_MB`, default 0 = off): with it
//!   `commit` uploads a polynomial once and `open` of the same `&LabeledPolynomial` finds the device copy and sends nothing
//!   over PCIe -- but a cached copy is identified by address, length and 64 sampled coefficients, so a caller that rewrites a
//!   polynomial IN PLACE (or frees it and allocates another at the same address) without touching a sampled coefficient would
//!   get a proof for the OLD polynomial (round-3 advisor finding).  Only a caller whose polynomials are immutable between
//!   `commit` and `open` should turn it on.  With the cache off `commit` and `open` hand the host slice to `pc_hip_msm` /
//!   `pc_hip_kzg_open`, which overlap the PCIe copy with the MSM themselves (ONE MSM in parts from 2^21 coefficients: a commit of 2^24 host
//!   coefficients costs 40 ms against 38.5 ms for resident ones, so the cache below buys little and stays off by default).
use ark_poly_commit::Error;
use core::ffi::{c_int, c_void};
use std::collections::VecDeque;
use std::ffi::CStr;
use std::sync::{Arc, Mutex, OnceLock};

use crate::curve::{pack_points, pack_scalars, HipCurve, HipField};
use crate::ffi;

/// The process's device context.  `pc_ctx` serialises its calls with an internal mutex, so sharing it between rayon
/// threads is safe (Hyrax calls the MSM from inside `par_iter`, `hyrax/mod.rs:233-242`).
pub struct Ctx {
    pub raw: *mut ffi::pc_ctx,
}
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}

static CTX: OnceLock<Result<Ctx, String>> = OnceLock::new();

fn env_usize(name: &str, default: usize) -> usize {
    std::env::var(name).ok().and_then(|v| v.parse().ok()).unwrap_or(default)
}

/// The context, created on first use.  Without a GPU this is an error (`PC_ERR_NO_DEVICE`), never a silent fallback:
/// the caller decides (the trait impls fall back to the reference's CPU path only below [`min_pairs`]).
pub fn ctx() -> Result<&'static Ctx, Error> {
    let r = CTX.get_or_init(|| {
        let mut raw = core::ptr::null_mut();
        let rc = unsafe { ffi::pc_hip_init(env_usize("PC_HIP_DEVICE", 0) as c_int, &mut raw) };
        if rc != ffi::PC_OK {
            return Err(format!("pc_hip_init: {}", strerror(rc)));
        }
        Ok(Ctx { raw })
    });
    r.as_ref().map_err(|e| Error::InvalidParameters(e.clone()))
}

pub fn strerror(rc: c_int) -> String {
    unsafe { CStr::from_ptr(ffi::pc_hip_strerror(rc)) }.to_string_lossy().into_owned()
}

/// `pc_status` -> `Result`; the HIP error string of the context rides along (`Error::InvalidParameters`, `error.rs:117`).
pub fn check(ctx: &Ctx, rc: c_int) -> Result<(), Error> {
    if rc == ffi::PC_OK {
        return Ok(());
    }
    let last = unsafe { CStr::from_ptr(ffi::pc_hip_last_error(ctx.raw)) }.to_string_lossy().into_owned();
    Err(Error::InvalidParameters(format!("pc_hip: {} ({})", strerror(rc), last)))
}

/// MSMs shorter than this stay on `ark-ec`'s CPU `msm_bigint` (`PC_HIP_MIN_PAIRS`; default 2^10: in the repository's
/// `workloads.latency` sweep the blocking device path wins from 2^10 coefficients upward against the CPU port).
pub fn min_pairs() -> usize {
    static V: OnceLock<usize> = OnceLock::new();
    *V.get_or_init(|| env_usize("PC_HIP_MIN_PAIRS", 1 << 10))
}

// ------------------------------------------------------------------------------------------------------------------
// resident keys
// ------------------------------------------------------------------------------------------------------------------
pub struct ResidentKey {
    pub srs: *mut ffi::pc_srs,
    pub n: usize,
    /// set by the first IPA opening on this key: the fold table of its upper half (pc_hip_srs_precompute_fold) exists or was tried
    pub fold_table_built: std::sync::atomic::AtomicBool,
    host_addr: usize,
    host_bytes: usize,
    elem_bytes: usize,
    fingerprint: Vec<u64>,
}
unsafe impl Send for ResidentKey {}
unsafe impl Sync for ResidentKey {}
impl Drop for ResidentKey {
    fn drop(&mut self) {
        unsafe { ffi::pc_hip_srs_free(self.srs) }
    }
}

static KEYS: OnceLock<Mutex<VecDeque<Arc<ResidentKey>>>> = OnceLock::new();

fn sample_positions(n: usize, k: usize) -> impl Iterator<Item = usize> {
    // first, last and k - 2 positions spread by a fixed odd stride: cheap, deterministic, hits every region of the vector
    let k = k.min(n);
    (0..k).map(move |i| if i == 0 { 0 } else if i + 1 == k { n - 1 } else { (i * 0x9E37_79B1usize) % n })
}

fn fingerprint_points<G: HipCurve>(pts: &[G]) -> Vec<u64> {
    let w = 2 * G::FQ_LIMBS;
    let mut out = Vec::with_capacity(8 * w);
    let mut buf = vec![0u64; w];
    for i in sample_positions(pts.len(), 8) {
        pts[i].write_xy(&mut buf);
        out.extend_from_slice(&buf);
    }
    out
}

/// The resident key that holds `bases` and the offset of `bases[0]` inside it; uploads the slice as a new key when no
/// resident allocation contains it.  At most `PC_HIP_MAX_KEYS` (default 8) keys stay resident, least recently used out.
pub fn resident<G: HipCurve>(bases: &[G]) -> Result<(Arc<ResidentKey>, usize), Error> {
    let ctx = ctx()?;
    let elem = core::mem::size_of::<G>();
    let addr = bases.as_ptr() as usize;
    let keys = KEYS.get_or_init(|| Mutex::new(VecDeque::new()));
    let mut q = keys.lock().unwrap();
    // a slice inside a known allocation (ck.powers[lz..], shifted_powers[max_bound - bound ..]): same key, an offset
    let hit = q.iter().position(|k| {
        k.elem_bytes == elem && addr >= k.host_addr && addr + bases.len() * elem <= k.host_addr + k.host_bytes && (addr - k.host_addr) % elem == 0
    });
    if let Some(i) = hit {
        let k = q.remove(i).unwrap();
        let off = (addr - k.host_addr) / elem;
        // Still the same data?  Only `bases` itself is known to be live memory (the rest of the old allocation may have been
        // freed), so the check reads nothing outside it: its first and last point against the DEVICE copy at those offsets
        // (two 96-byte reads), and the sampled positions of the upload-time fingerprint that fall inside the slice.
        let same = !bases.is_empty() && {
            let w = 2 * G::FQ_LIMBS;
            let (mut dev, mut host) = (vec![0u64; w], vec![0u64; w]);
            let mut ok = true;
            for idx in [0usize, bases.len() - 1] {
                bases[idx].write_xy(&mut host);
                ok &= unsafe { ffi::pc_hip_srs_read(ctx.raw, k.srs, off + idx, 1, dev.as_mut_ptr() as *mut c_void) } == ffi::PC_OK && dev == host;
            }
            for (j, pos) in sample_positions(k.n, 8).enumerate() {
                if pos >= off && pos < off + bases.len() {
                    bases[pos - off].write_xy(&mut host);
                    ok &= host[..] == k.fingerprint[j * w..(j + 1) * w];
                }
            }
            ok
        };
        if same {
            q.push_front(k.clone());
            return Ok((k, off));
        }
        // the allocation was reused for other data: forget the stale key (dropped here)
    }
    let mut srs = core::ptr::null_mut();
    let rc = if G::layout_is_abi() {
        unsafe { ffi::pc_hip_srs_upload(ctx.raw, G::CURVE, bases.as_ptr() as *const c_void, bases.len(), elem, ffi::PC_MEM_HOST, &mut srs) }
    } else {
        let packed = pack_points(bases);
        unsafe { ffi::pc_hip_srs_upload(ctx.raw, G::CURVE, packed.as_ptr() as *const c_void, bases.len(), 0, ffi::PC_MEM_HOST, &mut srs) }
    };
    check(ctx, rc)?;
    // A prover's key is long-lived: spend (bits / c + 1) x the key in HBM on the window table (same results, ~15-30 % faster
    // MSMs).  A failure (PC_ERR_OOM on a shared GPU) only means the table-free path stays in use.
    if bases.len() >= env_usize("PC_HIP_TABLE_MIN_POINTS", 1 << 12) {
        let _ = unsafe { ffi::pc_hip_srs_precompute(ctx.raw, srs, 0, 0) };
    }
    let key = Arc::new(ResidentKey { srs, n: bases.len(), fold_table_built: std::sync::atomic::AtomicBool::new(false), host_addr: addr, host_bytes: bases.len() * elem, elem_bytes: elem, fingerprint: fingerprint_points(bases) });
    q.push_front(key.clone());
    evict_over_budget(&mut q);
    Ok((key, 0))
}

/// Device bytes of one resident key: bases + window table(s) + fold table + the workspaces of its MSM pipelines.
pub fn key_bytes(k: &ResidentKey) -> usize {
    let mut b = [0usize; 4];
    if unsafe { ffi::pc_hip_srs_bytes_resident(k.srs, b.as_mut_ptr()) } == ffi::PC_OK { b.iter().sum() } else { k.host_bytes }
}

/// Least recently used keys leave until the cache holds at most `PC_HIP_MAX_KEYS` (default 8) keys and `PC_HIP_KEY_BUDGET_MB`
/// (default 65536) of device memory; the key just used always stays.  A key still referenced by a running call is freed when
/// that call drops its `Arc`.
fn evict_over_budget(q: &mut VecDeque<Arc<ResidentKey>>) {
    let max_keys = env_usize("PC_HIP_MAX_KEYS", 8).max(1);
    let budget = env_usize("PC_HIP_KEY_BUDGET_MB", 65536) << 20;
    while q.len() > max_keys {
        q.pop_back();
    }
    let mut total: usize = q.iter().map(|k| key_bytes(k)).sum();
    while total > budget && q.len() > 1 {
        total -= q.pop_back().map(|k| key_bytes(&k)).unwrap_or(0);
    }
}

/// Drop the resident copy (bases, tables, pipelines) of the key allocation that contains `bases`, if there is one: the explicit
/// counterpart of the lazy upload (`HipMarlinKZG10::release(&ck)`).  Returns whether a key was dropped.
pub fn release<G: HipCurve>(bases: &[G]) -> bool {
    let elem = core::mem::size_of::<G>();
    let addr = bases.as_ptr() as usize;
    let Some(keys) = KEYS.get() else { return false };
    let mut q = keys.lock().unwrap();
    let before = q.len();
    q.retain(|k| !(k.elem_bytes == elem && addr >= k.host_addr && addr < k.host_addr + k.host_bytes.max(1)));
    before != q.len()
}

/// Device memory the library holds for this process's context, by kind (`pc_hip_ctx_bytes_resident`): `[all, key bases, window
/// tables, fold tables, staging + scratch, number of keys]`.
pub fn bytes_resident() -> Result<[usize; 6], Error> {
    let c = ctx()?;
    let mut out = [0usize; 6];
    check(c, unsafe { ffi::pc_hip_ctx_bytes_resident(c.raw, out.as_mut_ptr()) })?;
    Ok(out)
}

/// Give back staging buffers, scratch and cached working keys (`pc_hip_ctx_trim`); they come back on demand.
pub fn trim() -> Result<(), Error> {
    let c = ctx()?;
    check(c, unsafe { ffi::pc_hip_ctx_trim(c.raw) })
}

// ------------------------------------------------------------------------------------------------------------------
// device copies of polynomials
// ------------------------------------------------------------------------------------------------------------------
pub struct DevicePoly {
    pub dev: *mut c_void,
    pub n: usize,
    host_addr: usize,
    fingerprint: Vec<u64>,
}
unsafe impl Send for DevicePoly {}
unsafe impl Sync for DevicePoly {}
impl Drop for DevicePoly {
    fn drop(&mut self) {
        if let Ok(c) = ctx() {
            unsafe { ffi::pc_hip_free(c.raw, self.dev) };
        }
    }
}
impl DevicePoly {
    /// Uninitialised device buffer of `n` field elements (quotients, combinations, IPA work vectors).
    pub fn alloc(n: usize) -> Result<Self, Error> {
        let c = ctx()?;
        let mut dev = core::ptr::null_mut();
        check(c, unsafe { ffi::pc_hip_malloc(c.raw, n.max(1) * 32, &mut dev) })?;
        Ok(Self { dev, n, host_addr: 0, fingerprint: Vec::new() })
    }
    pub fn upload<F: HipField>(coeffs: &[F]) -> Result<Self, Error> {
        let c = ctx()?;
        let p = Self::alloc(coeffs.len())?;
        let rc = if F::layout_is_abi() {
            unsafe { ffi::pc_hip_memcpy_h2d(c.raw, p.dev, coeffs.as_ptr() as *const c_void, coeffs.len() * 32) }
        } else {
            let packed = pack_scalars(coeffs);
            unsafe { ffi::pc_hip_memcpy_h2d(c.raw, p.dev, packed.as_ptr() as *const c_void, coeffs.len() * 32) }
        };
        check(c, rc)?;
        Ok(p)
    }
    pub fn download<F: HipField>(&self, n: usize) -> Result<Vec<F>, Error> {
        let c = ctx()?;
        let mut raw = vec![[0u64; 4]; n];
        check(c, unsafe { ffi::pc_hip_memcpy_d2h(c.raw, raw.as_mut_ptr() as *mut c_void, self.dev, n * 32) })?;
        Ok(raw.into_iter().map(F::from_mont_limbs).collect())
    }
    /// Device address of element `i`.
    pub fn at(&self, i: usize) -> *mut c_void {
        (self.dev as usize + 32 * i) as *mut c_void
    }
}

static POLYS: OnceLock<Mutex<VecDeque<Arc<DevicePoly>>>> = OnceLock::new();

fn fingerprint_scalars<F: HipField>(c: &[F]) -> Vec<u64> {
    sample_positions(c.len(), 64).flat_map(|i| c[i].to_mont_limbs()).collect()
}

/// Bytes the polynomial cache may hold (`PC_HIP_POLY_CACHE_MB`; default 0: the cache is off, see the module documentation).
pub fn poly_cache_bytes() -> usize {
    static V: OnceLock<usize> = OnceLock::new();
    *V.get_or_init(|| env_usize("PC_HIP_POLY_CACHE_MB", 0) << 20)
}

/// true iff the opt-in polynomial cache is on (`PC_HIP_POLY_CACHE_MB` > 0)
pub fn poly_cache_enabled() -> bool {
    poly_cache_bytes() > 0
}

/// The device copy of `coeffs`: found by address, length and fingerprint, else uploaded and remembered (LRU, bounded by
/// `PC_HIP_POLY_CACHE_MB`).  With the cache on, `commit` calls this and `open` of the same polynomial sends nothing over PCIe.
pub fn device_poly<F: HipField>(coeffs: &[F]) -> Result<Arc<DevicePoly>, Error> {
    let cap = poly_cache_bytes();
    if cap == 0 || coeffs.is_empty() {
        return Ok(Arc::new(DevicePoly::upload(coeffs)?));
    }
    let addr = coeffs.as_ptr() as usize;
    let fp = fingerprint_scalars(coeffs);
    let cache = POLYS.get_or_init(|| Mutex::new(VecDeque::new()));
    let mut q = cache.lock().unwrap();
    if let Some(i) = q.iter().position(|p| p.host_addr == addr && p.n == coeffs.len() && p.fingerprint == fp) {
        let p = q.remove(i).unwrap();
        q.push_front(p.clone());
        return Ok(p);
    }
    let mut p = DevicePoly::upload(coeffs)?;
    p.host_addr = addr;
    p.fingerprint = fp;
    let p = Arc::new(p);
    q.push_front(p.clone());
    let mut bytes: usize = q.iter().map(|p| p.n * 32).sum();
    while bytes > cap && q.len() > 1 {
        bytes -= q.pop_back().map(|p| p.n * 32).unwrap_or(0);
    }
    Ok(p)
}
