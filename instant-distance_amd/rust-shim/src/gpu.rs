//! `mod gpu` — the MI355X engine (libidist.so, include/idist.h) behind instant-distance's own types.
//!
//! SOURCE ONLY: this image has no Rust toolchain, so this module has never been compiled here.  The same
//! ABI is exercised from C++ (`host/instant_distance.hpp`, tests/host/all.cpp) and Python (ctypes).
//!
//! This file is ADDED to the reference crate (`instant-distance/src/gpu.rs`); `lib.rs.patch` next to it
//! is the whole change to the reference's own files.  Nothing is taken away:
//!
//! * `trait Point` gains two DEFAULTED items (`as_f32() -> None`, `METRIC = L2Sq`): every existing
//!   `impl Point` keeps compiling and keeps running the crate's CPU code (C1: the `isize` points of
//!   `tests/all.rs`/`benches`, any custom `distance()`);
//! * a point type that returns `Some(&[f32])` from `as_f32` (the binding's `FloatArray`, `[f32; N]`
//!   wrappers) is built and searched on the GPU;
//! * `Hnsw`'s own fields (`ef_search`, `points`, `zero`, `layers`, lib.rs:194-199) are always filled —
//!   after a GPU build the graph is exported into them — so `with-serde` (lib.rs:130, 193; types.rs),
//!   `iter`, `get`, `Index` work unchanged; the device index is a `#[serde(skip)]` cache that a
//!   deserialised `Hnsw` re-creates from those fields on its first search (`idist_index_import`).
//!
//! Line references `lib.rs:NN` / `types.rs:NN` are to /root/reference/instant-distance/src/.

use std::ffi::CStr;
use std::os::raw::c_char;
use std::sync::atomic::{AtomicU64, Ordering};
use std::sync::OnceLock;

use crate::types::{UpperNode, ZeroNode, INVALID};
use crate::{Builder, Candidate, Point, PointId, M};

// ---- include/idist.h -------------------------------------------------------------------------
#[repr(C)]
#[derive(Clone, Copy)]
pub(crate) struct IdistConfig {
    ef_search: u32,
    ef_construction: u32,
    ml: f32,
    has_heuristic: i32,
    extend_candidates: i32,
    keep_pruned: i32,
    metric: i32,
    max_batch: u32,    // 0 = concurrent inserts (the rayon schedule of lib.rs:316-318), 1 = the sequential loop
    tie_policy: i32,   // 0 = strict (the reference's results whatever the ties: the tie region grows, then spills to HBM), 1 = drop: see include/idist.h
    tie_capacity: u32, // 0 = 64
}
#[repr(C)] pub(crate) struct IdistIndex { _p: [u8; 0] }
#[repr(C)] pub(crate) struct IdistSearchCtx { _p: [u8; 0] }
#[repr(C)] pub(crate) struct IdistProgress { _p: [u8; 0] }
const IDIST_MAX_LAYERS: usize = 64;
#[repr(C)]
struct IdistIndexInfo {     // idist_index_info
    n: u32, dim: u32, row_stride: u32, n_upper: u32, ef_search: u32, metric: i32, device: i32,
    layer_len: [u32; IDIST_MAX_LAYERS], tie_capacity: u32,
}

extern "C" {
    fn idist_last_error() -> *const c_char;
    fn idist_device_count(out: *mut i32) -> i32;
    fn idist_default_config(cfg: *mut IdistConfig) -> i32;
    fn idist_index_build(points: *const f32, n: u32, dim: u32, cfg: *const IdistConfig, device: i32,
                         out: *mut *mut IdistIndex) -> i32;
    fn idist_index_import(points: *const f32, n: u32, dim: u32, cfg: *const IdistConfig, zero: *const u32,
                          layers: *const *const u32, layer_len: *const u32, n_upper: u32, device: i32,
                          out: *mut *mut IdistIndex) -> i32;
    fn idist_index_export(idx: *const IdistIndex, zero: *mut u32, layers: *const *mut u32) -> i32;
    fn idist_index_get_info(idx: *const IdistIndex, out: *mut IdistIndexInfo) -> i32;
    fn idist_index_free(idx: *mut IdistIndex);
    fn idist_search_ctx_new(idx: *const IdistIndex, slots: u32, out: *mut *mut IdistSearchCtx) -> i32;
    fn idist_search_ctx_free(ctx: *mut IdistSearchCtx);
    fn idist_search_batch(idx: *const IdistIndex, ctx: *mut IdistSearchCtx, queries: *const f32, nq: u32,
                          out_pid: *mut u32, out_dist: *mut f32, out_count: *mut u32,
                          out_counters: *mut u32) -> i32;
    // several GPUs of one node: replicate once, shard the queries of a batch
    fn idist_replicate(root: *const IdistIndex, devices: *const i32, n_devices: u32, replicas: *mut *mut IdistIndex) -> i32;
    // the same as one RCCL broadcast per buffer (ncclCommInitAll over the devices, inside libidist)
    fn idist_replicate_rccl(root: *const IdistIndex, devices: *const i32, n_devices: u32, replicas: *mut *mut IdistIndex,
                            seconds: *mut f64) -> i32;
    fn idist_search_batch_sharded(replicas: *const *const IdistIndex, ctxs: *const *mut IdistSearchCtx, n_devices: u32,
                                  queries: *const f32, nq: u32, out_pid: *mut u32, out_dist: *mut f32,
                                  out_count: *mut u32, out_counters: *mut u32) -> i32;
    #[cfg(feature = "indicatif")] fn idist_progress_new(out: *mut *mut IdistProgress) -> i32;
    #[cfg(feature = "indicatif")] fn idist_progress_free(p: *mut IdistProgress);
    #[cfg(feature = "indicatif")] fn idist_progress_watch_next_build(p: *mut IdistProgress) -> i32;
    #[cfg(feature = "indicatif")] fn idist_progress_get(p: *const IdistProgress, done: *mut u64, total: *mut u64, layer: *mut i32) -> i32;
}

fn expect(status: i32) {
    // the reference API is infallible (it only panics at lib.rs:256 and :148)
    if status != 0 {
        let msg = unsafe { CStr::from_ptr(idist_last_error()) }.to_string_lossy().into_owned();
        panic!("libidist: status {status}: {msg}");
    }
}

/// Which of the two distances the reference ships a point type computes (`Point::METRIC`).
#[derive(Clone, Copy, Debug, Eq, PartialEq)]
pub enum Metric {
    /// `FloatArray::distance`, instant-distance-py/src/lib.rs:378-421 (squared L2, eight FMA chains)
    L2Sq = 0,
    /// the sqrt of it: `tests/all.rs:93-97`, `examples/colors.rs:21-25`
    L2 = 1,
}

/// Hardware queues: every `Search` owns a HIP stream, and the runtime multiplexes a process's streams onto
/// `GPU_MAX_HW_QUEUES` queues (default 4, read once when the HIP runtime starts).  A host with more than four searching
/// threads sets it before its first call into this module — `std::env::set_var("GPU_MAX_HW_QUEUES", "16")` at the top of
/// `main`, or in the service's environment.  libidist does not edit the process environment itself.
///
/// True if a gfx950 device and libidist.so are usable; `Hnsw::new` falls back to the CPU code otherwise.
pub fn available() -> bool {
    let mut n = 0i32;
    unsafe { idist_device_count(&mut n) == 0 && n > 0 }
}

/// The device-side index: immutable after build/import, shared by all searching threads (`&self`, lib.rs:352).
pub(crate) struct GpuIndex { idx: *mut IdistIndex, uid: u64, dim: usize }
static NEXT_UID: AtomicU64 = AtomicU64::new(1);
impl GpuIndex {
    // uid: what a `Search` remembers its device context by (the engine checks its own uid of the index as well)
    fn new(idx: *mut IdistIndex, dim: usize) -> Self { Self { idx, uid: NEXT_UID.fetch_add(1, Ordering::Relaxed), dim } }
}
unsafe impl Sync for GpuIndex {}
unsafe impl Send for GpuIndex {}
impl Drop for GpuIndex { fn drop(&mut self) { unsafe { idist_index_free(self.idx) } } }

/// `#[serde(skip)] gpu: gpu::Cache` in `struct Hnsw`: `None` inside = this point type / machine has no GPU path.
/// The second field is the device the index lives on / is re-imported onto: `Builder::device` for an index built here,
/// `Hnsw::on_device` for one that came out of `serde` (0 until then).
#[derive(Default)]
pub(crate) struct Cache(OnceLock<Option<GpuIndex>>, std::sync::atomic::AtomicI32);

fn config(b: &Builder, metric: Metric) -> IdistConfig {
    let mut cfg = unsafe { std::mem::zeroed::<IdistConfig>() };
    expect(unsafe { idist_default_config(&mut cfg) });
    cfg.ef_search = b.ef_search as u32;
    cfg.ef_construction = b.ef_construction as u32;
    cfg.ml = b.ml;
    cfg.has_heuristic = b.heuristic.is_some() as i32;
    if let Some(h) = b.heuristic {
        cfg.extend_candidates = h.extend_candidates as i32;
        cfg.keep_pruned = h.keep_pruned as i32;
    }
    cfg.metric = metric as i32;
    cfg.max_batch = b.gpu_max_batch;     // new Builder field, default 0
    cfg
}

/// Row-major f32 copy of the points, or None if the type has no f32 view / the rows are ragged.
fn flatten<P: Point>(points: &[P]) -> Option<(Vec<f32>, usize)> {
    let dim = points.first()?.as_f32()?.len();
    if dim == 0 { return None; }
    let mut flat = Vec::with_capacity(points.len() * dim);
    for p in points {
        let v = p.as_f32()?;
        if v.len() != dim { return None; }
        flat.extend_from_slice(v);
    }
    Some((flat, dim))
}

fn info(idx: *const IdistIndex) -> IdistIndexInfo {
    let mut i = unsafe { std::mem::zeroed::<IdistIndexInfo>() };
    expect(unsafe { idist_index_get_info(idx, &mut i) });
    i
}

/// Replaces lib.rs:238-250 and :275-345 (layer sizing, the per-layer insertion loop with
/// `Construction::insert` :437-528 and the heuristics :616-698) when the point type has an f32 view.
/// `points` are already in PointId order: the shuffle (:214, :257-270) stays where it is, with the real `rand`.
/// Returns the graph in the crate's own node types plus the device index that built it.
pub(crate) fn try_build<P: Point>(points: &[P], b: &Builder) -> Option<(Vec<ZeroNode>, Vec<Vec<UpperNode>>, Cache)> {
    if b.device < 0 || !available() { return None; }
    let (flat, dim) = flatten(points)?;
    let cfg = config(b, P::METRIC);
    let mut idx = std::ptr::null_mut();
    #[cfg(feature = "indicatif")]
    let watch = b.progress.clone().map(|bar| progress::Watch::start(bar, points.len() as u64));
    expect(unsafe { idist_index_build(flat.as_ptr(), points.len() as u32, dim as u32, &cfg, b.device, &mut idx) });
    #[cfg(feature = "indicatif")]
    if let Some(w) = watch { w.finish(); }
    // export into the crate's own fields (zero: n x 64 ids, layers[l]: layer_len[l] x 32 ids)
    let inf = info(idx);
    let mut zero_raw = vec![u32::MAX; points.len() * M * 2];
    let mut upper_raw: Vec<Vec<u32>> = (0..inf.n_upper as usize).map(|l| vec![u32::MAX; inf.layer_len[l] as usize * M]).collect();
    let ptrs: Vec<*mut u32> = upper_raw.iter_mut().map(|v| v.as_mut_ptr()).collect();
    expect(unsafe { idist_index_export(idx, zero_raw.as_mut_ptr(), ptrs.as_ptr()) });
    let zero = zero_raw.chunks_exact(M * 2).map(|row| {
        let mut node = ZeroNode::default();
        for (i, &id) in row.iter().enumerate() { node.set(i, if id == u32::MAX { INVALID } else { PointId(id) }); }
        node
    }).collect();
    let layers = upper_raw.iter().map(|raw| raw.chunks_exact(M).map(UpperNode::from_ids).collect()).collect();   // from_ids: 3-line helper in the patch
    let cache = Cache::default();
    cache.1.store(b.device, Ordering::Relaxed);                       // a later re-import (never needed for this object) would land here too
    let _ = cache.0.set(Some(GpuIndex::new(idx, dim)));
    Some((zero, layers, cache))
}

impl Cache {
    /// `Hnsw::on_device`: where a deserialised index is imported on its first search (no effect once it is on a device)
    pub(crate) fn set_device(&self, device: i32) { self.1.store(device, Ordering::Relaxed); }
    /// The device index of `hnsw`, imported from its own fields on first use (after `serde` deserialisation,
    /// or for an index built by the CPU code on a machine that now has a GPU).
    pub(crate) fn get<P: Point>(&self, ef_search: usize, points: &[P], zero: &[ZeroNode], layers: &[Vec<UpperNode>]) -> Option<&GpuIndex> {
        self.0.get_or_init(|| {
            if points.is_empty() || !available() { return None; }
            let (flat, dim) = flatten(points)?;
            let mut cfg = unsafe { std::mem::zeroed::<IdistConfig>() };
            expect(unsafe { idist_default_config(&mut cfg) });
            cfg.ef_search = ef_search as u32;
            cfg.metric = P::METRIC as i32;
            let zero_raw: Vec<u32> = zero.iter().flat_map(|n| n.iter().map(|p| p.into_inner())).collect();
            let upper_raw: Vec<Vec<u32>> = layers.iter().map(|l| l.iter().flat_map(|n| n.ids().iter().map(|p| p.into_inner())).collect()).collect();
            let ptrs: Vec<*const u32> = upper_raw.iter().map(|v| v.as_ptr()).collect();
            let lens: Vec<u32> = layers.iter().map(|l| l.len() as u32).collect();
            let mut idx = std::ptr::null_mut();
            let device = self.1.load(Ordering::Relaxed);                 // Builder::device / Hnsw::on_device, not "GPU 0"
            expect(unsafe { idist_index_import(flat.as_ptr(), points.len() as u32, dim as u32, &cfg, zero_raw.as_ptr(),
                                               ptrs.as_ptr(), lens.as_ptr(), lens.len() as u32, device, &mut idx) });
            Some(GpuIndex::new(idx, dim))
        }).as_ref()
    }
}

/// The device half of `Search` (`gpu: gpu::Ctx` next to `visited`/`candidates`/`nearest`, lib.rs:560-574): a
/// stream plus visited-set slots, created on first use and re-created when the `Search` moves to another index.
/// The context is bound to the index by its uid, never by address alone (a freed index's address can be reused).
#[derive(Default)]
pub(crate) struct Ctx { ctx: Option<(*mut IdistSearchCtx, u64, u32)>, pid: Vec<u32>, dist: Vec<f32> }   // (context, index uid, slots asked for)
unsafe impl Send for Ctx {}
impl Drop for Ctx { fn drop(&mut self) { if let Some((c, _, _)) = self.ctx.take() { unsafe { idist_search_ctx_free(c) } } } }

impl Ctx {
    fn bind(&mut self, g: &GpuIndex, slots: u32) -> *mut IdistSearchCtx {
        match self.ctx {
            // a one-slot context serves scalar calls only; one made for batches (slots = 0: grows with the batch) serves both
            Some((c, uid, have)) if uid == g.uid && (have == 0 || slots == 1) => c,
            _ => {
                if let Some((c, _, _)) = self.ctx.take() { unsafe { idist_search_ctx_free(c) } }
                let mut c = std::ptr::null_mut();
                // slots = 1: `Search::default()` per thread (rayon map_init) must cost kilobytes, not a chip's worth of
                // visited bitmaps (the engine backs a context's slots lazily, up to what was asked for)
                expect(unsafe { idist_search_ctx_new(g.idx, slots, &mut c) });
                self.ctx = Some((c, g.uid, slots));
                c
            }
        }
    }

    /// `Hnsw::search` (lib.rs:352-383) for a point with an f32 view: fills `nearest` exactly as the CPU loop would
    /// (sorted, nearest first, at most ef_search entries); the caller returns `search.iter().map(map)` as before.
    pub(crate) fn search(&mut self, g: &GpuIndex, q: &[f32], ef: usize, nearest: &mut Vec<Candidate>) {
        assert_eq!(q.len(), g.dim, "query dimension differs from the index");
        let c = self.bind(g, 1);
        self.pid.resize(ef.max(1), 0);
        self.dist.resize(ef.max(1), 0.0);
        let mut cnt = 0u32;
        expect(unsafe { idist_search_batch(g.idx, c, q.as_ptr(), 1, self.pid.as_mut_ptr(), self.dist.as_mut_ptr(), &mut cnt, std::ptr::null_mut()) });
        nearest.clear();
        nearest.extend((0..cnt as usize).map(|i| Candidate { distance: self.dist[i].into(), pid: PointId(self.pid[i]) }));
    }

    /// Additive API (`Hnsw::search_batch`): all queries of a slice in one launch — what the GPU is for
    /// (1M x 300: ~960k queries/s in 10k batches against ~2k/s one call at a time).  Row i of the result holds
    /// the candidates of `queries[i]`, nearest first.
    pub(crate) fn search_batch(&mut self, g: &GpuIndex, queries: &[f32], ef: usize) -> Vec<Vec<Candidate>> {
        let nq = queries.len() / g.dim;
        let c = self.bind(g, 0);
        self.pid.resize(nq * ef.max(1), 0);
        self.dist.resize(nq * ef.max(1), 0.0);
        let mut cnt = vec![0u32; nq];
        expect(unsafe { idist_search_batch(g.idx, c, queries.as_ptr(), nq as u32, self.pid.as_mut_ptr(), self.dist.as_mut_ptr(), cnt.as_mut_ptr(), std::ptr::null_mut()) });
        (0..nq).map(|i| (0..cnt[i] as usize).map(|j| Candidate { distance: self.dist[i * ef + j].into(), pid: PointId(self.pid[i * ef + j]) }).collect()).collect()
    }
}

/// Several GPUs of one node: the index replicated once over xGMI, the queries of every batch block-partitioned over
/// the replicas, no collective in the search itself (`Hnsw` is `Sync`, lib.rs:352-356).
pub struct Replicas { idx: Vec<*mut IdistIndex>, ctx: Vec<*mut IdistSearchCtx>, dim: usize, ef: usize }
unsafe impl Send for Replicas {}
impl Drop for Replicas {
    fn drop(&mut self) {
        for &c in &self.ctx { unsafe { idist_search_ctx_free(c) } }
        for &i in &self.idx { unsafe { idist_index_free(i) } }
    }
}
impl Replicas {
    pub(crate) fn new(root: &GpuIndex, devices: &[i32], ef: usize) -> Self {
        let mut idx = vec![std::ptr::null_mut(); devices.len()];
        // RCCL broadcast over xGMI; a machine without librccl (status 4) gets the peer-copy flavour
        let st = unsafe { idist_replicate_rccl(root.idx, devices.as_ptr(), devices.len() as u32, idx.as_mut_ptr(), std::ptr::null_mut()) };
        if st == 4 { expect(unsafe { idist_replicate(root.idx, devices.as_ptr(), devices.len() as u32, idx.as_mut_ptr()) }); } else { expect(st); }
        let ctx = idx.iter().map(|&i| { let mut c = std::ptr::null_mut(); expect(unsafe { idist_search_ctx_new(i, 0, &mut c) }); c }).collect();
        Self { idx, ctx, dim: root.dim, ef }
    }
    pub fn search_batch(&mut self, queries: &[f32]) -> Vec<Vec<Candidate>> {
        let (nq, ef) = (queries.len() / self.dim, self.ef.max(1));
        let (mut pid, mut dist, mut cnt) = (vec![0u32; nq * ef], vec![0f32; nq * ef], vec![0u32; nq]);
        let idx: Vec<*const IdistIndex> = self.idx.iter().map(|&p| p as *const _).collect();
        expect(unsafe { idist_search_batch_sharded(idx.as_ptr(), self.ctx.as_ptr(), idx.len() as u32, queries.as_ptr(), nq as u32,
                                                   pid.as_mut_ptr(), dist.as_mut_ptr(), cnt.as_mut_ptr(), std::ptr::null_mut()) });
        (0..nq).map(|i| (0..cnt[i] as usize).map(|j| Candidate { distance: dist[i * ef + j].into(), pid: PointId(pid[i * ef + j]) }).collect()).collect()
    }
}

/// `Builder::progress` (lib.rs:70-75; bar driven at :217-221, :306-309, :332-334, :520-526).  No callback crosses the
/// C ABI: the engine publishes {points inserted, layer} to pinned host memory after every build step and a watcher
/// thread moves the caller's bar while the blocking build call runs.
#[cfg(feature = "indicatif")]
mod progress {
    use super::*;
    use std::sync::atomic::{AtomicBool, Ordering};
    use std::sync::Arc;

    pub(crate) struct Watch { p: *mut IdistProgress, stop: Arc<AtomicBool>, thr: std::thread::JoinHandle<()> }
    impl Watch {
        pub(crate) fn start(bar: indicatif::ProgressBar, total: u64) -> Self {
            let mut p = std::ptr::null_mut();
            expect(unsafe { idist_progress_new(&mut p) });
            expect(unsafe { idist_progress_watch_next_build(p) });
            bar.set_length(total);
            bar.set_message("Build index (preparation)");
            let stop = Arc::new(AtomicBool::new(false));
            let (stop2, addr) = (stop.clone(), p as usize);
            let thr = std::thread::spawn(move || {
                let p = addr as *const IdistProgress;
                while !stop2.load(Ordering::Relaxed) {
                    let (mut done, mut total, mut layer) = (0u64, 0u64, -1i32);
                    unsafe { idist_progress_get(p, &mut done, &mut total, &mut layer) };
                    if layer >= 0 { bar.set_message(format!("Building index (layer {})", layer)); }
                    bar.set_position(done);
                    std::thread::sleep(std::time::Duration::from_millis(50));
                }
                bar.finish();
            });
            Self { p, stop, thr }
        }
        pub(crate) fn finish(self) {
            self.stop.store(true, Ordering::Relaxed);
            let _ = self.thr.join();
            unsafe { idist_progress_free(self.p) };
        }
    }
}
