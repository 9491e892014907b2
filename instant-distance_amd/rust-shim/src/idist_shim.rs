//! instant-distance's public API (`Builder`, `Hnsw`, `HnswMap`, `Search`, `Point`, `PointId`,
//! `Item`, `MapItem`) backed by libidist.so — the MI355X HNSW engine.
//!
//! SOURCE ONLY: this image has no Rust toolchain, so this crate has never been compiled here.
//! It is the binding a maintainer adds on the reference side; the same ABI is
//! exercised from C++ (`host/instant_distance.hpp`, tests/host/all.cpp) and Python (ctypes).
//!
//! Signatures follow /root/reference/instant-distance/src/lib.rs; the line each item mirrors is
//! given as `lib.rs:NN`.

use std::ffi::CStr;
use std::os::raw::{c_char, c_void};

use rand::rngs::SmallRng;
use rand::{RngExt, SeedableRng};

// ---- include/idist.h -------------------------------------------------------------------------
#[repr(C)]
#[derive(Clone, Copy)]
pub struct IdistConfig {
    ef_search: u32,
    ef_construction: u32,
    ml: f32,
    has_heuristic: i32,
    extend_candidates: i32,
    keep_pruned: i32,
    metric: i32,
    max_batch: u32,
    tie_policy: i32,   // 0 = strict (IDIST_ERR_TIE_OVERFLOW), 1 = drop: see include/idist.h
    tie_capacity: u32, // 0 = 64
}
#[repr(C)] struct IdistIndex { _p: [u8; 0] }
#[repr(C)] struct IdistSearchCtx { _p: [u8; 0] }
#[repr(C)] struct IdistProgress { _p: [u8; 0] }

extern "C" {
    fn idist_last_error() -> *const c_char;
    fn idist_default_config(cfg: *mut IdistConfig) -> i32;
    fn idist_index_build(points: *const f32, n: u32, dim: u32, cfg: *const IdistConfig, device: i32,
                         out: *mut *mut IdistIndex) -> i32;
    fn idist_index_free(idx: *mut IdistIndex);
    #[cfg(feature = "indicatif")] fn idist_progress_new(out: *mut *mut IdistProgress) -> i32;
    #[cfg(feature = "indicatif")] fn idist_progress_free(p: *mut IdistProgress);
    #[cfg(feature = "indicatif")] fn idist_progress_watch_next_build(p: *mut IdistProgress) -> i32;
    #[cfg(feature = "indicatif")] fn idist_progress_get(p: *const IdistProgress, done: *mut u64, total: *mut u64, layer: *mut i32) -> i32;
    fn idist_search_ctx_new(idx: *const IdistIndex, slots: u32, out: *mut *mut IdistSearchCtx) -> i32;
    fn idist_search_ctx_free(ctx: *mut IdistSearchCtx);
    fn idist_search_batch(idx: *const IdistIndex, ctx: *mut IdistSearchCtx, queries: *const f32, nq: u32,
                          out_pid: *mut u32, out_dist: *mut f32, out_count: *mut u32,
                          out_counters: *mut u32) -> i32;
}

fn expect(status: i32) {
    // the reference API is infallible (it only panics at lib.rs:256 and :148)
    if status != 0 {
        let msg = unsafe { CStr::from_ptr(idist_last_error()) }.to_string_lossy().into_owned();
        panic!("libidist: status {status}: {msg}");
    }
}

pub const METRIC_L2SQ: i32 = 0; // FloatArray, instant-distance-py/src/lib.rs:378-421
pub const METRIC_L2: i32 = 1;   // tests/all.rs:93-97, examples/colors.rs:21-25

/// lib.rs:780-782.  `distance` stays for CPU-side uses (e.g. brute-force checks); the GPU engine
/// needs the coordinates as f32 and one of the two distances the reference ships.  Additive:
/// existing `impl Point` blocks only have to add `as_f32` (and `METRIC` if they use sqrt).
pub trait Point: Clone + Sync {
    const METRIC: i32 = METRIC_L2SQ;
    fn distance(&self, other: &Self) -> f32;
    fn as_f32(&self, out: &mut Vec<f32>);
}

/// core/types.rs:241-253
#[derive(Clone, Copy, Debug, Eq, Hash, Ord, PartialEq, PartialOrd)]
pub struct PointId(pub(crate) u32);
impl PointId {
    pub fn is_valid(self) -> bool { self.0 != u32::MAX }
    pub fn into_inner(self) -> u32 { self.0 }
}
impl From<u32> for PointId { fn from(id: u32) -> Self { PointId(id) } }

/// lib.rs:115-128
#[derive(Copy, Clone, Debug)]
pub struct Heuristic { pub extend_candidates: bool, pub keep_pruned: bool }
impl Default for Heuristic { fn default() -> Self { Heuristic { extend_candidates: false, keep_pruned: true } } }

/// lib.rs:23-113
#[derive(Clone)]
pub struct Builder {
    cfg: IdistConfig, seed: u64, device: i32,
    #[cfg(feature = "indicatif")] progress: Option<indicatif::ProgressBar>,   // lib.rs:29-30
}
impl Default for Builder {
    fn default() -> Self {
        let mut cfg = unsafe { std::mem::zeroed::<IdistConfig>() };
        expect(unsafe { idist_default_config(&mut cfg) });
        Self { cfg, seed: rand::random(), device: 0, #[cfg(feature = "indicatif")] progress: None }
    }
}
impl Builder {
    pub fn ef_construction(mut self, ef: usize) -> Self { self.cfg.ef_construction = ef as u32; self }   // :35-38
    pub fn ef_search(mut self, ef: usize) -> Self { self.cfg.ef_search = ef as u32; self }                 // :44-47
    pub fn select_heuristic(mut self, params: Option<Heuristic>) -> Self {                                  // :49-52
        self.cfg.has_heuristic = params.is_some() as i32;
        if let Some(h) = params { self.cfg.extend_candidates = h.extend_candidates as i32; self.cfg.keep_pruned = h.keep_pruned as i32; }
        self
    }
    pub fn ml(mut self, ml: f32) -> Self { self.cfg.ml = ml; self }                                         // :57-60
    pub fn seed(mut self, seed: u64) -> Self { self.seed = seed; self }                                     // :65-68
    /// A `ProgressBar` to track `Hnsw` construction progress, lib.rs:70-75
    #[cfg(feature = "indicatif")]
    pub fn progress(mut self, bar: indicatif::ProgressBar) -> Self { self.progress = Some(bar); self }
    pub fn build<P: Point, V: Clone>(self, points: Vec<P>, values: Vec<V>) -> HnswMap<P, V> { HnswMap::new(points, values, self) } // :78-80
    pub fn build_hnsw<P: Point>(self, points: Vec<P>) -> (Hnsw<P>, Vec<PointId>) { Hnsw::new(points, self) } // :83-85
    #[doc(hidden)]
    pub fn into_parts(self) -> (usize, usize, f32, u64) { (self.cfg.ef_search as usize, self.cfg.ef_construction as usize, self.cfg.ml, self.seed) }
}

/// lib.rs:194-199 — owns the permuted points (Item borrows from it) and the device index.
pub struct Hnsw<P> { idx: *mut IdistIndex, points: Vec<P>, ef_search: usize, dim: usize }
unsafe impl<P: Sync> Sync for Hnsw<P> {}   // the index is immutable after build
unsafe impl<P: Send> Send for Hnsw<P> {}
impl<P> Drop for Hnsw<P> { fn drop(&mut self) { unsafe { idist_index_free(self.idx) } } }

impl<P: Point> Hnsw<P> {
    pub fn builder() -> Builder { Builder::default() }

    fn new(points: Vec<P>, builder: Builder) -> (Self, Vec<PointId>) {                                      // :209-345
        // the shuffle stays in Rust with the real `rand` crate, :214, :257-270
        let mut rng = SmallRng::seed_from_u64(builder.seed);
        assert!(points.len() < u32::MAX as usize);                                                           // :256
        let mut shuffled = (0..points.len())
            .map(|i| (rng.random_range(0..points.len() as u32), i))
            .collect::<Vec<_>>();
        shuffled.sort_unstable();
        let mut out = vec![PointId(u32::MAX); points.len()];
        let points = shuffled.into_iter().enumerate()
            .map(|(i, (_, idx))| { out[idx] = PointId(i as u32); points[idx].clone() })
            .collect::<Vec<_>>();
        // flatten in PointId order and hand over to the GPU engine (layers, inserts, heuristic: :238-345)
        let mut flat = Vec::new();
        let mut dim = 1;
        for (i, p) in points.iter().enumerate() { p.as_f32(&mut flat); if i == 0 { dim = flat.len().max(1); } }
        let mut cfg = builder.cfg;
        cfg.metric = P::METRIC;
        let mut idx = std::ptr::null_mut();
        // lib.rs:217-221, :306-309, :332-334: the bar is driven from a watcher thread that polls the engine's
        // progress object while the (blocking) build call runs on this thread
        #[cfg(feature = "indicatif")]
        let watch = builder.progress.clone().map(|bar| {
            let mut p = std::ptr::null_mut();
            expect(unsafe { idist_progress_new(&mut p) });
            expect(unsafe { idist_progress_watch_next_build(p) });
            bar.set_length(points.len() as u64);
            bar.set_message("Build index (preparation)");
            let stop = std::sync::Arc::new(std::sync::atomic::AtomicBool::new(false));
            let (stop2, addr) = (stop.clone(), p as usize);
            let thr = std::thread::spawn(move || {
                let p = addr as *const IdistProgress;
                while !stop2.load(std::sync::atomic::Ordering::Relaxed) {
                    let (mut done, mut total, mut layer) = (0u64, 0u64, -1i32);
                    unsafe { idist_progress_get(p, &mut done, &mut total, &mut layer) };
                    if layer >= 0 { bar.set_message(format!("Building index (layer {})", layer)); }
                    bar.set_position(done);
                    std::thread::sleep(std::time::Duration::from_millis(50));
                }
                bar.finish();
            });
            (p, stop, thr)
        });
        expect(unsafe { idist_index_build(flat.as_ptr(), points.len() as u32, dim as u32, &cfg, builder.device, &mut idx) });
        #[cfg(feature = "indicatif")]
        if let Some((p, stop, thr)) = watch {
            stop.store(true, std::sync::atomic::Ordering::Relaxed);
            let _ = thr.join();
            unsafe { idist_progress_free(p) };
        }
        (Self { idx, points, ef_search: cfg.ef_search as usize, dim }, out)
    }

    /// lib.rs:352-383
    pub fn search<'a, 'b: 'a>(&'b self, point: &P, search: &'a mut Search) -> impl ExactSizeIterator<Item = Item<'b, P>> + 'a {
        let mut q = Vec::with_capacity(self.dim);
        point.as_f32(&mut q);
        search.run(self.idx, &q, self.ef_search);
        search.nearest.iter().map(move |&(distance, pid)| Item { distance, pid, point: &self.points[pid.0 as usize] })
    }
    pub fn iter(&self) -> impl Iterator<Item = (PointId, &P)> { self.points.iter().enumerate().map(|(i, p)| (PointId(i as u32), p)) }
    #[doc(hidden)]
    pub fn get(&self, i: usize, search: &Search) -> Option<Item<'_, P>> {
        let &(distance, pid) = search.nearest.get(i)?;
        Some(Item { distance, pid, point: &self.points[pid.0 as usize] })
    }
}
impl<P> std::ops::Index<PointId> for Hnsw<P> { type Output = P; fn index(&self, i: PointId) -> &P { &self.points[i.0 as usize] } }

/// lib.rs:399-403
pub struct Item<'a, P> { pub distance: f32, pub pid: PointId, pub point: &'a P }
/// lib.rs:175-180
pub struct MapItem<'a, P, V> { pub distance: f32, pub pid: PointId, pub point: &'a P, pub value: &'a V }

/// lib.rs:131-173
pub struct HnswMap<P, V> { hnsw: Hnsw<P>, pub values: Vec<V> }
impl<P: Point, V: Clone> HnswMap<P, V> {
    fn new(points: Vec<P>, values: Vec<V>, builder: Builder) -> Self {
        let (hnsw, ids) = Hnsw::new(points, builder);
        let mut sorted = ids.into_iter().enumerate().collect::<Vec<_>>();
        sorted.sort_unstable_by_key(|id| id.1);
        let new = sorted.into_iter().map(|(src, _)| values[src].clone()).collect();                           // :144-149
        Self { hnsw, values: new }
    }
    pub fn search<'a>(&'a self, point: &P, search: &'a mut Search) -> impl ExactSizeIterator<Item = MapItem<'a, P, V>> + 'a {
        self.hnsw.search(point, search).map(move |item| MapItem { distance: item.distance, pid: item.pid, point: item.point, value: &self.values[item.pid.0 as usize] })
    }
    pub fn iter(&self) -> impl Iterator<Item = (PointId, &P)> { self.hnsw.iter() }
}

/// lib.rs:560-574 — reusable scratch: owns a device-side search context (stream + visited slots).
pub struct Search { ctx: *mut IdistSearchCtx, owner: *const IdistIndex, nearest: Vec<(f32, PointId)>, pid: Vec<u32>, dist: Vec<f32> }
impl Default for Search {
    fn default() -> Self { Self { ctx: std::ptr::null_mut(), owner: std::ptr::null(), nearest: Vec::new(), pid: Vec::new(), dist: Vec::new() } }
}
impl Drop for Search { fn drop(&mut self) { if !self.ctx.is_null() { unsafe { idist_search_ctx_free(self.ctx) } } } }
impl Search {
    fn run(&mut self, idx: *const IdistIndex, q: &[f32], ef: usize) {
        if self.owner != idx {
            if !self.ctx.is_null() { unsafe { idist_search_ctx_free(self.ctx) } }
            expect(unsafe { idist_search_ctx_new(idx, 0, &mut self.ctx) });
            self.owner = idx;
        }
        self.pid.resize(ef.max(1), 0);
        self.dist.resize(ef.max(1), 0.0);
        let mut cnt = 0u32;
        expect(unsafe { idist_search_batch(idx, self.ctx, q.as_ptr(), 1, self.pid.as_mut_ptr(), self.dist.as_mut_ptr(), &mut cnt, std::ptr::null_mut()) });
        self.nearest.clear();
        for i in 0..cnt as usize { self.nearest.push((self.dist[i], PointId(self.pid[i]))); }
    }
}

#[allow(dead_code)]
fn _unused(_: *mut c_void) {}
