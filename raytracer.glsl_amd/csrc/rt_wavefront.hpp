// rt_wavefront.hpp -- bounce-wavefront pipeline: one launch per bounce over a compacted queue of
// live rays.
//
//   generate_rays_kernel   per pixel: seed PCG4D, camera ray (+DOF)            -> queue[0]
//   bounce_kernel<R,MODE>  per live ray: sphere scan, brute-force triangle pass, closest hit,
//                          material response / miss / termination               -> queue[b+1]
//                          live rays are compacted with wave ballot + prefix popcount and ONE
//                          atomicAdd per wave, so every lane of every wave of the next bounce is busy
//   resolve_kernel         only for u_samples > 1: running mean of the per-pixel sums
//
// Brute force makes ray coherence irrelevant for the triangle pass (every ray tests every triangle),
// so compaction costs nothing there; what it buys is full 64-lane occupancy on every bounce.
//
// Triangle pass variants (MODE): all read the same precomputed TriEdges records.
//   kScalar  records are wave-uniform -> fetched with scalar loads, fma operands come from SGPRs
//   kLds     256-thread work-group stages tiles of kTile records HBM -> LDS with coalesced 16-byte
//            loads (double buffered, one barrier per tile); lanes read them back as broadcasts
// R = rays per lane: each fetched record is used for R rays, dividing the operand traffic per test.
#pragma once
#include "rt_device.hpp"

#pragma clang fp contract(off)

namespace rt {

enum { kScalar = 0, kLds = 1 };
constexpr int kTile = 256;        // triangles per LDS tile (256 * 80 B = 20 KiB per buffer)
constexpr int kBoundGroup = 64;   // triangles per conservative-bound group

// Structure-of-arrays ray queue, 68 B per ray, every stream read and written coalesced.
struct RayQueue {
    float4 *a;        // o.xyz, d.x
    float4 *b;        // d.yz, thr.xy
    float4 *c;        // thr.z, radiance.xyz
    uint4 *rng;       // PCG4D state
    uint32_t *pixel;  // local pixel index (local_row * width + x)
};

struct WaveBuffers {
    RayQueue q[2];
    uint32_t *counts;     // counts[b] = rays entering bounce b (b = 0 .. max_bounce)
    float4 *sums;         // per local pixel: sum of finished samples (only u_samples > 1)
    float4 *cam_a, *cam_b;// per local pixel camera ray (only u_samples > 1)
    uint4 *pix_rng;       // per local pixel RNG state between samples (only u_samples > 1)
    const float2 *group_bounds;   // per kBoundGroup triangles: (max bound_e, max bound_m)
    unsigned long long *best[2];  // split pipeline: per queue slot, packed (t bits << 32 | visit index) of the nearest mesh hit
    uint2 *cand;                  // kernel 4: (queue slot, storage position) pairs that survived the broad phase of the current bounce,
                                  // one region of `cand_region` pairs per wave of the scan launch
    uint32_t *cand_counts;        // kernel 4: pairs stored in each region by the scan launch of the current bounce
    uint32_t cand_region;         // capacity of one region (pairs); what does not fit is tested in place by the scan
    uint32_t *keep;               // kernel 4 packet culling: per granule of 128 rays of the queue being scanned, one bit per tile of 10 triangles (rt_scan.hpp, packet_cull_kernel)
    uint32_t keep_words;          // 32-bit words per granule = ceil(tiles / 32)
    uint32_t *items;              // kernel 4, culled dynamic launches: per chunk of the scan launch the granules that have anything to scan
    uint32_t *item_counts;        //   ... their number per chunk; items_stride entries are reserved per chunk (cull_items_kernel)
    uint32_t items_stride;
    uint32_t hybrid_div;          // kernel 4, hybrid work distribution: every hybrid_div-th granule is claimed, the others are taken in turns
    uint32_t *plan_prefix;        // kernel 4, planned work distribution of a culled launch: per chunk the inclusive prefix sum of the items' costs over
    uint32_t plan_stride;         //   the granules (plan_stride entries per chunk), the chunks' totals and where each chunk starts on the cost line
    uint32_t *plan_total;         //   (scan_plan_kernel, scan_plan_base_kernel; rt_scan.hpp)
    unsigned long long *plan_base;
    uint32_t *sched;              // kernel 4: per scan launch (bounce) and chunk the next unclaimed item; zeroed with the ray counts at frame start
    uint32_t sched_stride;        //   entries per bounce
    uint32_t *cand_peak;          // max over the frame's scan waves of the pairs a wave wanted to append (host: sizes the regions)
    // ray binning (kernel 4, option "cull" = 3): shade_kernel leaves the rays of the next bounce in the staging queue `qt` with a bin key
    // each, sort_scatter_kernel moves them into the next queue in key order, so that the 128 rays of a granule are neighbours in origin
    // AND direction and packet culling has something to certify on bounces >= 1.  Queue order never shows in a result: the state of
    // a path travels with its ray, hits merge by visit index, every pixel has one path per frame.
    RayQueue qt;                  // staging queue (a third queue)
    uint2 *sort_kr;               // per staging slot: (bin key, rank inside the bin)
    uint32_t *sort_hist;          // rays per bin -> first slot of the bin (sort_prefix_kernel); 2^sort_bits entries + one block sum per 4096
    uint32_t *sort_hist_other;    // the counters of the NEXT binned bounce: sort_scatter_kernel leaves them zero (two sets take turns; no fill launch per bounce)
    uint32_t sort_bits;           // key = direction bin (8 bits: 16 x 16 octahedral cells in Morton order) << 3 sort_ob | origin word (3 sort_ob bits)
    uint32_t sort_ob;
    uint32_t sort_db, sort_T;     // direction cells per axis = 2^sort_db (4); origin cell bits behind the flag
    // origin word = [outside flag][3 sort_ob - 1 cell bits].  Inside the mesh's box (+ one cell): cells of the box, the bits dealt to the
    // axes by extent (a flat mesh spends none on its thin axis) and interleaved longest-cell-first.  Outside (the ground, the spheres: at
    // C2 58 % of the rays entering bounce 2): 16 cells per axis whose size doubles with the distance from the box's centre -- clamped
    // into the box's border cells those rays used to blow up the bounds of every granule there.
    float sort_lo[3], sort_inv_cell[3];        // inside: cell_a = (o_a - lo_a) * inv_cell_a, clamped
    float sort_in_lo[3], sort_in_hi[3];        // the inside test
    float sort_cen[3], sort_inv_unit;          // outside: j = floor(log2(1 + |o_a - cen_a| * inv_unit)), clamped to 7; cell = 8 + j or 7 - j
    uint32_t sort_in_bits, sort_out_bits;      // bits per axis, 4 bits each (x | y << 4 | z << 8)
    uint32_t sort_in_order, sort_out_order;    // the axis of every key bit from the top, 2 bits each
    float4 *batch_rad;            // frame batching (option "frame_batch"): the paths of B consecutive frames travel through ONE set of launches;
    uint32_t batch_px;            //   a finished path leaves its radiance in batch_rad[frame slot * batch_px + pixel] (the slot rides in the top
                                  //   four bits of the path's pixel word), resolve_batch_kernel folds the slots into the image in frame order
};
constexpr uint32_t kBatchMax = 16u;           // frames per batch (the slot has four bits)
constexpr uint32_t kBatchPixelMask = 0x0FFFFFFFu;
struct BatchInfo { uint32_t n; int32_t frames[kBatchMax]; int32_t reset[kBatchMax]; };

struct PathState { f3 o, d, thr, rad; Rng rng; uint32_t pixel; };

// Stores of data the NEXT kernels read (ray queues, hit keys): written through to memory (system scope) instead of left as dirty lines in
// the L2 of whichever XCD the wave ran on.  MI355X has eight L2s; a line written by one kernel from XCD A and, a frame later, by the same
// kernel from XCD B can be written back out of order if A's dirty copy is still around -- memory then falls back to the old value.  With
// one pipeline on the device the block -> XCD mapping repeats from frame to frame and this does not happen; with several pipelines
// dispatching at once it did: 16 queue slots (256 bytes) of a frame reverting to the previous frame's rays after ray generation had
// verifiably written the new ones (tools/diagnostics/flaky_tiled.py: 49 and 22 wrong images of 600 with ordinary stores,
// 1 and 3 -- of another kind -- with these, same box, alternating).
// (inline asm: the compiler's hazard recogniser does not see these stores, so the wait states a store of more than 8 bytes needs before a
// vector instruction may overwrite its data registers are part of the statement)
typedef float rt_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t rt_u4v __attribute__((ext_vector_type(4)));
typedef uint32_t rt_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_through(float4 *p, float x, float y, float z, float w) { const rt_f4v v = {x, y, z, w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(uint4 *p, uint32_t x, uint32_t y, uint32_t z, uint32_t w) { const rt_u4v v = {x, y, z, w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(uint32_t *p, uint32_t x) { asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(x) : "memory"); }
__device__ __forceinline__ void store_through(unsigned long long *p, unsigned long long x) { const rt_u2v v = {(uint32_t)x, (uint32_t)(x >> 32)}; asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

// (camera = the queue of bounce 0: every ray there has throughput 1 and radiance 0, so the `c` stream -- a quarter of the queue's bytes --
// is neither written by generate_rays_kernel nor read by the kernels of bounce 0)
__device__ __forceinline__ void store_ray(const RayQueue &q, uint32_t slot, const PathState &s, bool camera = false)
{
    store_through(q.a + slot, s.o.x, s.o.y, s.o.z, s.d.x);
    store_through(q.b + slot, s.d.y, s.d.z, s.thr.x, s.thr.y);
    if (!camera) store_through(q.c + slot, s.thr.z, s.rad.x, s.rad.y, s.rad.z);
    store_through(q.rng + slot, s.rng.x, s.rng.y, s.rng.z, s.rng.w);
    store_through(q.pixel + slot, s.pixel);
}
__device__ __forceinline__ PathState load_ray(const RayQueue &q, uint32_t slot, bool camera = false)
{
    PathState s;
    float4 a = q.a[slot], b = q.b[slot], c = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    if (!camera) c = q.c[slot];
    uint4 g = q.rng[slot];
    s.o = mk(a.x, a.y, a.z); s.d = mk(a.w, b.x, b.y); s.thr = mk(b.z, b.w, c.x); s.rad = mk(c.y, c.z, c.w);
    s.rng.x = g.x; s.rng.y = g.y; s.rng.z = g.z; s.rng.w = g.w;
    s.pixel = q.pixel[slot];
    return s;
}

// A finished path: fold its radiance into the image.  With u_samples == 1 (the only value the
// reference ever uses, src/renderer.h:168) this is the running mean of main() (:561-568) directly;
// otherwise the sample is added to the pixel's sum and the RNG state is parked for the next sample.
__device__ __forceinline__ void finish_path(const FrameParams &P, const ImageView &im, const WaveBuffers &wb,
                                            const PathState &s, uint4 *rng_out)
{
    if (wb.batch_rad) {                  // batched frames: the image is updated by resolve_batch_kernel, in frame order
        store_through(wb.batch_rad + ((size_t)(s.pixel >> 28) * wb.batch_px + (s.pixel & kBatchPixelMask)), s.rad.x, s.rad.y, s.rad.z, 0.0f);
        return;
    }
    if (P.samples == 1u) {
        float4 *pix = im.pixels + s.pixel;
        f3 prev = mk(0.0f, 0.0f, 0.0f);
        if (!P.reset_flag) { float4 q = *pix; prev = mk(q.x, q.y, q.z); }
        { const float4 v = accumulate_pixel(P, s.rad, prev); store_through(pix, v.x, v.y, v.z, v.w); }
    } else {
        float4 acc = wb.sums[s.pixel];
        store_through(wb.sums + s.pixel, acc.x + s.rad.x, acc.y + s.rad.y, acc.z + s.rad.z, 0.0f);
        store_through(wb.pix_rng + s.pixel, s.rng.x, s.rng.y, s.rng.z, s.rng.w);
    }
    if (rng_out) rng_out[s.pixel] = make_uint4(s.rng.x, s.rng.y, s.rng.z, s.rng.w);
}

// ---- queue 0 ---------------------------------------------------------------------------------------
// sample 0: camera rays from scratch.  sample > 0: same camera ray, RNG continued (:556-559).
// Frame batching: the launch of frame f of a batch writes slots [slot_off, slot_off + n0) with pixel_tag = f << 28; only the first stores
// the batch's ray count (count0 = rays of all its frames; 0: leave the count alone).  A single frame: slot_off = pixel_tag = 0, count0 = n0.
__global__ void __launch_bounds__(256) generate_rays_kernel(FrameParams P, ImageView im, WaveBuffers wb, uint32_t sample, uint32_t n0, Counters *counters, uint32_t n_counts,
                                                            uint32_t slot_off, uint32_t pixel_tag, uint32_t count0, uint32_t n_sched)
{
    // the ray counts of the bounces (and the word behind them) start at zero: cleared here rather than by a memset launch of their own
    // (n_counts = 0: the host has done it); nothing touches them before the first shade kernel
    if (blockIdx.x == 0 && threadIdx.x < n_counts && threadIdx.x > 0u) store_through(wb.counts + threadIdx.x, 0u);
    // queue order = 8x8 pixel blocks, row-major over blocks: neighbouring lanes start as neighbouring pixels
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx < n_sched) store_through(wb.sched + idx, 0u);      // the scan launches' work counters of this frame (n_sched = 0: the host has cleared them)
    if (idx >= n0) return;
    if (idx == 0u) {
        if (count0) store_through(wb.counts, count0);
        if (counters) atomicAdd(&counters->paths, (unsigned long long)n0);
    }
    const uint32_t blocks_x = im.disp_w >> 3;
    const uint32_t blk = idx >> 6, in = idx & 63u;
    const uint32_t bx = blk % blocks_x, by = blk / blocks_x;
    const int px = (int)(bx * 8u + (in & 7u)), lrow = (int)(by * 8u + (in >> 3));
    if (lrow >= im.local_rows) return;
    const int py = local_to_global_row(im, lrow);
    if (py >= im.disp_h) return;      // rows past the dispatch footprint come last in every strip order
    PathState s;
    s.pixel = (uint32_t)lrow * (uint32_t)im.width + (uint32_t)px;
    if (sample == 0u) {
        s.rng.x = (uint32_t)px; s.rng.y = (uint32_t)py; s.rng.z = (uint32_t)P.random;
        s.rng.w = (uint32_t)px + (uint32_t)py + (uint32_t)P.random;
        camera_ray(P, px, py, im.width, im.height, s.rng, s.o, s.d);
        if (P.samples > 1u) {
            store_through(wb.cam_a + s.pixel, s.o.x, s.o.y, s.o.z, s.d.x);
            store_through(wb.cam_b + s.pixel, s.d.y, s.d.z, 0.0f, 0.0f);
            store_through(wb.sums + s.pixel, 0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else {
        float4 a = wb.cam_a[s.pixel], b = wb.cam_b[s.pixel];
        uint4 g = wb.pix_rng[s.pixel];
        s.o = mk(a.x, a.y, a.z); s.d = mk(a.w, b.x, b.y);
        s.rng.x = g.x; s.rng.y = g.y; s.rng.z = g.z; s.rng.w = g.w;
    }
    s.thr = mk(1.0f, 1.0f, 1.0f); s.rad = mk(0.0f, 0.0f, 0.0f);
    s.pixel |= pixel_tag;
    store_ray(wb.q[0], slot_off + idx, s, true);
    if (wb.best[0]) store_through(wb.best[0] + (slot_off + idx), 0xFFFFFFFFFFFFFFFFull);
}

// Frame batching: the running mean of main() (:561-568) applied once per frame of the batch, oldest first, from the radiance each
// frame's path left in its slot -- the same operations in the same order as B separate frames.
__global__ void __launch_bounds__(256) resolve_batch_kernel(ImageView im, WaveBuffers wb, BatchInfo bi)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t blocks_x = im.disp_w >> 3;
    const uint32_t blk = idx >> 6, in = idx & 63u;
    const int px = (int)((blk % blocks_x) * 8u + (in & 7u)), lrow = (int)((blk / blocks_x) * 8u + (in >> 3));
    if (lrow >= im.local_rows || local_to_global_row(im, lrow) >= im.disp_h) return;
    const uint32_t pixel = (uint32_t)lrow * (uint32_t)im.width + (uint32_t)px;
    float4 *pix = im.pixels + pixel;
    float4 cur = *pix;
    FrameParams P{};
    P.samples = 1u;
    for (uint32_t f = 0; f < bi.n; ++f) {
        const float4 r = wb.batch_rad[(size_t)f * wb.batch_px + pixel];
        P.frames = bi.frames[f];
        const f3 prev = bi.reset[f] ? mk(0.0f, 0.0f, 0.0f) : mk(cur.x, cur.y, cur.z);
        cur = accumulate_pixel(P, mk(r.x, r.y, r.z), prev);
    }
    store_through(pix, cur.x, cur.y, cur.z, cur.w);
}

__global__ void __launch_bounds__(256) resolve_kernel(FrameParams P, ImageView im, WaveBuffers wb)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t blocks_x = im.disp_w >> 3;
    const uint32_t blk = idx >> 6, in = idx & 63u;
    const int px = (int)((blk % blocks_x) * 8u + (in & 7u)), lrow = (int)((blk / blocks_x) * 8u + (in >> 3));
    if (lrow >= im.local_rows || local_to_global_row(im, lrow) >= im.disp_h) return;
    const uint32_t pixel = (uint32_t)lrow * (uint32_t)im.width + (uint32_t)px;
    float4 *pix = im.pixels + pixel;
    f3 prev = mk(0.0f, 0.0f, 0.0f);
    if (!P.reset_flag) { float4 q = *pix; prev = mk(q.x, q.y, q.z); }
    float4 sum = wb.sums[pixel];
    { const float4 v = accumulate_pixel(P, mk(sum.x, sum.y, sum.z), prev); store_through(pix, v.x, v.y, v.z, v.w); }
}

// ---- one bounce --------------------------------------------------------------------------------------
// Register discipline: during the triangle pass a ray is only (cv, d, margin scales, threshold,
// best t, best index) = 11 VGPRs; origin, throughput, radiance, RNG state and pixel stay in the
// queue in HBM and are (re)loaded after the pass.  The exact re-test of a candidate (rare) fetches
// what it needs from memory instead of keeping it live.
template <int R>
struct HotRays {
    f3 cv[R], d[R];        // cross(d, o), d
    float ncv[R], nd[R];   // margin scales (-inf marks an empty slot: its threshold becomes +inf)
    float thresh[R];       // -margin for the current bound group
    float best_t[R];
    uint32_t best_v[R];
};

template <int R>
__device__ __forceinline__ void set_thresh(HotRays<R> &hr, float2 gb)
{
#pragma unroll
    for (int r = 0; r < R; ++r) hr.thresh[r] = -(__builtin_fmaf(gb.x, hr.ncv[r], gb.y * hr.nd[r]) + 1e-30f);
}

// the 18 edge-function coefficients of one triangle as they sit in a TriEdges record
struct TriCoef { float4 q0, q1, q2, q3; float2 q4; };
// q0 = e0x e0y e0z e1x | q1 = e1y e1z e2x e2y | q2 = e2z m0x m0y m0z | q3 = m1x m1y m1z m2x | q4 = m2y m2z

__device__ __forceinline__ float filter_min(const TriCoef &T, f3 cv, f3 d)
{
    float f0 = T.q0.x * cv.x;
    float f1 = T.q0.w * cv.x;
    float f2 = T.q1.z * cv.x;
    f0 = __builtin_fmaf(T.q0.y, cv.y, f0); f1 = __builtin_fmaf(T.q1.x, cv.y, f1); f2 = __builtin_fmaf(T.q1.w, cv.y, f2);
    f0 = __builtin_fmaf(T.q0.z, cv.z, f0); f1 = __builtin_fmaf(T.q1.y, cv.z, f1); f2 = __builtin_fmaf(T.q2.x, cv.z, f2);
    f0 = __builtin_fmaf(T.q2.y, d.x, f0);  f1 = __builtin_fmaf(T.q3.x, d.x, f1);  f2 = __builtin_fmaf(T.q3.w, d.x, f2);
    f0 = __builtin_fmaf(T.q2.z, d.y, f0);  f1 = __builtin_fmaf(T.q3.y, d.y, f1);  f2 = __builtin_fmaf(T.q4.x, d.y, f2);
    f0 = __builtin_fmaf(T.q2.w, d.z, f0);  f1 = __builtin_fmaf(T.q3.z, d.z, f1);  f2 = __builtin_fmaf(T.q4.y, d.z, f2);
    return __builtin_fminf(__builtin_fminf(f0, f1), f2);
}

// exact re-test of a filter survivor, everything fetched from memory (wave-uniform v: scalar loads)
__device__ __noinline__ float candidate_t(const SceneView &sc, uint32_t v, f3 o, f3 d)
{
    TriRay tr; tr.o = o; tr.d = d; tr.cv = cross3(d, o); tr.ncv = tr.nd = 0.0f;
    return tri_exact(sc.tri_edges[v], sc.tri_planes[v], tr);
}

template <int R, bool kCount>
__device__ __forceinline__ void test_triangle(const TriCoef &T, const SceneView &sc, const RayQueue &qin, uint32_t slot0,
                                              uint32_t v, HotRays<R> &hr, unsigned long long &c_cand)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float mn = filter_min(T, hr.cv[r], hr.d[r]);
        if (!(mn <= hr.thresh[r])) {                         // rare: a handful of triangles per ray
            if (kCount) c_cand++;
            float4 a = qin.a[slot0 + r * 256u];
            float t = candidate_t(sc, v, mk(a.x, a.y, a.z), hr.d[r]);
            if (kEps < t && t < hr.best_t[r]) { hr.best_t[r] = t; hr.best_v[r] = v; }
        }
    }
}

__device__ __forceinline__ TriCoef load_coef(const float4 *p)
{
    TriCoef T;
    T.q0 = p[0]; T.q1 = p[1]; T.q2 = p[2]; T.q3 = p[3];
    T.q4 = *reinterpret_cast<const float2 *>(p + 4);
    return T;
}

template <int R, int MODE, bool kCount>
__global__ void __launch_bounds__(256) bounce_kernel(SceneView sc, FrameParams P, ImageView im, WaveBuffers wb,
                                                     uint32_t bounce, uint4 *rng_out, Counters *counters)
{
    __shared__ float4 lds_tile[MODE == kLds ? 2 * kTile * 5 : 1];
    const uint32_t n_rays = wb.counts[bounce];
    const uint32_t base = blockIdx.x * (256u * R);
    if (base >= n_rays) return;                               // uniform per work-group
    const RayQueue qin = (bounce & 1u) ? wb.q[1] : wb.q[0];
    const RayQueue qout = (bounce & 1u) ? wb.q[0] : wb.q[1];
    const bool last_bounce = (bounce + 1u >= P.max_bounce);
    const uint32_t slot0 = base + threadIdx.x;                // ray r of this lane lives in slot0 + r*256

    HotRays<R> hr;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t slot = slot0 + r * 256u;
        const bool valid = slot < n_rays;
        f3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 0.0f);
        if (valid) { float4 a = qin.a[slot], b = qin.b[slot]; o = mk(a.x, a.y, a.z); d = mk(a.w, b.x, b.y); }
        TriRay tr = make_tri_ray(o, d);
        hr.cv[r] = tr.cv; hr.d[r] = d;
        hr.ncv[r] = valid ? tr.ncv : -__builtin_inff();
        hr.nd[r] = valid ? tr.nd : -__builtin_inff();
        hr.thresh[r] = __builtin_inff();
        hr.best_t[r] = kInf; hr.best_v[r] = 0xFFFFFFFFu;
    }

    unsigned long long c_cand = 0, c_env = 0;
    const uint32_t n_tri = sc.n_tri_visits;
    // ---- find_closest_mesh (:331-361): every ray against every triangle, in reference order
    if (MODE == kScalar) {
        const float4 *src = reinterpret_cast<const float4 *>(sc.tri_edges);
        for (uint32_t v0 = 0; v0 < n_tri; v0 += kBoundGroup) {
            set_thresh<R>(hr, wb.group_bounds[v0 / kBoundGroup]);
            const uint32_t v1 = min(v0 + (uint32_t)kBoundGroup, n_tri);
#pragma unroll 2
            for (uint32_t v = v0; v < v1; ++v)
                test_triangle<R, kCount>(load_coef(src + (size_t)v * 5), sc, qin, slot0, v, hr, c_cand);
        }
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(sc.tri_edges);
        const uint32_t n_tiles = (n_tri + kTile - 1) / kTile;
        const uint32_t total_f4 = n_tri * 5u;
        float4 stage[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            uint32_t i = k * 256u + threadIdx.x;
            if (i < total_f4) lds_tile[i] = src[i];
        }
        __syncthreads();
        for (uint32_t t = 0; t < n_tiles; ++t) {
            const uint32_t nxt = (t + 1u) * (kTile * 5u);
            const bool have_next = (t + 1u < n_tiles);
            if (have_next) {
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    uint32_t i = nxt + k * 256u + threadIdx.x;
                    stage[k] = (i < total_f4) ? src[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            const float4 *buf = lds_tile + (t & 1u) * (kTile * 5);
            const uint32_t vbase = t * kTile;
            const uint32_t cnt = min((uint32_t)kTile, n_tri - vbase);
            for (uint32_t j0 = 0; j0 < cnt; j0 += kBoundGroup) {
                set_thresh<R>(hr, wb.group_bounds[(vbase + j0) / kBoundGroup]);
                const uint32_t j1 = min(j0 + (uint32_t)kBoundGroup, cnt);
                TriCoef cur = load_coef(buf + j0 * 5);
#pragma unroll 2
                for (uint32_t j = j0; j < j1; ++j) {
                    // prefetch the next record (the slot after the tile's last record is readable LDS)
                    TriCoef nxt_coef = load_coef(buf + min(j + 1u, (uint32_t)kTile - 1u) * 5);
                    test_triangle<R, kCount>(cur, sc, qin, slot0, vbase + j, hr, c_cand);
                    cur = nxt_coef;
                }
            }
            if (have_next) {
                float4 *dst = lds_tile + ((t + 1u) & 1u) * (kTile * 5);
#pragma unroll
                for (int k = 0; k < 5; ++k) dst[k * 256u + threadIdx.x] = stage[k];
            }
            __syncthreads();
        }
    }

    // ---- per ray: reload the path, sphere scan, closest hit, material response, compaction
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t slot = slot0 + r * 256u;
        const bool valid = slot < n_rays;
        bool alive = false;
        PathState s;
        if (valid) {
            s = load_ray(qin, slot, bounce == 0u);
            Hit h; h.t = kInf; h.material = 0; h.point = h.normal = mk(0.0f, 0.0f, 0.0f);
            const bool hit_sphere = sphere_pass(sc, s.o, s.d, h);                        // traverse (:433)
            const bool hit_mesh = hr.best_v[r] != 0xFFFFFFFFu;
            if (!hit_sphere && !hit_mesh) {                                              // :441-445
                f3 bg;
                if (P.use_envmap) { bg = env_lookup(sc, s.d); if (kCount) c_env++; }
                else bg = mk(P.background[0], P.background[1], P.background[2]);
                s.rad = s.rad + bg * s.thr;
            } else {
                if (!(h.t < hr.best_t[r])) {                                             // :447
                    const TriPlane pl = sc.tri_planes[hr.best_v[r]];
                    h.t = hr.best_t[r]; h.point = s.o + s.d * hr.best_t[r]; h.normal = mk(pl.nx, pl.ny, pl.nz); h.material = pl.material;
                }
                alive = shade_hit(sc, h, s.rng, s.o, s.d, s.thr, s.rad) && !last_bounce;
            }
            if (!alive) finish_path(P, im, wb, s, rng_out);
        }
        // wave-level compaction: ballot + prefix popcount, one atomic per wave
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            const int lane = threadIdx.x & 63;
            const int leader = (int)__builtin_ctzll(mask);
            uint32_t wave_base = 0;
            if (lane == leader) wave_base = atomicAdd(&wb.counts[bounce + 1u], (uint32_t)__popcll(mask));
            wave_base = __shfl(wave_base, leader);
            if (alive) store_ray(qout, wave_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)), s);
        }
    }
    if (kCount) {
        atomicAdd(&counters->candidates, c_cand);
        atomicAdd(&counters->env_lookups, c_env);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(&counters->segments, (unsigned long long)n_rays);
            atomicAdd(&counters->tri_tests, (unsigned long long)n_rays * n_tri);
        }
    }
}

// ---- split pipeline: intersect (2-D grid: ray blocks x triangle chunks) + shade ---------------------
// Every (ray block, triangle chunk) pair is an independent work item of identical cost, so the grid is
// thousands of equal blocks and the chip stays full on every bounce (the fused kernel above runs
// n_rays/(256 R) long blocks and loses up to a third of the machine to the last partial round).
// Chunks merge through one 64-bit atomicMin per ray that found a hit: key = (float bits of t) << 32 |
// visit index.  t > 0, so the unsigned order of the key is the order of (t, visit index) and the
// minimum is exactly the hit the reference's in-order scan with `t < max_t` keeps (:349).
constexpr unsigned long long kNoHitKey = 0xFFFFFFFFFFFFFFFFull;

// Hot-loop state of the split intersect kernel: 9 VGPRs per ray.  Filter survivors are not examined in
// the loop at all: their (ray, triangle) pair is parked in a small per-lane LDS list and drained after
// the scan, where the exact reference-order test runs with everything fetched from memory and a hit
// goes straight to the ray's atomicMin key.
template <int R>
struct ScanRays {
    f3 cv[R], d[R];
    float ncv[R], nd[R];
    float thresh[R];
};
constexpr int kCandSlots = 8;           // parked candidates per lane per work item (4 KiB of LDS per work-group)
constexpr uint32_t kMaxChunk = 4096;    // candidate entry = ray (4 bits) << 12 | triangle offset in chunk (12 bits)

// Two rays per instruction: v_pk_fma_f32 evaluates the same fma chain for a pair of rays with the triangle
// coefficient broadcast through op_sel, halving the instructions issued per test (the plain v_fma_f32
// stream is issue-limited at ~2.4 cycles per wave-instruction, tools/valu_rate.hip).  Each half is an
// ordinary IEEE fma, so the filter values are the ones filter_min() computes.
typedef float f2 __attribute__((ext_vector_type(2)));
struct RayPair { f2 cvx, cvy, cvz, dx, dy, dz; };

__device__ __forceinline__ f2 bc(float c) { return f2{c, c}; }
__device__ __forceinline__ f2 edge_chain2(float ex, float ey, float ez, float mx, float my, float mz, const RayPair &p)
{
    f2 f = bc(ex) * p.cvx;
    f = __builtin_elementwise_fma(bc(ey), p.cvy, f); f = __builtin_elementwise_fma(bc(ez), p.cvz, f);
    f = __builtin_elementwise_fma(bc(mx), p.dx, f);  f = __builtin_elementwise_fma(bc(my), p.dy, f);
    f = __builtin_elementwise_fma(bc(mz), p.dz, f);
    return f;
}
__device__ __forceinline__ f2 filter_min2(const TriCoef &T, const RayPair &p)
{
    f2 f0 = edge_chain2(T.q0.x, T.q0.y, T.q0.z, T.q2.y, T.q2.z, T.q2.w, p);
    f2 f1 = edge_chain2(T.q0.w, T.q1.x, T.q1.y, T.q3.x, T.q3.y, T.q3.z, p);
    f2 f2_ = edge_chain2(T.q1.z, T.q1.w, T.q2.x, T.q3.w, T.q4.x, T.q4.y, p);
    return __builtin_elementwise_min(__builtin_elementwise_min(f0, f1), f2_);
}

// one edge function: dot(e_k, cv) + dot(m_k, d) as a single fma chain
__device__ __forceinline__ float edge_chain(float ex, float ey, float ez, float mx, float my, float mz, f3 cv, f3 d)
{
    float f = ex * cv.x;
    f = __builtin_fmaf(ey, cv.y, f); f = __builtin_fmaf(ez, cv.z, f);
    f = __builtin_fmaf(mx, d.x, f);  f = __builtin_fmaf(my, d.y, f);  f = __builtin_fmaf(mz, d.z, f);
    return f;
}

// EARLY = wave-level short circuit (the SIMD form of the reference's `&&` between the three edge tests,
// :243-245): if no ray of the wave can pass edge 0 the other two edges are not evaluated, likewise after
// edge 1.  Pays off when the rays of a wave are coherent (camera rays: bounce 0), costs a few percent when
// they are not, so the host enables it per bounce.
template <int R, bool EARLY, bool PACKED>
__device__ __forceinline__ void scan_triangle(const TriCoef &T, const ScanRays<R> &sr, uint32_t off, uint16_t *cand, uint32_t &n_cand)
{
    if (!EARLY && PACKED && (R % 2) == 0) {
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            RayPair p;
            p.cvx = f2{sr.cv[2 * q].x, sr.cv[2 * q + 1].x}; p.cvy = f2{sr.cv[2 * q].y, sr.cv[2 * q + 1].y}; p.cvz = f2{sr.cv[2 * q].z, sr.cv[2 * q + 1].z};
            p.dx = f2{sr.d[2 * q].x, sr.d[2 * q + 1].x}; p.dy = f2{sr.d[2 * q].y, sr.d[2 * q + 1].y}; p.dz = f2{sr.d[2 * q].z, sr.d[2 * q + 1].z};
            const f2 mn = filter_min2(T, p);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 2 * q + h;
                if (!((h ? mn.y : mn.x) <= sr.thresh[r])) {                  // rare
                    if (n_cand < (uint32_t)kCandSlots) cand[n_cand * 256u] = (uint16_t)((r << 12) | off);
                    n_cand++;                                                // > kCandSlots marks overflow
                }
            }
        }
    } else if (!EARLY) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float mn = filter_min(T, sr.cv[r], sr.d[r]);
            if (!(mn <= sr.thresh[r])) {
                if (n_cand < (uint32_t)kCandSlots) cand[n_cand * 256u] = (uint16_t)((r << 12) | off);
                n_cand++;
            }
        }
    } else {
        float mn[R];
        unsigned long long any = 0;                                         // wave masks: v_cmp -> SGPR pair, s_or_b64
#pragma unroll
        for (int r = 0; r < R; ++r) {
            mn[r] = edge_chain(T.q0.x, T.q0.y, T.q0.z, T.q2.y, T.q2.z, T.q2.w, sr.cv[r], sr.d[r]);
            any |= __builtin_amdgcn_ballot_w64(!(mn[r] <= sr.thresh[r]));
        }
        if (any) {
            any = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mn[r] = __builtin_fminf(mn[r], edge_chain(T.q0.w, T.q1.x, T.q1.y, T.q3.x, T.q3.y, T.q3.z, sr.cv[r], sr.d[r]));
                any |= __builtin_amdgcn_ballot_w64(!(mn[r] <= sr.thresh[r]));
            }
            if (any) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    mn[r] = __builtin_fminf(mn[r], edge_chain(T.q1.z, T.q1.w, T.q2.x, T.q3.w, T.q4.x, T.q4.y, sr.cv[r], sr.d[r]));
                    if (!(mn[r] <= sr.thresh[r])) {
                        if (n_cand < (uint32_t)kCandSlots) cand[n_cand * 256u] = (uint16_t)((r << 12) | off);
                        n_cand++;
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ void exact_and_merge(const SceneView &sc, const RayQueue &qin, unsigned long long *best, uint32_t slot, uint32_t v)
{
    float4 a = qin.a[slot], b = qin.b[slot];
    TriRay tr; tr.o = mk(a.x, a.y, a.z); tr.d = mk(a.w, b.x, b.y); tr.cv = cross3(tr.d, tr.o); tr.ncv = tr.nd = 0.0f;
    float t = tri_exact(sc.tri_edges[v], sc.tri_planes[v], tr);
    if (kEps < t && t < kInf) atomicMin(&best[slot], ((unsigned long long)__float_as_uint(t) << 32) | v);
}

template <int R, int MODE, bool EARLY, bool PACKED, bool kCount>
__global__ void __launch_bounds__(256) intersect_kernel(SceneView sc, WaveBuffers wb, uint32_t bounce, uint32_t chunk_tris, Counters *counters)
{
    __shared__ float4 lds_tile[MODE == kLds ? kTile * 5 : 1];
    __shared__ uint16_t lds_cand[kCandSlots * 256];
    const uint32_t n_rays = wb.counts[bounce];
    const RayQueue qin = (bounce & 1u) ? wb.q[1] : wb.q[0];
    unsigned long long *best = (bounce & 1u) ? wb.best[1] : wb.best[0];
    const uint32_t v_begin = blockIdx.y * chunk_tris;
    const uint32_t v_end = min(v_begin + chunk_tris, sc.n_tri_visits);
    unsigned long long c_cand_total = 0;
    // grid-stride over ray blocks: the host sizes gridDim.x from the previous frame's ray counts (the
    // live count is only known on the device); any grid size is correct
    for (uint32_t base = blockIdx.x * (256u * R); base < n_rays; base += gridDim.x * (256u * R)) {
    const uint32_t slot0 = base + threadIdx.x;
    if (base != blockIdx.x * (256u * R)) __syncthreads();     // LDS of the previous ray block is still being read

    ScanRays<R> sr;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t slot = slot0 + r * 256u;
        const bool valid = slot < n_rays;
        f3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 0.0f);
        if (valid) { float4 a = qin.a[slot], b = qin.b[slot]; o = mk(a.x, a.y, a.z); d = mk(a.w, b.x, b.y); }
        TriRay tr = make_tri_ray(o, d);
        sr.cv[r] = tr.cv; sr.d[r] = d;
        sr.ncv[r] = valid ? tr.ncv : -__builtin_inff();       // empty slot: threshold +inf, nothing survives
        sr.nd[r] = valid ? tr.nd : -__builtin_inff();
        sr.thresh[r] = __builtin_inff();
    }
    uint16_t *cand = lds_cand + threadIdx.x;                  // entry k of this lane at cand[k * 256]
    uint32_t n_cand = 0;
    const float4 *src = reinterpret_cast<const float4 *>(sc.tri_edges);
    if (MODE == kScalar) {
        for (uint32_t v0 = v_begin; v0 < v_end; v0 += kBoundGroup) {        // chunk_tris is a multiple of kBoundGroup
            const float2 gb = wb.group_bounds[v0 / kBoundGroup];
#pragma unroll
            for (int r = 0; r < R; ++r) sr.thresh[r] = -(__builtin_fmaf(gb.x, sr.ncv[r], gb.y * sr.nd[r]) + 1e-30f);
            const uint32_t v1 = min(v0 + (uint32_t)kBoundGroup, v_end);
#pragma unroll 2
            for (uint32_t v = v0; v < v1; ++v)
                scan_triangle<R, EARLY, PACKED>(load_coef(src + (size_t)v * 5), sr, v - v_begin, cand, n_cand);
        }
    } else {
        // tiles of kTile records: HBM -> LDS with coalesced 16-byte loads (thread i moves float4 i, i+256, ...),
        // then every lane reads the records back as broadcasts.  No register staging across the scan:
        // the other work-groups resident on the CU cover the fill latency.
        const uint32_t n_tiles = (v_end - v_begin + kTile - 1) / kTile;
        for (uint32_t t = 0; t < n_tiles; ++t) {
            const uint32_t vbase = v_begin + t * kTile;
            const uint32_t cnt = min((uint32_t)kTile, v_end - vbase);
            const float4 *tsrc = src + (size_t)vbase * 5;
            if (t > 0) __syncthreads();
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                uint32_t i = k * 256u + threadIdx.x;
                if (i < cnt * 5u) lds_tile[i] = tsrc[i];
            }
            __syncthreads();
            for (uint32_t j0 = 0; j0 < cnt; j0 += kBoundGroup) {
                const float2 gb = wb.group_bounds[(vbase + j0) / kBoundGroup];
#pragma unroll
                for (int r = 0; r < R; ++r) sr.thresh[r] = -(__builtin_fmaf(gb.x, sr.ncv[r], gb.y * sr.nd[r]) + 1e-30f);
                const uint32_t j1 = min(j0 + (uint32_t)kBoundGroup, cnt);
                // no software prefetch of the next record: 78 VGPRs instead of 95 buys a sixth wave per SIMD, which
                // hides the LDS latency at least as well (A/B on one device: +1.5 %)
#pragma unroll 2
                for (uint32_t j = j0; j < j1; ++j)
                    scan_triangle<R, EARLY, PACKED>(load_coef(lds_tile + j * 5), sr, vbase - v_begin + j, cand, n_cand);
            }
        }
    }
    // ---- drain: exact test (reference operation order) of the parked survivors
    if (n_cand <= (uint32_t)kCandSlots) {
        for (uint32_t k = 0; k < n_cand; ++k) {
            const uint32_t e = cand[k * 256u];
            exact_and_merge(sc, qin, best, slot0 + (e >> 12) * 256u, v_begin + (e & 4095u));
        }
    } else {
        // more survivors than slots (stacked coplanar geometry): re-test this lane's rays against the whole chunk
        for (int r = 0; r < R; ++r) {
            const uint32_t slot = slot0 + r * 256u;
            if (slot < n_rays)
                for (uint32_t v = v_begin; v < v_end; ++v) exact_and_merge(sc, qin, best, slot, v);
        }
    }
    c_cand_total += n_cand;
    }
    if (kCount) {
        atomicAdd(&counters->candidates, c_cand_total);
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters->tri_tests, (unsigned long long)n_rays * (v_end - v_begin));
    }
}

// One lane per live ray: sphere scan, pick the nearer hit, material response, termination or
// compaction into the next queue (wave ballot + prefix popcount, one atomicAdd per wave).
// kSort: the survivors go to the staging queue with their bin key and their rank inside the bin (one atomic per ray on the bin's
// counter: the rays of a wave scatter over hundreds of bins); sort_prefix_kernel + sort_scatter_kernel finish the job.
__device__ __forceinline__ uint32_t spread2(uint32_t x)        // 8 bits -> every second bit
{
    x &= 0xffu; x = (x | (x << 4)) & 0x0f0fu; x = (x | (x << 2)) & 0x3333u; x = (x | (x << 1)) & 0x5555u;
    return x;
}
__device__ __forceinline__ uint32_t ray_bin_key(const WaveBuffers &wb, f3 o, f3 d)
{
    // direction: octahedral map of d / (|dx| + |dy| + |dz|) to 16 x 16 cells (a zero or non-finite direction lands in some cell: any
    // key is valid, only the tightness of the granules depends on it)
    const float l1 = fabsf(d.x) + fabsf(d.y) + fabsf(d.z);
    const float il = l1 > 0.0f ? 1.0f / l1 : 0.0f;
    float u = d.x * il, v = d.y * il;
    if (d.z < 0.0f) { const float uu = (1.0f - fabsf(v)) * (u < 0.0f ? -1.0f : 1.0f), vv = (1.0f - fabsf(u)) * (v < 0.0f ? -1.0f : 1.0f); u = uu; v = vv; }
    const int nd = 1 << wb.sort_db;
    const int iu = min(nd - 1, max(0, (int)((u * 0.5f + 0.5f) * (float)nd))), iv = min(nd - 1, max(0, (int)((v * 0.5f + 0.5f) * (float)nd)));
    const uint32_t dir = spread2((uint32_t)iu) | (spread2((uint32_t)iv) << 1);
    const uint32_t T = wb.sort_T;
    const float oa[3] = {o.x, o.y, o.z};
    const bool inside = o.x >= wb.sort_in_lo[0] && o.x <= wb.sort_in_hi[0] && o.y >= wb.sort_in_lo[1] && o.y <= wb.sort_in_hi[1] && o.z >= wb.sort_in_lo[2] && o.z <= wb.sort_in_hi[2];   // (NaN: outside)
    const uint32_t bits3 = inside ? wb.sort_in_bits : wb.sort_out_bits, order = inside ? wb.sort_in_order : wb.sort_out_order;
    uint32_t c[3], left[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t nb = (bits3 >> (4 * a)) & 15u;
        left[a] = nb;
        if (inside) c[a] = (uint32_t)fminf(fmaxf((oa[a] - wb.sort_lo[a]) * wb.sort_inv_cell[a], 0.0f), (float)((1u << nb) - 1u));
        else {
            const float u = fminf(fabsf(oa[a] - wb.sort_cen[a]) * wb.sort_inv_unit, 1.0e6f);                 // (NaN -> 1e6)
            const uint32_t j = min(7u, (__float_as_uint(1.0f + u) >> 23) - 127u);
            c[a] = (oa[a] >= wb.sort_cen[a] ? 8u + j : 7u - j) >> (4u - nb);
        }
    }
    uint32_t cell = 0u;
    const uint32_t n = left[0] + left[1] + left[2];             // <= T (the outside cells stop at four bits per axis)
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t a = (order >> (2u * i)) & 3u;
        const uint32_t l = (a == 0u ? left[0] : a == 1u ? left[1] : left[2]) - 1u;
        const uint32_t v = a == 0u ? c[0] : a == 1u ? c[1] : c[2];
        cell = (cell << 1) | ((v >> l) & 1u);
        left[0] = a == 0u ? l : left[0]; left[1] = a == 1u ? l : left[1]; left[2] = a == 2u ? l : left[2];
    }
    return (dir << (T + 1u)) | (inside ? 0u : 1u << T) | (cell << (T - n));
}

template <bool kCount, bool kSort>
__global__ void __launch_bounds__(256) shade_kernel(SceneView sc, FrameParams P, ImageView im, WaveBuffers wb,
                                                    uint32_t bounce, uint4 *rng_out, Counters *counters)
{
    const uint32_t n_rays = wb.counts[bounce];
    const RayQueue qin = (bounce & 1u) ? wb.q[1] : wb.q[0];
    const RayQueue qout = kSort ? wb.qt : ((bounce & 1u) ? wb.q[0] : wb.q[1]);
    const unsigned long long *best_in = (bounce & 1u) ? wb.best[1] : wb.best[0];
    unsigned long long *best_out = (bounce & 1u) ? wb.best[0] : wb.best[1];
    const bool last_bounce = (bounce + 1u >= P.max_bounce);
    unsigned long long c_env = 0;
    __shared__ uint32_t s_cnt[5];
    for (uint32_t blk = blockIdx.x; blk * 256u < n_rays; blk += gridDim.x) {      // grid-stride, see intersect_kernel
    const uint32_t slot = blk * 256u + threadIdx.x;
    const bool valid = slot < n_rays;
    bool alive = false;
    PathState s;
    if (valid) {
        s = load_ray(qin, slot, bounce == 0u);
        const unsigned long long key = best_in[slot];
        Hit h; h.t = kInf; h.material = 0; h.point = h.normal = mk(0.0f, 0.0f, 0.0f);
        const bool hit_sphere = sphere_pass(sc, s.o, s.d, h);                            // traverse (:433)
        const bool hit_mesh = key != kNoHitKey;
        if (!hit_sphere && !hit_mesh) {                                                  // :441-445
            f3 bg;
            if (P.use_envmap) { bg = env_lookup(sc, s.d); if (kCount) c_env++; }
            else bg = mk(P.background[0], P.background[1], P.background[2]);
            s.rad = s.rad + bg * s.thr;
        } else {
            const float mesh_t = hit_mesh ? __uint_as_float((uint32_t)(key >> 32)) : kInf;
            if (!(h.t < mesh_t)) {                                                       // :447
                const TriPlane pl = sc.tri_planes[(uint32_t)key];
                h.t = mesh_t; h.point = s.o + s.d * mesh_t; h.normal = mk(pl.nx, pl.ny, pl.nz); h.material = pl.material;
            }
            alive = shade_hit(sc, h, s.rng, s.o, s.d, s.thr, s.rad) && !last_bounce;
        }
        if (!alive) finish_path(P, im, wb, s, rng_out);
    }
    // compaction: ballot + prefix popcount inside the wave, the four wave totals summed through LDS, ONE atomicAdd per block on
    // the next queue's counter (every wave of the launch adding to that one address was the kernel's bottleneck: 32k atomics on
    // the camera-ray bounce)
    const unsigned long long mask = __ballot(alive);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_cnt[4] = total ? atomicAdd(&wb.counts[bounce + 1u], total) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t wave_base = s_cnt[4];
        for (int w = 0; w < wave; ++w) wave_base += s_cnt[w];
        const uint32_t out_slot = wave_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (kSort) {
            // (the bin's counter first: its round trip -- a device-scope atomic is performed at the memory side -- runs beside the ray's stores)
            // Lanes of the wave that share a bin (camera rays off a mirror, neighbours in a binned queue) take their ranks from ONE
            // atomic: the groups are found with a readlane + ballot per distinct key, the leaders' atomics go out together.
            const uint32_t key = ray_bin_key(wb, s.o, s.d);
            uint32_t grp_n = 1u, grp_rank = 0u; int grp_leader = lane;
            for (unsigned long long todo = __ballot(1); todo;) {
                const int l = __builtin_ctzll(todo);
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, l);
                const unsigned long long m = __ballot(key == k0);
                if (key == k0) { grp_n = (uint32_t)__popcll(m); grp_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); grp_leader = l; }
                todo &= ~m;
            }
            uint32_t base = 0u;
            if (lane == grp_leader) base = atomicAdd(wb.sort_hist + key, grp_n);
            const uint32_t rank = (uint32_t)__shfl((int)base, grp_leader) + grp_rank;
            store_ray(qout, out_slot, s);
            store_through(reinterpret_cast<unsigned long long *>(wb.sort_kr + out_slot), (unsigned long long)key | ((unsigned long long)rank << 32));
        } else { store_ray(qout, out_slot, s); store_through(best_out + out_slot, kNoHitKey); }
    }
    __syncthreads();                                                              // s_cnt is rewritten by the next batch
    }
    if (kCount) {
        atomicAdd(&counters->env_lookups, c_env);
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters->segments, (unsigned long long)n_rays);
    }
}

// ---- ray binning: bin counts -> first slots (two launches), then the move staging queue -> next queue.
// sort_sums_kernel: one block per 4096 bins, their sum.  sort_prefix_kernel: every block adds up the sums of the blocks before it and
// turns its own 4096 counts into first slots, in place.
constexpr uint32_t kSortSeg = 4096;
__global__ void __launch_bounds__(256) sort_sums_kernel(WaveBuffers wb)
{
    __shared__ uint32_t part[4];
    const uint32_t *h = wb.sort_hist + (size_t)blockIdx.x * kSortSeg;
    const uint4 *h4 = reinterpret_cast<const uint4 *>(h);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint4 v = h4[k * 256 + threadIdx.x]; s += v.x + v.y + v.z + v.w; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) store_through(wb.sort_hist + ((size_t)1u << wb.sort_bits) + blockIdx.x, part[0] + part[1] + part[2] + part[3]);
}
__global__ void __launch_bounds__(256) sort_prefix_kernel(WaveBuffers wb)
{
    __shared__ uint32_t part[4], wsum[4];
    const uint32_t *sums = wb.sort_hist + ((size_t)1u << wb.sort_bits);
    uint32_t before = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256u) before += sums[i];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = before;
    __syncthreads();
    before = part[0] + part[1] + part[2] + part[3];
    // 16 consecutive bins per thread: local sums, wave scan, block scan
    uint32_t *h = wb.sort_hist + (size_t)blockIdx.x * kSortSeg + threadIdx.x * 16u;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const uint4 *>(h)[k];
    uint32_t c[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint32_t t = c[k]; c[k] = tot; tot += t; }
    uint32_t inc = tot;                                       // inclusive scan over the wave
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = before + inc - tot;
    for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) store_through(reinterpret_cast<uint4 *>(h) + k, base + c[4 * k], base + c[4 * k + 1], base + c[4 * k + 2], base + c[4 * k + 3]);
}
// the rays entering `bounce` (just left in the staging queue by the shade kernel of bounce - 1) to their slots in key order
__global__ void __launch_bounds__(256) sort_scatter_kernel(WaveBuffers wb, uint32_t bounce)
{
    const uint32_t n_rays = wb.counts[bounce];
    const RayQueue qout = (bounce & 1u) ? wb.q[1] : wb.q[0];
    unsigned long long *best_out = (bounce & 1u) ? wb.best[1] : wb.best[0];
    // the other set of bin counters (last used two binned bounces ago; its scatter launch has ended): zero for the next binned bounce
    // (4 MB of stores beside this launch's ~200: in sort_prefix_kernel, a launch of 5 us, they cost 3 us)
    {
        uint4 *const z = reinterpret_cast<uint4 *>(wb.sort_hist_other);
        const uint32_t n4 = (uint32_t)(((size_t)1u << wb.sort_bits) / 4u);
        for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < n4; w += gridDim.x * 256u) store_through(z + w, 0u, 0u, 0u, 0u);
    }
    for (uint32_t slot = blockIdx.x * 256u + threadIdx.x; slot < n_rays; slot += gridDim.x * 256u) {
        const uint2 kr = wb.sort_kr[slot];
        const uint32_t to = wb.sort_hist[kr.x] + kr.y;
        const float4 a = wb.qt.a[slot], b = wb.qt.b[slot], c = wb.qt.c[slot];
        const uint4 g = wb.qt.rng[slot];
        const uint32_t px = wb.qt.pixel[slot];
        if (to >= n_rays) { store_through(best_out + slot, kNoHitKey); continue; }      // (cannot happen: the bins hold exactly the staged rays)
        store_through(qout.a + to, a.x, a.y, a.z, a.w);
        store_through(qout.b + to, b.x, b.y, b.z, b.w);
        store_through(qout.c + to, c.x, c.y, c.z, c.w);
        store_through(qout.rng + to, g.x, g.y, g.z, g.w);
        store_through(qout.pixel + to, px);
        store_through(best_out + slot, kNoHitKey);           // (every slot below n_rays is some thread's own: a coalesced store instead of a sixth scattered one)
    }
}

// per-group conservative bounds: max over kBoundGroup consecutive triangle records
__global__ void __launch_bounds__(64) group_bounds_kernel(const TriEdges *__restrict__ edges, uint32_t n, float2 *__restrict__ out)
{
    const uint32_t g = blockIdx.x, v = g * kBoundGroup + threadIdx.x;
    float be = 0.0f, bm = 0.0f;
    if (v < n) { be = edges[v].bound_e; bm = edges[v].bound_m; }
    // NaN bounds must poison the group (so nothing is ever rejected): fmaxf would drop them
    bool bad = !(be == be) || !(bm == bm);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        be = fmaxf(be, __shfl_xor(be, off));
        bm = fmaxf(bm, __shfl_xor(bm, off));
    }
    if (__any(bad)) { be = __builtin_nanf(""); bm = __builtin_nanf(""); }
    if (threadIdx.x == 0) out[g] = make_float2(be, bm);
}

}  // namespace rt
