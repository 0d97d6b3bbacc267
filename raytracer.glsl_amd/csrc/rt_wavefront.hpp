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
};

struct PathState { f3 o, d, thr, rad; Rng rng; uint32_t pixel; };

__device__ __forceinline__ void store_ray(const RayQueue &q, uint32_t slot, const PathState &s)
{
    q.a[slot] = make_float4(s.o.x, s.o.y, s.o.z, s.d.x);
    q.b[slot] = make_float4(s.d.y, s.d.z, s.thr.x, s.thr.y);
    q.c[slot] = make_float4(s.thr.z, s.rad.x, s.rad.y, s.rad.z);
    q.rng[slot] = make_uint4(s.rng.x, s.rng.y, s.rng.z, s.rng.w);
    q.pixel[slot] = s.pixel;
}
__device__ __forceinline__ PathState load_ray(const RayQueue &q, uint32_t slot)
{
    PathState s;
    float4 a = q.a[slot], b = q.b[slot], c = q.c[slot];
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
    if (P.samples == 1u) {
        float4 *pix = im.pixels + s.pixel;
        f3 prev = mk(0.0f, 0.0f, 0.0f);
        if (!P.reset_flag) { float4 q = *pix; prev = mk(q.x, q.y, q.z); }
        *pix = accumulate_pixel(P, s.rad, prev);
    } else {
        float4 acc = wb.sums[s.pixel];
        wb.sums[s.pixel] = make_float4(acc.x + s.rad.x, acc.y + s.rad.y, acc.z + s.rad.z, 0.0f);
        wb.pix_rng[s.pixel] = make_uint4(s.rng.x, s.rng.y, s.rng.z, s.rng.w);
    }
    if (rng_out) rng_out[s.pixel] = make_uint4(s.rng.x, s.rng.y, s.rng.z, s.rng.w);
}

// ---- queue 0 ---------------------------------------------------------------------------------------
// sample 0: camera rays from scratch.  sample > 0: same camera ray, RNG continued (:556-559).
__global__ void __launch_bounds__(256) generate_rays_kernel(FrameParams P, ImageView im, WaveBuffers wb, uint32_t sample, uint32_t n0, Counters *counters)
{
    // queue order = 8x8 pixel blocks, row-major over blocks: neighbouring lanes start as neighbouring pixels
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n0) return;
    if (idx == 0u) {
        wb.counts[0] = n0;                               // the rest of counts[] was zeroed by the host
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
            wb.cam_a[s.pixel] = make_float4(s.o.x, s.o.y, s.o.z, s.d.x);
            wb.cam_b[s.pixel] = make_float4(s.d.y, s.d.z, 0.0f, 0.0f);
            wb.sums[s.pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else {
        float4 a = wb.cam_a[s.pixel], b = wb.cam_b[s.pixel];
        uint4 g = wb.pix_rng[s.pixel];
        s.o = mk(a.x, a.y, a.z); s.d = mk(a.w, b.x, b.y);
        s.rng.x = g.x; s.rng.y = g.y; s.rng.z = g.z; s.rng.w = g.w;
    }
    s.thr = mk(1.0f, 1.0f, 1.0f); s.rad = mk(0.0f, 0.0f, 0.0f);
    store_ray(wb.q[0], idx, s);
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
    *pix = accumulate_pixel(P, mk(sum.x, sum.y, sum.z), prev);
}

// ---- one bounce --------------------------------------------------------------------------------------
template <int R>
struct RayRegs {           // the hot-loop view of R rays
    TriRay tr[R];
    float ncv[R], nd[R];   // margin scales (-inf marks an empty slot)
    float thresh[R];       // -margin for the current bound group
    float best_t[R];
    uint32_t best_v[R];
};

template <int R>
__device__ __forceinline__ void set_thresh(RayRegs<R> &rr, float2 gb)
{
#pragma unroll
    for (int r = 0; r < R; ++r) rr.thresh[r] = -(__builtin_fmaf(gb.x, rr.ncv[r], gb.y * rr.nd[r]) + 1e-30f);
}

// filter value of one triangle for one ray: min over the three edge functions (see tri_filter)
__device__ __forceinline__ float tri_filter_min(const TriEdges &T, const TriRay &r)
{
    float f0 = T.e0x * r.cv.x;
    f0 = __builtin_fmaf(T.e0y, r.cv.y, f0); f0 = __builtin_fmaf(T.e0z, r.cv.z, f0);
    f0 = __builtin_fmaf(T.m0x, r.d.x, f0); f0 = __builtin_fmaf(T.m0y, r.d.y, f0); f0 = __builtin_fmaf(T.m0z, r.d.z, f0);
    float f1 = T.e1x * r.cv.x;
    f1 = __builtin_fmaf(T.e1y, r.cv.y, f1); f1 = __builtin_fmaf(T.e1z, r.cv.z, f1);
    f1 = __builtin_fmaf(T.m1x, r.d.x, f1); f1 = __builtin_fmaf(T.m1y, r.d.y, f1); f1 = __builtin_fmaf(T.m1z, r.d.z, f1);
    float f2 = T.e2x * r.cv.x;
    f2 = __builtin_fmaf(T.e2y, r.cv.y, f2); f2 = __builtin_fmaf(T.e2z, r.cv.z, f2);
    f2 = __builtin_fmaf(T.m2x, r.d.x, f2); f2 = __builtin_fmaf(T.m2y, r.d.y, f2); f2 = __builtin_fmaf(T.m2z, r.d.z, f2);
    return __builtin_fminf(__builtin_fminf(f0, f1), f2);
}

template <int R, bool kCount>
__device__ __forceinline__ void test_triangle(const TriEdges &T, const SceneView &sc, uint32_t v, RayRegs<R> &rr, unsigned long long &c_cand)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float mn = tri_filter_min(T, rr.tr[r]);
        if (!(mn <= rr.thresh[r])) {                         // rare: ~1 triangle in thousands
            if (kCount) c_cand++;
            float t = tri_exact(T, sc.tri_planes[v], rr.tr[r]);
            if (kEps < t && t < rr.best_t[r]) { rr.best_t[r] = t; rr.best_v[r] = v; }
        }
    }
}

template <int R, int MODE, bool kCount>
__global__ void __launch_bounds__(256) bounce_kernel(SceneView sc, FrameParams P, ImageView im, WaveBuffers wb,
                                                     uint32_t bounce, uint4 *rng_out, Counters *counters)
{
    __shared__ float4 lds_tile[MODE == kLds ? 2 * kTile * 5 : 1];
    const uint32_t n_rays = wb.counts[bounce];
    const uint32_t base = blockIdx.x * (256u * R);
    if (base >= n_rays) return;                               // uniform per work-group
    const RayQueue &qin = wb.q[bounce & 1u], &qout = wb.q[(bounce + 1u) & 1u];
    const bool last_bounce = (bounce + 1u >= P.max_bounce);

    PathState st[R];
    bool valid[R];
    RayRegs<R> rr;
    Hit h1[R];
    bool hit_sphere[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t slot = base + r * 256u + threadIdx.x;
        valid[r] = slot < n_rays;
        if (valid[r]) st[r] = load_ray(qin, slot);
        else { st[r].o = st[r].d = st[r].thr = st[r].rad = mk(0.0f, 0.0f, 0.0f); st[r].pixel = 0; st[r].rng.x = st[r].rng.y = st[r].rng.z = st[r].rng.w = 0; }
        h1[r].t = kInf; h1[r].material = 0; h1[r].point = h1[r].normal = mk(0.0f, 0.0f, 0.0f);
        hit_sphere[r] = valid[r] && sphere_pass(sc, st[r].o, st[r].d, h1[r]);          // traverse (:433)
        rr.tr[r] = make_tri_ray(st[r].o, st[r].d);
        rr.ncv[r] = valid[r] ? rr.tr[r].ncv : -__builtin_inff();   // empty slot: threshold becomes +inf, nothing passes
        rr.nd[r] = valid[r] ? rr.tr[r].nd : -__builtin_inff();
        rr.thresh[r] = __builtin_inff();
        rr.best_t[r] = kInf; rr.best_v[r] = 0xFFFFFFFFu;
    }

    unsigned long long c_cand = 0, c_env = 0;
    const uint32_t n_tri = sc.n_tri_visits;
    // ---- find_closest_mesh (:331-361): every ray against every triangle, in reference order
    if (MODE == kScalar) {
        for (uint32_t v0 = 0; v0 < n_tri; v0 += kBoundGroup) {
            set_thresh<R>(rr, wb.group_bounds[v0 / kBoundGroup]);
            const uint32_t v1 = min(v0 + (uint32_t)kBoundGroup, n_tri);
            for (uint32_t v = v0; v < v1; ++v)
                test_triangle<R, kCount>(sc.tri_edges[v], sc, v, rr, c_cand);
        }
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(sc.tri_edges);
        const uint32_t n_tiles = (n_tri + kTile - 1) / kTile;
        const uint32_t total_f4 = n_tri * 5u;
        float4 stage[5];
        // tile 0
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
                set_thresh<R>(rr, wb.group_bounds[(vbase + j0) / kBoundGroup]);
                const uint32_t j1 = min(j0 + (uint32_t)kBoundGroup, cnt);
                for (uint32_t j = j0; j < j1; ++j) {
                    TriEdges T;
                    float4 *tp = reinterpret_cast<float4 *>(&T);
#pragma unroll
                    for (int k = 0; k < 5; ++k) tp[k] = buf[j * 5 + k];
                    test_triangle<R, kCount>(T, sc, vbase + j, rr, c_cand);
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

    // ---- closest hit, material response, compaction into the next queue
#pragma unroll
    for (int r = 0; r < R; ++r) {
        bool alive = false;
        PathState &s = st[r];
        if (valid[r]) {
            const bool hit_mesh = rr.best_v[r] != 0xFFFFFFFFu;
            if (!hit_sphere[r] && !hit_mesh) {                                          // :441-445
                f3 bg;
                if (P.use_envmap) { bg = env_lookup(sc, s.d); if (kCount) c_env++; }
                else bg = mk(P.background[0], P.background[1], P.background[2]);
                s.rad = s.rad + bg * s.thr;
            } else {
                Hit h = h1[r];
                if (!(h1[r].t < rr.best_t[r])) {                                        // :447
                    const TriPlane &pl = sc.tri_planes[rr.best_v[r]];
                    h.t = rr.best_t[r]; h.point = s.o + s.d * rr.best_t[r]; h.normal = mk(pl.nx, pl.ny, pl.nz); h.material = pl.material;
                }
                alive = shade_hit(sc, h, s.rng, s.o, s.d, s.thr, s.rad) && !last_bounce;
            }
            if (!alive) finish_path(P, im, wb, s, rng_out);
        }
        // wave-level compaction: ballot + prefix popcount, one atomic per wave
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            const int lane = threadIdx.x & 63;
            uint32_t wave_base = 0;
            if (lane == (int)__builtin_ctzll(mask)) wave_base = atomicAdd(&wb.counts[bounce + 1u], (uint32_t)__popcll(mask));
            wave_base = __shfl(wave_base, (int)__builtin_ctzll(mask));
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
