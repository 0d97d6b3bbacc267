// rt_mfma.hpp -- ray x triangle scan with a bf16 matrix-core broad phase (kernel variant 3).
//
// The three edge functions of the triangle test (:243-245) are a K = 6 contraction
//     F[edge row][ray] = sum_k coef[edge row][k] * plucker[k][ray],   coef = (e_k, m_k), plucker = (d x o, d)
// i.e. a (3 N_tri) x 6 by 6 x N_ray matrix product.  The fp32 VALU scan (rt_wavefront.hpp) spends 20 vector
// instructions per test on it and is bound by the FP32 datapath (the exact-f32 MFMA shares that datapath,
// tools/mfma_valu_rate.hip).  The bf16 matrix pipe does the same contraction 16x faster but only with 8-bit
// significands -- far too coarse to DECIDE a hit, yet enough to REJECT almost everything conservatively:
//
//   * triangles are stored in Morton order and grouped (1..16 "quads" of 4 MFMA tiles x 10 triangles) around a
//     local origin c, so Plucker magnitudes are those of the neighbourhood, not of the world origin;
//   * per (ray, group) the lane computes cv' = d x (o - c) in fp32, packs it to bf16 (B operand) and computes a
//     threshold = -(bf16 error bound + fp32 bounds), see mf_margin() below;
//   * v_mfma_f32_32x32x16_bf16 has K = 16 but the contraction only needs 7 slots (6 + a bias that keeps padding
//     rows out), so the other 9 carry the bf16 RESIDUALS of the precomputed side and of d: the triangle
//     coefficients and the ray direction enter with 16-bit significands, only cv' stays at 8 bits (K layout at
//     MfView).  This cuts the error bound -- and with it the survivors -- by ~2.5x at no extra matrix work;
//   * one MFMA per (tile, 32 rays) yields the 30 edge values of 10 triangles for each ray (fp32 accumulate); a
//     triangle survives unless min(F0,F1,F2) <= threshold;
//   * survivors (a few per ray over the whole mesh) are queued in LDS and run through the exact reference-order
//     test (tri_exact), whose hits merge by the same 64-bit atomicMin as the fp32 scan.
//
// Exactness: a triangle the reference accepts has exact edge values F_k > -(rounding of the reference's own
// evaluation); the bf16 value differs from the exact one by at most the local bound (derivation at mf_margin());
// so it can never fall under the threshold.  What the broad phase lets through is irrelevant to the result:
// every survivor gets the exact test.  NaN / inf anywhere make the threshold NaN => everything survives.
#pragma once
#include "rt_wavefront.hpp"

#pragma clang fp contract(off)

namespace rt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMfTileTris = 10;                         // 2 lane-halves x 5 triangles x 3 edge rows (+1 spare row each)
constexpr int kMfQuadTiles = 4;                         // tiles fetched together (one "quad" = 40 triangles)
constexpr int kMfQuadTris = kMfTileTris * kMfQuadTiles;
constexpr uint32_t kMfMaxChunkQuads = 128;              // queued entry keeps the triangle offset inside the chunk in 16 bits
constexpr uint32_t kMfMaxGroupQuads = 64;               // a group = 1, 2, 4 .. 64 quads sharing one local origin and one set of bounds

struct alignas(16) MfGroup {
    float cx, cy, cz;   // local origin (centre of the group's bounding box)
    float E;            // >= max |e_k|
    float Ml;           // >= max |m'_k|                        (local moments)
    float Pw;           // >= max |v_a| |v_b| + E |c|           (fp32 rounding of the reference's own cross(v_a, v_b); origin shift with rounded e)
    float P;            // >= max |v'_a| |v'_b|                 (bounds the fp32 rounding of the local cross products)
    float pad1;
};

struct MfView {
    const MfGroup *groups; uint32_t n_groups;
    uint32_t group_quads;    // quads per group (power of two); storage is allocated in whole groups
    uint32_t n_quads;        // = n_groups * group_quads
    // A: per tile two 32-row x 8-bf16 panels (1 KiB), n_quads quads + one all-zero quad.  With x_hi = bf16(x) and
    // x_lo = bf16(x - x_hi), the K slots of a triangle-edge row and the matching B entries of a ray are
    //   panel 0 (k 0..7,  lanes 0-31):  e_hi.x e_hi.y e_hi.z  m_hi.x m_hi.y m_hi.z  bias  m_hi.x     x   cv.x cv.y cv.z  d_hi.x d_hi.y d_hi.z  1  d_lo.x
    //   panel 1 (k 8..15, lanes 32-63): e_lo.x e_lo.y e_lo.z  m_lo.x m_lo.y m_lo.z  m_hi.y m_hi.z    x   cv.x cv.y cv.z  d_hi.x d_hi.y d_hi.z  d_lo.y d_lo.z
    // i.e. F~ = (e_hi + e_lo).cv_hi + (m_hi + m_lo).d_hi + m_hi.d_lo  (+ bias, -3e38 on padding rows, 0 otherwise)
    const uint4 *A;
    uint32_t *dbg_log;       // diagnostics only
    const uint32_t *order;   // storage position -> visit index.  Triangles are stored in Morton order of their centroids
                             // so that the triangles of a group are neighbours (tight local bounds); hits are merged by VISIT
                             // index, so the reference's first-visited-wins tie rule (:349) is unaffected by the reordering
};

// row of the 32x32 accumulator tile held by lane-half h in register slot rho (ISA C/D layout)
__host__ __device__ constexpr int mf_row(int rho, int h) { return (rho & 3) + 8 * (rho >> 2) + 4 * h; }

// ---- upload time: local origins, bounds and the bf16 A matrices -------------------------------------------------
// One wave per group: bounding box (wave reduction) -> local origin; the group's A region zero-filled with the padding bias;
// one lane per triangle for the rows and the bounds (wave reduction of the maxima).
__device__ __forceinline__ float wave_max(float x) { for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off)); return x; }
__device__ __forceinline__ float wave_min(float x) { for (int off = 32; off > 0; off >>= 1) x = fminf(x, __shfl_xor(x, off)); return x; }

__global__ void __launch_bounds__(64) prepare_mfma_kernel(const float4 *__restrict__ vertices, const uint32_t *__restrict__ visit_tri,
                                                          const uint32_t *__restrict__ order, uint32_t n_visits, uint32_t n_groups,
                                                          uint32_t group_quads, MfGroup *__restrict__ groups, uint4 *__restrict__ A)
{
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t group_tiles = group_quads * kMfQuadTiles, group_tris = group_tiles * kMfTileTris;
    const uint32_t v_begin = g * group_tris, v_end = min(v_begin + group_tris, n_visits);
    f3 lo = mk(__builtin_inff(), __builtin_inff(), __builtin_inff()), hi = mk(-__builtin_inff(), -__builtin_inff(), -__builtin_inff());
    bool bad = false;
    for (uint32_t v = v_begin + lane; v < v_end; v += 64u)
        for (int k = 0; k < 3; ++k) {
            float4 p = vertices[3 * (size_t)visit_tri[order[v]] + k];
            lo = mk(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
            hi = mk(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
            bad |= !(fabsf(p.x) < 1e18f) || !(fabsf(p.y) < 1e18f) || !(fabsf(p.z) < 1e18f);   // non-finite or so large that products overflow
        }
    lo = mk(wave_min(lo.x), wave_min(lo.y), wave_min(lo.z));
    hi = mk(wave_max(hi.x), wave_max(hi.y), wave_max(hi.z));
    bad = __any(bad);
    const f3 c = mk(0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z);
    // [tile][panel][row][8 bf16]: everything zero except the bias (k = 6, panel 0) of -3e38: spare rows / missing triangles never survive
    uint4 *region = A + (size_t)g * group_tiles * 64;
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    bf16x2v biasv; biasv[0] = (__bf16)(-3.0e38f); biasv[1] = (__bf16)0.0f;
    const uint32_t bias_word = __builtin_bit_cast(uint32_t, biasv);
    for (uint32_t i = lane; i < group_tiles * 64u; i += 64u) region[i] = make_uint4(0u, 0u, 0u, (i & 32u) ? 0u : bias_word);
    __syncthreads();
    float E = 0.0f, Ml = 0.0f, Pw = 0.0f, P = 0.0f;
    __bf16 *rows = reinterpret_cast<__bf16 *>(region);
    for (uint32_t v = v_begin + lane; v < v_end; v += 64u) {
        const uint32_t tri = visit_tri[order[v]];                       // v = storage position
        f3 w[3], wl[3];
        for (int k = 0; k < 3; ++k) { float4 p = vertices[3 * (size_t)tri + k]; w[k] = mk(p.x, p.y, p.z); wl[k] = w[k] - c; }
        const uint32_t in_group = v - v_begin, tile = in_group / kMfTileTris, tt = in_group % kMfTileTris;
        const int h = (int)(tt / 5), u = (int)(tt % 5);
        for (int k = 0; k < 3; ++k) {
            const int a = (k + 1) % 3;                                   // edge k runs from vertex k to vertex a
            const f3 e = w[a] - w[k];
            const f3 ml = cross3(wl[a], wl[k]);
            E = fmaxf(E, __builtin_sqrtf(dot3(e, e)));
            Ml = fmaxf(Ml, __builtin_sqrtf(dot3(ml, ml)));
            P = fmaxf(P, __builtin_sqrtf(dot3(wl[a], wl[a])) * __builtin_sqrtf(dot3(wl[k], wl[k])));
            Pw = fmaxf(Pw, __builtin_sqrtf(dot3(w[a], w[a])) * __builtin_sqrtf(dot3(w[k], w[k])));
            __bf16 *row = rows + ((size_t)tile * 64 + mf_row(3 * u + k, h)) * 8, *row1 = row + 32 * 8;
            const float ev[3] = {e.x, e.y, e.z}, mv[3] = {ml.x, ml.y, ml.z};
            __bf16 mh[3];
            for (int i = 0; i < 3; ++i) {
                const __bf16 eh = (__bf16)ev[i]; mh[i] = (__bf16)mv[i];
                row[i] = eh; row[3 + i] = mh[i];
                row1[i] = (__bf16)(ev[i] - (float)eh);                   // the differences are exact in fp32
                row1[3 + i] = (__bf16)(mv[i] - (float)mh[i]);
            }
            row[6] = (__bf16)0.0f; row[7] = mh[0]; row1[6] = mh[1]; row1[7] = mh[2];
        }
    }
    // fmaxf drops NaNs: a NaN norm (non-finite vertex) is covered by `bad`
    E = wave_max(E); Ml = wave_max(Ml); P = wave_max(P); Pw = wave_max(Pw);
    if (lane != 0) return;
    MfGroup G;
    G.cx = c.x; G.cy = c.y; G.cz = c.z;
    const float nanv = __builtin_nanf("");
    // Moving the origin to c is exact only for e = v_a - v_k; with the rounded e the two forms of F differ by
    // d.((e_exact - e) x c) <= 2^-24 |e||c||d|: folded into Pw (the 2^-20 "world" term of mf_margin)
    Pw = Pw + E * __builtin_sqrtf(dot3(c, c));
    G.E = bad ? nanv : E * 1.001f; G.Ml = bad ? nanv : Ml * 1.001f; G.Pw = bad ? nanv : Pw * 1.001f;   // NaN bounds: nothing is ever rejected
    G.P = bad ? nanv : P * 1.001f; G.pad1 = 0.0f;
    groups[g] = G;
}

// ---- per ray (constant over the scan) and per (ray, group) quantities ------------------------------------------
struct MfRay {
    f3 o, d;
    float wd;        // >= |d|
    float wod;       // >= |o| |d|
    uint32_t dyz, tail;  // bf16 pairs (d_hi.y, d_hi.z) and, by lane half, (1, d_lo.x) or (d_lo.y, d_lo.z)
    uint32_t dx_hi;      // bf16(d.x) in the upper 16 bits
    bool valid;
};

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 p; p[0] = (__bf16)lo; p[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, p);
}

// mf_margin(): F := e.(d x o) + m.d in exact arithmetic on the float inputs (identical for world and local origin).
// (1) bf16 side.  u = 2^-8 is the bf16 unit round-off (8-bit significand, round to nearest even), x_hi = bf16(x),
//     x_lo = bf16(x - x_hi) (the difference is exact in fp32), so |x - x_hi| <= u|x| and |x - x_hi - x_lo| <= u^2|x|,
//     componentwise and hence in norm.  With e = fl(v_a - v_k), m' = fl(v'_a x v'_k), cv' = fl(d x fl(o - c)):
//         e.cv' - (e_hi + e_lo).cv_hi              = e.(cv' - cv_hi) + (e - e_hi - e_lo).cv_hi   <= (u + u^2(1+u)) |e||cv'|
//         m'.d - (m_hi + m_lo).d_hi - m_hi.d_lo    = (m' - m_hi - m_lo).d_hi + (m' - m_hi).d_lo + m'.(d - d_hi - d_lo)
//                                                                                               <= 3 u^2 (1+u) |m'||d|
//     The 16 products are exact in fp32; their accumulation inside the MFMA is bounded by 2^-17 (|e||cv'| + |m'||d|)
//     (16 additions, 2^-21 each: 8x the round-to-nearest figure, so truncating adders are covered too).  Together:
//         |F~ - F_local| <= 2^-8 (1 + 2^-6) |e||cv'| + 2^-14 |m'||d|
//     where F_local uses the fp32 values above; those differ from the exact local quantities by fp32 rounding of
//     d x (o - c) and v'_a x v'_k: <= 2^-21 (|e||o'| + P)|d| with P = |v'_a||v'_b| >= |m'|  ("cancel" below, 2^-20).
// (2) reference side.  The shader accepts edge k iff -A < B with A = dot(fl(e), fl(d x o)), B = dot(fl(v_a x v_b), d)
//     evaluated in fp32 (:226-245).  |A + B - F| <= 4w|e||d x o| + 3w|e||d||o| + 3w|v_a||v_b||d| + 3w|m||d|, w = 2^-24,
//     so an accepted edge has F > -(7w E|o||d| + 6w Pw|d|) with Pw >= |v_a||v_b| >= |m|  ("world" below, 2^-20 = 16w).
// A triangle the reference accepts therefore has F~_k > -(local + cancel + world) for all three edges.  E, Ml, P, Pw and
// the per-ray norms are upper bounds (inflated by 1.001 against their own rounding).
__device__ __forceinline__ float mf_margin(const MfGroup &G, float ncv, float no, const MfRay &r)
{
    float local = __builtin_fmaf(0.00396728515625f * G.E, ncv, 6.103515625e-05f * (G.Ml * r.wd));   // 2^-8 (1 + 2^-6) E|cv'| + 2^-14 Ml|d|
    float cancel = 9.5367431640625e-07f * (__builtin_fmaf(G.E, no, G.P) * r.wd);              // 2^-20 (E|o'| + P)|d|
    float world = 9.5367431640625e-07f * __builtin_fmaf(G.E, r.wod, G.Pw * r.wd);             // 2^-20 (E|o||d| + Pw|d|)
    return (local + cancel) + (world + 1e-30f);
}

// Exact reference-order test of the queued survivors, 64 per pass with every lane busy.  Inlined at exactly five places
// (twice per unrolled loop step, once after the loop): the queue holds half a step's worst case, so no flush is needed
// between the two tiles of a half step.  History: with a small queue the flush sat at every (tile, ray set, triangle) position;
// inlined there it put ~40 copies of the exact test between the hot instructions, and as an out-of-line function it
// LOST HITS nondeterministically (a few per 2 M rays; s_swappc callee reading the LDS queue -- flat_load or ds_read alike;
// the inlined form of the same source never did in any run).  Not understood, so no device function calls in this kernel.
// (A separate narrow-phase kernel fed through a global candidate buffer was also measured: 30 us per bounce on its own,
// but the extra launch and the buffer traffic made the frame 3% slower than doing it here.)
struct MfFlushArgs {
    const float4 *ray_a, *ray_b;
    const TriEdges *tri_edges; const TriPlane *tri_planes;
    unsigned long long *best;
    const uint32_t *order;
    uint32_t wave_slot0, v_chunk_begin, v_chunk_end;
    int debug_skip_exact;
    uint32_t *dbg_log;      // diagnostics (debug_skip_exact = 4): [0] = count, then (slot, storage position) pairs of every queued survivor
};

__device__ __forceinline__ void mf_flush(const MfFlushArgs &f, const uint32_t *queue, uint32_t qn)
{
    for (uint32_t i = threadIdx.x & 63u; i < qn; i += 64u) {
        const uint32_t e = queue[i];
        const uint32_t pos = f.v_chunk_begin + (e & 0xffffu);                 // storage position
        if (f.debug_skip_exact == 4 && pos < f.v_chunk_end) {
            const uint32_t at = atomicAdd(f.dbg_log, 1u);
            if (at < (1u << 22)) { f.dbg_log[2 + 2 * at] = f.wave_slot0 + (e >> 16); f.dbg_log[3 + 2 * at] = pos; }
        }
        if (pos < f.v_chunk_end && (f.debug_skip_exact == 0 || f.debug_skip_exact >= 4)) {
            const uint32_t slot = f.wave_slot0 + (e >> 16), v = f.order[pos];
            const float4 a = f.ray_a[slot], b = f.ray_b[slot];
            TriRay tr; tr.o = mk(a.x, a.y, a.z); tr.d = mk(a.w, b.x, b.y); tr.cv = cross3(tr.d, tr.o); tr.ncv = tr.nd = 0.0f;
            const float t = tri_exact(f.tri_edges[v], f.tri_planes[v], tr);
            if (kEps < t && t < kInf) atomicMin(&f.best[slot], ((unsigned long long)__float_as_uint(t) << 32) | v);   // as exact_and_merge()
        }
    }
}

// diagnostics (option debug_skip_exact = 3): every (ray, triangle) pair the broad phase REJECTS also gets the exact test;
// pairs the exact test accepts are logged -- the log must stay empty, anything in it is a hole in mf_margin()
struct MfVerifyLog { uint32_t n; uint32_t pad[3]; float ev[64][16]; };

#ifndef MF_EXAMINE_GAP
#define MF_EXAMINE_GAP 7     // s_nop operand of the fence behind the products: >= 8 wait states before the examination, whatever the compiler adds (costs 1.6%)
#endif
#ifndef MF_ISSUE_GAP
#define MF_ISSUE_GAP 1       // s_nop operand in front of the products: >= 2 wait states between the last VALU read and the overwrite of a block (-1: none)
#endif
#ifndef MF_MIN_BLOCKS
#define MF_MIN_BLOCKS 3      // waves per SIMD the register allocator must allow (four accumulator sets: 64 VGPRs)
#endif
// (The one-wave-per-SIMD form of this scan, kernel variant 4, lives in rt_scan.hpp.)
template <int S, bool kCount, bool kVerify = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MF_MIN_BLOCKS))) intersect_mfma_kernel(SceneView sc, WaveBuffers wb, MfView mf, uint32_t bounce,
                                                             uint32_t chunk_quads, Counters *__restrict__ counters, int debug_skip_exact)
{
    // grid: x = blocks of 4 waves x S ray sets x 32 rays (grid-stride), y = chunks of `chunk_quads` quads (a chunk may start
    // in the middle of a group: the group's origin and bounds are set up at the first quad of every chunk as well)
    // per-wave survivor queue, entry = (ray in wave) << 16 | triangle offset in chunk.  Half a loop step (2 tiles x S ray sets
    // x 5 triangles x 64 lanes) can add at most kStepMax entries, and the queue is drained between half steps once it holds kDrain
    constexpr uint32_t kStepMax = (kMfQuadTiles / 2) * S * 5 * 64, kDrain = 192, kQueue = kStepMax + kDrain;
    __shared__ uint32_t lds_queue[4 * kQueue];
    const uint32_t n_rays = wb.counts[bounce];
    const RayQueue qin = (bounce & 1u) ? wb.q[1] : wb.q[0];
    unsigned long long *best = (bounce & 1u) ? wb.best[1] : wb.best[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
    // only quads that hold triangles; the padding rows inside the last one carry a -3e38 bias and never survive
    const uint32_t q_begin = blockIdx.y * chunk_quads, q_end = min(q_begin + chunk_quads, min(mf.n_quads, (sc.n_tri_visits + kMfQuadTris - 1u) / kMfQuadTris));
    const uint32_t tile_begin = q_begin * kMfQuadTiles;
    const uint32_t v_chunk_begin = q_begin * kMfQuadTris, v_chunk_end = min(q_end * (uint32_t)kMfQuadTris, sc.n_tri_visits);
    const uint32_t group_mask = mf.group_quads - 1u, group_shift = (uint32_t)__builtin_ctz(mf.group_quads);
    if (q_begin >= q_end) return;                             // chunk behind the last quad that holds triangles
    constexpr uint32_t kRaysPerBlock = 4u * S * 32u;
    unsigned long long c_cand_total = 0;
    uint32_t *queue = lds_queue + wave * kQueue;
    MfFlushArgs fa{qin.a, qin.b, sc.tri_edges, sc.tri_planes, best, mf.order, 0u, v_chunk_begin, v_chunk_end, debug_skip_exact, mf.dbg_log};
    // group records through the constant address space: uniform index => s_load, which neither waits on nor is held up by
    // the vector-memory counter the A-tile prefetch uses
    typedef const float __attribute__((address_space(4))) *ConstFloats;
    const ConstFloats groups_k = (ConstFloats)(uintptr_t)mf.groups;
    // A tiles: every lane loads 16 bytes per tile (its row of its K panel) from base + a_off + 1024 t
    const char *A_bytes = reinterpret_cast<const char *>(mf.A);
    constexpr uint32_t kQuadBytes = kMfQuadTiles * 1024;

    for (uint32_t base = blockIdx.x * kRaysPerBlock; base < n_rays; base += gridDim.x * kRaysPerBlock) {
        MfRay ray[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t slot = base + (uint32_t)(wave * S + s) * 32u + (uint32_t)col;   // both lane halves hold the same ray
            MfRay &r = ray[s];
            r.valid = slot < n_rays;
            r.o = mk(0.0f, 0.0f, 0.0f); r.d = mk(0.0f, 0.0f, 0.0f);
            if (r.valid) { float4 a = qin.a[slot], b = qin.b[slot]; r.o = mk(a.x, a.y, a.z); r.d = mk(a.w, b.x, b.y); }
            r.wd = __builtin_sqrtf(dot3(r.d, r.d)) * 1.001f;
            r.wod = (__builtin_sqrtf(dot3(r.o, r.o)) * 1.001f) * r.wd;
            const uint32_t dxy = pack_bf16(r.d.x, r.d.y);
            r.dyz = pack_bf16(r.d.y, r.d.z);
            r.dx_hi = dxy << 16;
            const f3 dl = mk(r.d.x - __uint_as_float(dxy << 16), r.d.y - __uint_as_float(dxy & 0xffff0000u), r.d.z - __uint_as_float(r.dyz & 0xffff0000u));
            r.tail = half ? pack_bf16(dl.y, dl.z) : pack_bf16(1.0f, dl.x);
        }
        uint32_t qn = 0, n_total = 0;                            // wave-uniform
        const uint32_t wave_slot0 = base + (uint32_t)(wave * S) * 32u;
        auto flush = [&]() {
            fa.wave_slot0 = wave_slot0;
            mf_flush(fa, queue, qn);
            n_total += qn;
            qn = 0;
        };
        // The A tiles of the next quad are fetched while the current one is processed (the loads would otherwise sit
        // right in front of the MFMA that needs them: one exposed L2 round trip per tile).  Two register sets, the quad
        // loop is unrolled by two so that they swap roles without moves.
        uint32_t a_off = q_begin * kQuadBytes + (uint32_t)half * 512u + (uint32_t)col * 16u;
        auto fetch_quad = [&](uint4 (&dst)[kMfQuadTiles]) {            // fetches the quad a_off points at, then advances
#pragma unroll
            for (int t = 0; t < kMfQuadTiles; ++t) dst[t] = *reinterpret_cast<const uint4 *>(A_bytes + a_off + (uint32_t)(t * 1024));
            a_off += kQuadBytes;                                       // the quad after the last one is the zero padding: in bounds
        };
        bf16x8 B[S];
        float thresh[S];
#ifdef MF_CHECKSUM
        unsigned long long chk = 0ull, chk_any = 0ull;
#endif
        f32x16 accX[S], accY[S];
        uint32_t pend_tile = 0u; bool have_pend = false;                  // wave-uniform: the tile whose products wait in accY
        const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        // five v_min3 per ray set and lane, two v_max3, one compare: "does any triangle of this lane survive"
        auto minima = [&](int s, const f32x16 &acc, float (&mn)[5]) -> bool {
#pragma unroll
            for (int u = 0; u < 5; ++u) mn[u] = __builtin_fminf(__builtin_fminf(acc[3 * u], acc[3 * u + 1]), acc[3 * u + 2]);
            // A finite threshold means finite operands and edge values below 2^7 * 1e30 in magnitude, hence finite
            // minima: max-of-minima is then exactly "some triangle of this lane survives".  A NaN threshold passes all.
            const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mn[0], mn[1]), mn[2]), mn[3]), mn[4]);   // two v_max3
#ifdef MF_CHECKSUM          // diagnostics build: order-independent checksum of every examined minimum (debug_skip_exact = 5 prints it)
#pragma unroll
            for (int u = 0; u < 5; ++u) chk += (unsigned long long)__float_as_uint(mn[u]) * (0x9E3779B97F4A7C15ull + 2ull * (unsigned)u);
            chk_any += (unsigned long long)(!(mx <= thresh[s]));
#endif
            return !(mx <= thresh[s]);
        };
        // rare path: some lane has a survivor in `tile` -> per (ray set, triangle) ballots, survivors into the wave's LDS queue
        auto park = [&](uint32_t tile, const f32x16 (&acc)[S], const float (&mn)[S][5]) {
            if (kVerify) {
                MfVerifyLog *log = reinterpret_cast<MfVerifyLog *>(reinterpret_cast<char *>(counters) + 64);
                for (int s = 0; s < S; ++s)
                    for (int u = 0; u < 5; ++u) {
                        const uint32_t pos = v_chunk_begin + (tile - tile_begin) * kMfTileTris + 5u * (uint32_t)half + (uint32_t)u;
                        if (pos < v_chunk_end && ray[s].valid && mn[s][u] <= thresh[s]) {
                            const uint32_t v = mf.order[pos];
                            TriRay tr; tr.o = ray[s].o; tr.d = ray[s].d; tr.cv = cross3(tr.d, tr.o); tr.ncv = tr.nd = 0.0f;
                            const float t = tri_exact(sc.tri_edges[v], sc.tri_planes[v], tr);
                            if (kEps < t && t < kInf) {
                                const uint32_t at = atomicAdd(&log->n, 1u);
                                if (at < 64u) {
                                    float *e = log->ev[at];
                                    e[0] = (float)(wave_slot0 + s * 32 + col); e[1] = (float)v; e[2] = (float)pos; e[3] = (float)((tile / kMfQuadTiles) >> group_shift);
                                    e[4] = acc[s][3 * u]; e[5] = acc[s][3 * u + 1]; e[6] = acc[s][3 * u + 2]; e[7] = thresh[s];
                                    e[8] = tr.o.x; e[9] = tr.o.y; e[10] = tr.o.z; e[11] = tr.d.x; e[12] = tr.d.y; e[13] = tr.d.z; e[14] = t; e[15] = (float)bounce;
                                }
                            }
                        }
                    }
            }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(!(mn[s][u] <= thresh[s]));
                    if (m) {                                                        // wave-uniform
                        if ((m >> lane) & 1ull) {
                            const uint32_t v_off = (tile - tile_begin) * kMfTileTris + 5u * (uint32_t)half + (uint32_t)u;
                            queue[qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)(s * 32 + col) << 16) | v_off;
                        }
                        qn += (uint32_t)__popcll(m);
                    }
                }
        };
        // the common "nothing survived" case costs 8 VALU per product + 1 branch per tile
        auto examine = [&](uint32_t tile, f32x16 (&acc)[S]) {
            float mn[S][5];
            bool any_lane = false;
#pragma unroll
            for (int s = 0; s < S; ++s) any_lane |= minima(s, acc[s], mn[s]);
            if (__builtin_amdgcn_ballot_w64(any_lane) != 0ull) park(tile, acc, mn);
        };
        // One pipeline stage: the S products of the NEXT tile are issued back to back, then the PENDING tile (whose products were
        // issued one stage ago) is examined while they run: its 16 VALU instructions overlap the second matrix instruction
        // instead of following it.  `nxt` and `pend` are the two accumulator sets.
        //
        // HAZARD FENCE.  All four accumulators pass through the asm statement behind the products: they
        // stay in distinct registers for the whole loop, and the examination cannot move in front of the products.  The first
        // versions of this kernel let the compiler reuse one register block for consecutive products -- "mfma v[18:33]; s_nop;
        // 5 x v_min3 reading v18..v32; mfma v[18:33]" -- and a few times per 10^8 tiles the lanes 16-31 / 48-63 of one tile saw
        // wrong values (spurious or LOST survivors, i.e. lost hits, different in every run).  Also NOT done, although
        // tools/mfma_shadow_probe.hip shows that 5-6 independent VALU instructions issue for free right behind an MFMA:
        // examining one ray set directly behind each matrix instruction ("mfma; 8 VALU; mfma; 8 VALU") brought the fault back at
        // 10x the rate (and was no faster: 3 waves per SIMD already overlap) -- also with every temporary kept out of the
        // accumulator blocks, 2-8 wait states in front of each matrix instruction, up to 12 behind it, and compare masks given time
        // before scalar code reads them.  What every faulty variant had and no clean one: an examination placed BETWEEN the two
        // products of a tile.  tools/mfma_pipeline_probe.hip finds no fault in the same instruction patterns on known operands.
        // The mechanism is not established; this form (>= 8 wait states between the last product and the examination) repeats
        // its survivor set exactly (scripts/dbg_cand.py, scripts/dbg_soak.py, tests/test_gpu_fullsize.py).
        auto stage = [&](const uint4 &a, f32x16 (&nxt)[S], bool examine_pending, uint32_t pending_tile, f32x16 (&pend)[S]) {
            const bf16x8 Aop = __builtin_bit_cast(bf16x8, a);
#if MF_ISSUE_GAP >= 0
            asm volatile("s_nop %0" :: "n"(MF_ISSUE_GAP) : "memory");          // no register operands: pinning the accumulators here cost 5%
#endif
#pragma unroll
            for (int s = 0; s < S; ++s) nxt[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aop, B[s], zero, 0, 0, 0);
            asm volatile("s_nop %4" : "+v"(accX[0]), "+v"(accX[S - 1]), "+v"(accY[0]), "+v"(accY[S - 1]) : "n"(MF_EXAMINE_GAP));
            if (examine_pending) examine(pending_tile, pend);                            // wave-uniform condition
        };
        auto step = [&](uint32_t q, uint4 (&a_cur)[kMfQuadTiles], uint4 (&a_nxt)[kMfQuadTiles]) {
            fetch_quad(a_nxt);
            if ((q & group_mask) == 0u || q == q_begin) {                     // first quad of a group or of this chunk: local origin and bounds of its group
                if (have_pend) { examine(pend_tile, accY); have_pend = false; }  // judged by ITS group's thresholds, before they go
                const ConstFloats gp = groups_k + (size_t)(q >> group_shift) * (sizeof(MfGroup) / 4);   // wave-uniform: scalar loads
                MfGroup G;
                G.cx = gp[0]; G.cy = gp[1]; G.cz = gp[2]; G.E = gp[3]; G.Ml = gp[4]; G.Pw = gp[5]; G.P = gp[6]; G.pad1 = 0.0f;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const MfRay &r = ray[s];
                    const f3 ol = r.o - mk(G.cx, G.cy, G.cz);
                    const f3 cvl = cross3(r.d, ol);
                    // v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence: these are bounds, inflated by 1.001
                    const float ncv = __builtin_amdgcn_sqrtf(dot3(cvl, cvl)) * 1.001f, no = __builtin_amdgcn_sqrtf(dot3(ol, ol)) * 1.001f;
                    const float margin = mf_margin(G, ncv, no, r);
                    // empty slot: nothing survives.  Margin not finite or so large that the bf16 products could overflow
                    // (bounds NaN for non-finite vertices, huge coordinates): NaN threshold, everything survives.
                    thresh[s] = (!r.valid || debug_skip_exact == 2) ? __builtin_inff() : (margin < 1.0e30f ? -margin : __builtin_nanf(""));
                    uint4 bw;                                                                 // K layout: see MfView
                    bw.x = pack_bf16(cvl.x, cvl.y);
                    bw.y = (pack_bf16(cvl.z, 0.0f) & 0xffffu) | r.dx_hi;
                    bw.z = r.dyz; bw.w = r.tail;
                    B[s] = __builtin_bit_cast(bf16x8, bw);
                }
            }
            // Software pipeline over tiles (see stage()): two accumulator sets alternate; the one left pending at the end of a
            // step is always accY.
            const uint32_t tile0 = q * kMfQuadTiles;
            stage(a_cur[0], accX, have_pend, pend_tile, accY);
            stage(a_cur[1], accY, true, tile0, accX);
            if (qn >= kDrain) flush();
            stage(a_cur[2], accX, true, tile0 + 1u, accY);
            stage(a_cur[3], accY, true, tile0 + 2u, accX);
            pend_tile = tile0 + 3u; have_pend = true;
        };
        uint4 a0[kMfQuadTiles], a1[kMfQuadTiles];
        fetch_quad(a0);
        for (uint32_t q = q_begin; q < q_end; q += 2u) {
            step(q, a0, a1);
            if (qn >= kDrain) flush();
            if (q + 1u < q_end) {
                step(q + 1u, a1, a0);
                if (qn >= kDrain) flush();
            }
        }
        if (have_pend) examine(pend_tile, accY);
        flush();
#ifdef MF_CHECKSUM
        if (mf.dbg_log) { atomicAdd(reinterpret_cast<unsigned long long *>(mf.dbg_log) + 1, chk); atomicAdd(reinterpret_cast<unsigned long long *>(mf.dbg_log) + 2, chk_any); }
#endif
        const uint32_t n_cand = (lane == 0) ? n_total : 0u;
        c_cand_total += n_cand;
    }
    if (kCount) {
        atomicAdd(&counters->candidates, c_cand_total);
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters->tri_tests, (unsigned long long)n_rays * (v_chunk_end - v_chunk_begin));
    }
}

}  // namespace rt
