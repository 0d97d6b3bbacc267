// rt_mfma.hpp -- the bf16 matrix-core broad phase of the ray x triangle scan: data preparation and the error bound.
// (The scan kernel itself is rt_scan.hpp, kernel variant 4.)
//
// The three edge functions of the triangle test (:243-245) are a K = 6 contraction
//     F[edge row][ray] = sum_k coef[edge row][k] * plucker[k][ray],   coef = (e_k, m_k), plucker = (d x o, d)
// i.e. a (3 N_tri) x 6 by 6 x N_ray matrix product.  The fp32 VALU scan (rt_wavefront.hpp) spends 20 vector
// instructions per test on it and is bound by the FP32 datapath (the exact-f32 MFMA shares that datapath,
// tools/mfma_valu_rate.hip).  The bf16 matrix pipe does the same contraction 16x faster but only with 8-bit
// significands -- far too coarse to DECIDE a hit, yet enough to REJECT almost everything conservatively:
//
//   * triangles are stored in Morton order and grouped (1..64 "quads" of 4 MFMA tiles x 10 triangles) around a
//     local origin c, so Plucker magnitudes are those of the neighbourhood, not of the world origin;
//   * per (ray, group) the lane computes cv' = d x (o - c) in fp32, packs it to bf16 (B operand) and computes a
//     threshold = -(bf16 error bound + fp32 bounds), see mf_margin() below;
//   * v_mfma_f32_32x32x16_bf16 has K = 16 but the contraction only needs 7 slots (6 + a bias that keeps padding
//     rows out), so the other 9 carry the bf16 RESIDUALS of the precomputed side and of d: the triangle
//     coefficients and the ray direction enter with 16-bit significands, only cv' stays at 8 bits (K layout at
//     MfView).  This cuts the error bound -- and with it the survivors -- by ~2.5x at no extra matrix work;
//   * one MFMA per (tile, 32 rays) yields the 30 edge values of 10 triangles for each ray (fp32 accumulate); a
//     triangle survives unless min(F0,F1,F2) <= threshold;
//   * survivors (a few per ray over the whole mesh) are queued and run through the exact reference-order
//     test (tri_exact) by the narrow-phase kernel, whose hits merge by the same 64-bit atomicMin as the fp32 scan.
//
// Exactness: a triangle the reference accepts has exact edge values F_k > -(rounding of the reference's own
// evaluation); the bf16 value differs from the exact one by at most the local bound (derivation at mf_margin());
// so it can never fall under the threshold.  What the broad phase lets through is irrelevant to the result:
// every survivor gets the exact test.  NaN / inf anywhere make the threshold NaN => everything survives.
//
// History: round 1 also shipped this scan with three waves per SIMD ("kernel 3").  That variant lost or invented survivors a
// few times per 10^8 products, nondeterministically; the cause was never established (DESIGN.md section 5 lists what was ruled
// out, including a pipe-dense probe of the RAW / WAR patterns, tools/mfma_dense_probe.hip) and the variant was removed in round 2.
#pragma once
#include "rt_wavefront.hpp"

#pragma clang fp contract(off)

namespace rt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMfTileTris = 10;                         // 2 lane-halves x 5 triangles x 3 edge rows (+1 spare row each)
constexpr int kMfQuadTiles = 4;                         // tiles fetched together (one "quad" = 40 triangles)
constexpr int kMfQuadTris = kMfTileTris * kMfQuadTiles;
constexpr uint32_t kMfMaxChunkQuads = 32;               // kernel 4 keeps a chunk's A tiles in LDS: 32 quads = 128 KB
constexpr uint32_t kMfMaxGroupQuads = 64;               // a group = 1, 2, 4 .. 64 quads sharing one local origin and one set of bounds

struct alignas(16) MfGroup {
    float cx, cy, cz;   // local origin (centre of the group's bounding box)
    float E;            // >= max |e_k|
    float Ml;           // >= max |m'_k|                        (local moments)
    float Pw;           // >= max |v_a| |v_b| + E |c|           (fp32 rounding of the reference's own cross(v_a, v_b); origin shift with rounded e)
    float P;            // >= max |v'_a| |v'_b|                 (bounds the fp32 rounding of the local cross products)
    float pad1;
};

struct MfCull;
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
    const MfCull *cull;      // one record per quad (packet culling)
    const TriEdges *edges_s; const TriPlane *planes_s;   // the exact test's records in STORAGE order (copies of SceneView's, gathered at
                             // upload): the narrow phase reads them by storage position, one dependent round trip less than through `order`
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

// storage-ordered copies of the exact test's triangle records
__global__ void __launch_bounds__(256) gather_storage_order_kernel(const TriEdges *__restrict__ edges, const TriPlane *__restrict__ planes, const uint32_t *__restrict__ order,
                                                                   uint32_t n, TriEdges *__restrict__ edges_s, TriPlane *__restrict__ planes_s)
{
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos < n) { const uint32_t v = order[pos]; edges_s[pos] = edges[v]; planes_s[pos] = planes[v]; }
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

// ---- packet culling (the reference's disabled AABB cull, raytracer.glsl:258-270,288-292, done rigorously) ----------
// A wave scans 128 rays against a chunk of quads (40 triangles each).  A quad may be skipped for the whole wave when EVERY ray of
// the wave is certified to be rejected by the reference's own test for EVERY triangle of the quad.  "The ray misses the quad's
// bounding volume" is not such a certificate: the reference tests LINES, and a line lying in (or within rounding noise of) the
// plane of a far-away triangle has all three edge functions at +-noise, so the reference may accept it with an arbitrary t.  The
// certificate therefore lives in edge-function space.  With N = (v1-v0) x (v2-v0), p = line /\ plane and beta_k the barycentric
// coordinates of p, the exact edge functions are F_k = -(d.N) beta_k (F_0 = d.((o-v0) x e0) = d.((p-v0) x e0)), so
//   * front facing (d.N < 0): p at distance >= delta from the triangle  =>  some beta_k <= -delta s / h_max, s = sin(smallest
//     angle / 2), h_max the largest height (closest point on an edge: the edge's own coordinate; at a vertex: the normal cone of
//     the vertex gives cos >= sin(angle/2) for one of the two edges)  =>  min_k F_k <= -|d.N| delta s / h_max;
//   * back facing (d.N > 0): sum beta = 1  =>  max beta >= 1/3  =>  min_k F_k <= -(d.N)/3.
// The reference rejects an edge whenever its exact value is <= -(7w E|o||d| + 6w Pw|d|), w = 2^-24 (mf_margin, part (2)); so
//   |d^.N_T| min(1/3, delta s_T / h_T)  >=  2^-20 (E |o| + Pw)          for every triangle T of the quad, every ray of the wave
// certifies the skip (d^ = d/|d|: both sides scale with |d|).  Per quad the record below bounds the left side from below for any
// line that misses the quad's bounding sphere (C, R) by delta and whose direction makes |cos| >= cmin with every triangle normal
// (the unit normals of the quad lie in the box [nlo, nhi]: interval arithmetic on D.n^ -- a bumpy height field spreads its
// normals by +-50 degrees inside 40 triangles, far too much for a cone, but mostly across the viewing direction); per wave the
// kernel bounds its rays by an origin sphere (O, ro) and a direction cone (D, sigma = max |d^ - D|):
//   delta >= |(C-O) x D| - |C-O| sigma - ro - R,     cmin >= min |[D.n^]| - sigma  (0 not inside the interval),     |o| <= |O| + ro.
// Degenerate or non-finite triangles make the record unusable (Nmin = 0 or NaN: the comparison fails, the quad is scanned).
struct alignas(16) MfCull {
    float cx, cy, cz, R;          // bounding sphere of the quad's vertices
    float nlx, nly, nlz;          // box of the unit normals
    float nhx, nhy, nhz;
    float Nmin, shape;            // min |N_T|, min s_T / h_max,T
    float E, Pw;                  // max |e_k|, max |v_a||v_b|  (the reference's own rounding, world coordinates)
    float pad[2];
};
static_assert(sizeof(MfCull) == 64, "one LDS row of four uint4 per quad");

__device__ __forceinline__ float wave_sum(float x) { for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off); return x; }

// one wave per quad, one lane per triangle (storage order)
__global__ void __launch_bounds__(64) prepare_cull_kernel(const float4 *__restrict__ vertices, const uint32_t *__restrict__ visit_tri,
                                                          const uint32_t *__restrict__ order, uint32_t n_visits, uint32_t n_quads, MfCull *__restrict__ out)
{
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    if (q >= n_quads) return;
    const uint32_t pos = q * kMfQuadTris + lane;
    const bool have = lane < (uint32_t)kMfQuadTris && pos < n_visits;
    const float inf = __builtin_inff();
    f3 w[3] = {mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 0.0f)};
    if (have) { const uint32_t tri = visit_tri[order[pos]]; for (int k = 0; k < 3; ++k) { const float4 p = vertices[3 * (size_t)tri + k]; w[k] = mk(p.x, p.y, p.z); } }
    f3 lo = mk(inf, inf, inf), hi = mk(-inf, -inf, -inf);
    bool bad = false;
    if (have) for (int k = 0; k < 3; ++k) {
        lo = mk(fminf(lo.x, w[k].x), fminf(lo.y, w[k].y), fminf(lo.z, w[k].z));
        hi = mk(fmaxf(hi.x, w[k].x), fmaxf(hi.y, w[k].y), fmaxf(hi.z, w[k].z));
        bad |= !(fabsf(w[k].x) < 1e18f) || !(fabsf(w[k].y) < 1e18f) || !(fabsf(w[k].z) < 1e18f);
    }
    lo = mk(wave_min(lo.x), wave_min(lo.y), wave_min(lo.z));
    hi = mk(wave_max(hi.x), wave_max(hi.y), wave_max(hi.z));
    const f3 c = mk(0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z);
    float R = 0.0f, E = 0.0f, Pw = 0.0f, Nmin = inf, shape = inf;
    f3 nh = mk(0.0f, 0.0f, 0.0f);
    if (have) {
        const f3 e0 = w[1] - w[0], e1 = w[2] - w[1], e2 = w[0] - w[2];
        const f3 N = cross3(e0, mk(-e2.x, -e2.y, -e2.z));                    // (v1 - v0) x (v2 - v0)
        const float nn = __builtin_sqrtf(dot3(N, N));
        const float l0 = __builtin_sqrtf(dot3(e0, e0)), l1 = __builtin_sqrtf(dot3(e1, e1)), l2 = __builtin_sqrtf(dot3(e2, e2));
        for (int k = 0; k < 3; ++k) { const f3 r = w[k] - c; R = fmaxf(R, __builtin_sqrtf(dot3(r, r))); }
        E = fmaxf(l0, fmaxf(l1, l2));
        const float a0 = __builtin_sqrtf(dot3(w[0], w[0])), a1 = __builtin_sqrtf(dot3(w[1], w[1])), a2 = __builtin_sqrtf(dot3(w[2], w[2]));
        Pw = fmaxf(a0 * a1, fmaxf(a1 * a2, a2 * a0));
        Nmin = nn;
        nh = (nn > 0.0f) ? mk(N.x / nn, N.y / nn, N.z / nn) : mk(0.0f, 0.0f, 0.0f);
        // smallest interior angle: cos at the vertex between the two edges leaving it; sin(angle / 2) = sqrt((1 - cos) / 2)
        const float c0 = -dot3(e0, e2) / (l0 * l2), c1 = -dot3(e1, e0) / (l1 * l0), c2 = -dot3(e2, e1) / (l2 * l1);
        const float cmax = fminf(1.0f, fmaxf(c0, fmaxf(c1, c2)));
        const float s = __builtin_sqrtf(fmaxf(0.0f, 0.5f * (1.0f - cmax)));
        const float lmin = fminf(l0, fminf(l1, l2));
        shape = (nn > 0.0f) ? 0.999f * s * lmin / nn : 0.0f;                 // s / h_max, h_max = |N| / shortest edge
        if (!(shape == shape)) shape = 0.0f;
    }
    R = wave_max(R); E = wave_max(E); Pw = wave_max(Pw); Nmin = wave_min(Nmin); shape = wave_min(shape);
    f3 nl = have ? nh : mk(inf, inf, inf), nu = have ? nh : mk(-inf, -inf, -inf);
    nl = mk(wave_min(nl.x), wave_min(nl.y), wave_min(nl.z)); nu = mk(wave_max(nu.x), wave_max(nu.y), wave_max(nu.z));
    bad = __any(bad);
    if (lane != 0) return;
    MfCull rec;
    rec.cx = c.x; rec.cy = c.y; rec.cz = c.z; rec.R = R * 1.0001f + 1e-30f;
    rec.nlx = nl.x - 1e-6f; rec.nly = nl.y - 1e-6f; rec.nlz = nl.z - 1e-6f; rec.nhx = nu.x + 1e-6f; rec.nhy = nu.y + 1e-6f; rec.nhz = nu.z + 1e-6f;
    rec.Nmin = bad ? 0.0f : Nmin * 0.999f; rec.shape = bad ? 0.0f : shape;
    rec.E = E * 1.001f; rec.Pw = Pw * 1.001f;
    rec.pad[0] = rec.pad[1] = 0.0f;
    out[q] = rec;
}

}  // namespace rt
