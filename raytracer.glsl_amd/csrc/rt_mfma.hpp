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
    const MfCull *cull;      // one record per tile (packet culling)
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
                                                          uint32_t group_quads, MfGroup *__restrict__ groups, uint4 *__restrict__ A, float row_gamma)
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
        const float cn = __builtin_sqrtf(dot3(c, c));
        for (int k = 0; k < 3; ++k) {
            const int a = (k + 1) % 3;                                   // edge k runs from vertex k to vertex a
            const f3 e = w[a] - w[k];
            const f3 ml = cross3(wl[a], wl[k]);
            // Row scale (round 3).  All rows of a group are compared with ONE threshold per ray, so the threshold has to cover the
            // row that needs most: unscaled, a group with one long edge (E) taxed every short edge with E's bf16 error, E / |e| times
            // what it needed -- on a fine mesh with steep faces (100k benchmark field: edges 0.1 .. 0.65) survivors were ten times
            // the hits.  Each row is multiplied by a power of two sc ~ 1 / (|e| + 2^-13 pw), pw = |v_a||v_k| + |e||c| (the edge's
            // share of the reference's own rounding): exact in binary floating point, so the derivation of mf_margin holds for the
            // scaled row as it stands, with the group's bounds taken over the SCALED quantities; the two dominant terms of the
            // margin, 2^-8 sc|e||cv'| and 2^-20 sc pw |d|, then cost every edge about the same fraction of its own length.
            const float el = __builtin_sqrtf(dot3(e, e)), mll = __builtin_sqrtf(dot3(ml, ml));
            const float pl = __builtin_sqrtf(dot3(wl[a], wl[a])) * __builtin_sqrtf(dot3(wl[k], wl[k]));
            const float pw = __builtin_sqrtf(dot3(w[a], w[a])) * __builtin_sqrtf(dot3(w[k], w[k])) + el * cn;
            const float want = 1.0f / (el + row_gamma * pw);      // (row_gamma = 2^-13 unless tuned)
            float sc = (want > 1.0e-12f && want < 1.0e12f) ? __uint_as_float(__float_as_uint(want) & 0x7f800000u) : 1.0f;      // 2^floor(log2 want); degenerate edges: 1
            E = fmaxf(E, sc * el); Ml = fmaxf(Ml, sc * mll); P = fmaxf(P, sc * pl); Pw = fmaxf(Pw, sc * pw);
            __bf16 *row = rows + ((size_t)tile * 64 + mf_row(3 * u + k, h)) * 8, *row1 = row + 32 * 8;
            const float ev[3] = {sc * e.x, sc * e.y, sc * e.z}, mv[3] = {sc * ml.x, sc * ml.y, sc * ml.z};
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
    // d.((e_exact - e) x c) <= 2^-24 |e||c||d|: part of every edge's pw above (the 2^-20 "world" term of mf_margin)
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
// A wave scans a granule of 128 rays against a chunk of tiles (10 triangles each).  A tile may be skipped for the whole granule when
// EVERY ray of the granule is certified to be rejected by the reference's own test for EVERY triangle of the tile.  "The ray misses
// the tile's bounding volume" is not such a certificate: the reference tests LINES, and a line lying in (or within rounding noise of)
// the plane of a far-away triangle has all three edge functions at +-noise, so the reference may accept it with an arbitrary t.  The
// certificates therefore live in edge-function space.  Notation: triangle T with vertices v_k, edges e_k = v_{k+1} - v_k,
// N = e_0 x (v_2 - v_0), unit normal n^, centroid G, longest edge l_max, smallest altitude h_min (|N| = l_max h_min); line (o, d);
// F_k = d.((o - v_k) x e_k) the exact edge functions.  The reference rejects an edge whenever its exact value is
// <= -(7u E|o||d| + 6u Pw|d|), u = 2^-24 (mf_margin, part (2)); noise := 2^-20 (E|o| + Pw) per unit |d| bounds that from above.
//
//   (A) plane form.  With p = line /\ plane and beta_k the barycentric coordinates of p: F_k = -(d.N) beta_k.
//       front facing (d.N < 0): p at distance >= delta from the triangle  =>  some beta_k <= -delta s / h_max (s = sin(smallest
//       angle / 2), h_max the largest height)  =>  min_k F_k <= -|d.N| delta s / h_max;  back facing: max beta >= 1/3.
//       Certified when  |d^.N| min(1/3, delta s / h_max) >= noise.
//   (K) back faces.  d.N > 0: sum beta = 1  =>  max beta >= 1/3  =>  min_k F_k <= -(d.N)/3, wherever the line is.
//       Certified when  (d^.N)/3 >= noise.  (Secondary rays leave the front side of a single-sided mesh: for most of them most of
//       the mesh is back-facing.)
//   (B) moment form.  For any point q of the line, (q - v_k) x e_k = (q - G) x e_k - N/3, so with the moment of the line about the
//       centroid, w = d x (o - G):   F_k = e_k.w - (d.N)/3.   sum e_k = 0  =>  min_k e_k.w <= -mu/2 with mu = max_k |e_k.w|; the
//       e_k lie in the plane, so mu = max_k |e_k.w_p| (w_p = w - (w.n^) n^) = the extent of the triangle along w_p >= h_min |w_p|;
//       and d _|_ w  =>  |d^.n^| <= |w_p| / |w|.  Hence
//           min_k F_k <= -h_min |w_p| / 2 + |N| |d^.n^| / 3 <= -h_min |w_p| (1/2 - l_max / (3 |w|)),        |w| = distance(line, G)
//       Certified when  h_min |w_p| (1/2 - l_max / (3 |w|)) >= noise  and  |w| > 4/3 l_max.  |w_p| = |d^.n^| |p - G|: this is (A) with
//       the distance measured IN the plane, where a grazing line is far from the triangle -- what (A) loses to its separate guard.
//
// Per tile the record below bounds the right quantities for all 10 triangles: bounding sphere (C, R) of the vertices, Rc >= |G - C|,
// the unit normals as a rectangle in gnomonic coordinates about an axis a -- n^ = (a + x t1 + y t2) / sqrt(1 + x^2 + y^2), |x| <= X,
// |y| <= Y, (a, t1, t2) orthonormal, t1 along the direction the normals spread most (a curved strip of a height field spreads its
// normals by +-30 degrees one way and +-5 the other: a cone around a would be six times the area) --, Nmin = min |N|, shape = min
// s / h_max, hmin, lmax, E, Pw.  Per granule the kernel bounds the rays by an origin sphere (O, ro) and a direction cone (unit D,
// sigma = max |d^ - D|).  Then for every line of the granule and triangle of the tile
//       |d^.n^ - D.n^| <= sigma;     delta >= |(C - O) x D| - |C - O| sigma - ro - R;     |o| <= |O| + ro;
//       |w - W| <= ro + Rc + sigma |O - C| =: slack,   W = D x (O - C);     |w_p| >= sqrt(|W|^2 - M^2) - slack,  M >= max |W.n^|.
// Degenerate or non-finite triangles, or normals spread over more than ~84 degrees, make the record unusable (Nmin = 0: every
// comparison fails, the tile is scanned).  The theory is tested on the CPU against the reference's own fp32 test
// (tests/test_cull_certificate.py restates the record and the three certificates in numpy).
struct alignas(16) MfCull {
    float cx, cy, cz, R;          // bounding sphere of the tile's vertices
    float ax, ay, az, Rc;         // axis of the normals; radius of the centroids about (cx, cy, cz)
    float t1x, t1y, t1z, X;       // tangent frame and the half extents of the normals' gnomonic coordinates
    float t2x, t2y, t2z, Y;
    float Nmin, shape, hmin, lmax;  // lmax = the longest edge: also the E of the reference's rounding bound
    float Pw, inv_len;            // inv_len <= 1 / sqrt(1 + X^2 + Y^2)
    float iB, iB2;                // >= 1 / (1 + X^2), 1 / (1 + Y^2)
    float iB_lo, iB2_lo, pad0, pad1;   // <= the same
};
static_assert(sizeof(MfCull) == 112, "seven uint4 per tile");

__device__ __forceinline__ float wave_sum(float x) { for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off); return x; }

// one wave per tile, one lane per triangle (storage order)
__global__ void __launch_bounds__(64) prepare_cull_kernel(const float4 *__restrict__ vertices, const uint32_t *__restrict__ visit_tri,
                                                          const uint32_t *__restrict__ order, uint32_t n_visits, uint32_t n_tiles, MfCull *__restrict__ out)
{
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    if (q >= n_tiles) return;
    const uint32_t pos = q * kMfTileTris + lane;
    const bool have = lane < (uint32_t)kMfTileTris && pos < n_visits;
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
    float R = 0.0f, Rc = 0.0f, E = 0.0f, Pw = 0.0f, Nmin = inf, shape = inf, hmin = inf;
    f3 nh = mk(0.0f, 0.0f, 0.0f);
    if (have) {
        const f3 e0 = w[1] - w[0], e1 = w[2] - w[1], e2 = w[0] - w[2];
        const f3 N = cross3(e0, mk(-e2.x, -e2.y, -e2.z));                    // (v1 - v0) x (v2 - v0)
        const float nn = __builtin_sqrtf(dot3(N, N));
        const float l0 = __builtin_sqrtf(dot3(e0, e0)), l1 = __builtin_sqrtf(dot3(e1, e1)), l2 = __builtin_sqrtf(dot3(e2, e2));
        for (int k = 0; k < 3; ++k) { const f3 r = w[k] - c; R = fmaxf(R, __builtin_sqrtf(dot3(r, r))); }
        { const f3 g = mk((w[0].x + w[1].x + w[2].x) * (1.0f / 3.0f), (w[0].y + w[1].y + w[2].y) * (1.0f / 3.0f), (w[0].z + w[1].z + w[2].z) * (1.0f / 3.0f)) - c;
          Rc = __builtin_sqrtf(dot3(g, g)); }
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
        hmin = (E > 0.0f) ? 0.999f * nn / E : 0.0f;                          // smallest altitude = |N| / longest edge
        if (!(hmin == hmin)) hmin = 0.0f;
    }
    R = wave_max(R); Rc = wave_max(Rc); E = wave_max(E); Pw = wave_max(Pw); Nmin = wave_min(Nmin); shape = wave_min(shape); hmin = wave_min(hmin);
    bad = __any(bad) || __any(have && !(nh.x == nh.x && nh.y == nh.y && nh.z == nh.z));
    // axis of the normals: their normalised sum
    f3 a = mk(wave_sum(have ? nh.x : 0.0f), wave_sum(have ? nh.y : 0.0f), wave_sum(have ? nh.z : 0.0f));
    const float al = __builtin_sqrtf(dot3(a, a));
    a = (al > 0.0f) ? mk(a.x / al, a.y / al, a.z / al) : mk(0.0f, 0.0f, 1.0f);
    const float ca = have ? dot3(nh, a) : 1.0f;
    bad |= !(al > 0.0f) || __any(have && !(ca > 0.1f));                       // normals spread over more than ~84 degrees: no rectangle
    // gnomonic coordinates of the normals in a first tangent basis (b1, b2); t1 = the principal axis of their second moments
    const f3 ref = (fabsf(a.x) < 0.7f) ? mk(1.0f, 0.0f, 0.0f) : mk(0.0f, 1.0f, 0.0f);
    f3 b1 = cross3(a, ref);
    { const float bl = __builtin_sqrtf(dot3(b1, b1)); b1 = mk(b1.x / bl, b1.y / bl, b1.z / bl); }
    const f3 b2 = cross3(a, b1);
    const float ica = 1.0f / fmaxf(ca, 0.1f);
    const float gu = have ? dot3(nh, b1) * ica : 0.0f, gv = have ? dot3(nh, b2) * ica : 0.0f;
    const float suu = wave_sum(gu * gu), suv = wave_sum(gu * gv), svv = wave_sum(gv * gv);
    const float phi = 0.5f * atan2f(2.0f * suv, suu - svv);                  // (any frame is valid: only the tightness of the rectangle depends on phi)
    const float cp = cosf(phi), sp = sinf(phi);
    f3 t1 = mk(cp * b1.x + sp * b2.x, cp * b1.y + sp * b2.y, cp * b1.z + sp * b2.z);
    { const float tl = __builtin_sqrtf(dot3(t1, t1)); t1 = mk(t1.x / tl, t1.y / tl, t1.z / tl); }
    f3 t2 = cross3(a, t1);
    { const float tl = __builtin_sqrtf(dot3(t2, t2)); t2 = mk(t2.x / tl, t2.y / tl, t2.z / tl); }
    const float X = wave_max(have ? fabsf(dot3(nh, t1) * ica) : 0.0f), Y = wave_max(have ? fabsf(dot3(nh, t2) * ica) : 0.0f);
    if (lane != 0) return;
    MfCull rec;
    rec.cx = c.x; rec.cy = c.y; rec.cz = c.z; rec.R = R * 1.0001f + 1e-30f;
    rec.ax = a.x; rec.ay = a.y; rec.az = a.z; rec.Rc = Rc * 1.0001f + 1e-30f;
    rec.t1x = t1.x; rec.t1y = t1.y; rec.t1z = t1.z; rec.X = X * 1.001f + 1e-5f;
    rec.t2x = t2.x; rec.t2y = t2.y; rec.t2z = t2.z; rec.Y = Y * 1.001f + 1e-5f;
    bad |= !(rec.X < 16.0f) || !(rec.Y < 16.0f);
    rec.Nmin = bad ? 0.0f : Nmin * 0.999f; rec.shape = bad ? 0.0f : shape; rec.hmin = bad ? 0.0f : hmin; rec.lmax = E * 1.001f;
    rec.Pw = Pw * 1.001f;
    rec.inv_len = 0.9999f / __builtin_sqrtf(1.0f + rec.X * rec.X + rec.Y * rec.Y);
    rec.iB = 1.0001f / (1.0f + rec.X * rec.X); rec.iB2 = 1.0001f / (1.0f + rec.Y * rec.Y);
    rec.iB_lo = 0.9999f / (1.0f + rec.X * rec.X); rec.iB2_lo = 0.9999f / (1.0f + rec.Y * rec.Y);
    rec.pad0 = rec.pad1 = 0.0f;
    out[q] = rec;
}

// The packet of one granule and the three certificates against one tile record (the derivations are above).  Every bound carries
// explicit slack (1e-4 relative, 1e-5 / 1e-6 absolute) against its own fp32 evaluation, which uses 1-ulp v_rsq / v_sqrt.
struct MfPacket { f3 O, D; float ro, sigma, On; };

// upper bound of max over the record's normals of (w . n^), n^ = (a + x t1 + y t2) / sqrt(1 + x^2 + y^2) in the rectangle: the maximum
// of f(x, y) = (wa + x w1 + y w2) / sqrt(1 + x^2 + y^2) over a rectangle is |w| if the rectangle contains (w1, w2) / wa (wa > 0), else it
// sits on the boundary: a corner, or the stationary point of an edge (edge x = X: y* = w2 (1 + X^2) / A, A = wa + X w1, value
// sqrt(A^2 / (1 + X^2) + w2^2), a maximum iff A > 0).  w1, w2 enter by magnitude: the rectangle is symmetric.
__device__ __forceinline__ float mf_max_dot(const MfCull &c, float wa, float w1, float w2, float wn)
{
    const bool inside = (wa > 0.0f) && (w1 <= c.X * wa) && (w2 <= c.Y * wa);
    const float num = __builtin_fmaf(c.Y, w2, __builtin_fmaf(c.X, w1, wa));
    const float corner = num * c.inv_len * (num > 0.0f ? 1.0003f : 1.0f);           // (inv_len is a lower bound of 1 / sqrt(1 + X^2 + Y^2))
    const float A = __builtin_fmaf(c.X, w1, wa), A2 = __builtin_fmaf(c.Y, w2, wa);
    // (a stationary point behind the end of its edge: the function still rises at the corner, which has it.  iB = 1 / (1 + X^2) from
    // the record, rounded up: w2 (1 + X^2) <= Y A is tested as w2 <= Y A iB' with the rounded-down reciprocal -- erring towards
    // "behind the end" only moves the bound from the stationary value to the corner's, which differ in second order)
    const float e1 = (A > 0.0f && w2 <= c.Y * A * c.iB_lo) ? __builtin_amdgcn_sqrtf(__builtin_fmaf(A * A, c.iB, w2 * w2)) : -__builtin_inff();
    const float e2 = (A2 > 0.0f && w1 <= c.X * A2 * c.iB2_lo) ? __builtin_amdgcn_sqrtf(__builtin_fmaf(A2 * A2, c.iB2, w1 * w1)) : -__builtin_inff();
    const float m = fmaxf(corner, fmaxf(e1, e2));
    return inside ? wn : fminf(m + 1e-4f * wn, wn);
}

// (K) and (A): cheap, and enough for most tiles of a coherent granule; cmin and delta are handed on to nobody: (B) stands on its own
__device__ __forceinline__ bool mf_certified_ka(const MfCull &c, const MfPacket &p, const f3 g, float L, float nz, float Wn)
{
    const f3 a = mk(c.ax, c.ay, c.az), t1 = mk(c.t1x, c.t1y, c.t1z), t2 = mk(c.t2x, c.t2y, c.t2z);
    // directions against the normals: lower bounds of d^.n^ (all back facing when > 0) and of -d^.n^ (all front facing when > 0)
    const float Da = dot3(p.D, a), D1 = fabsf(dot3(p.D, t1)), D2 = fabsf(dot3(p.D, t2));
    const float spread = __builtin_fmaf(c.X, D1, c.Y * D2);
    const float num_pos = Da - spread, num_neg = -Da - spread;
    const float lo_pos = (num_pos > 0.0f ? num_pos * c.inv_len : num_pos) - p.sigma - 1e-5f;
    const float lo_neg = (num_neg > 0.0f ? num_neg * c.inv_len : num_neg) - p.sigma - 1e-5f;
    // (K) every triangle back facing for every ray
    const bool cert_k = (lo_pos > 0.0f) && ((c.Nmin * lo_pos) * 0.3333f * 0.99f >= nz);
    // (A) the lines miss the tile's sphere by delta, and |d^.n^| >= cmin
    const float delta = (Wn * 0.9999f - L * p.sigma) - (p.ro + c.R) - 1e-5f * (L + p.ro + c.R);
    const float cmin = fmaxf(lo_pos, lo_neg);
    const float lhs_a = (c.Nmin * cmin) * fminf(0.3333f, delta * c.shape) * 0.99f;
    const bool cert_a = (delta > 0.0f) && (cmin > 0.0f) && (lhs_a > 0.0f) && (lhs_a >= nz);
    return cert_k || cert_a;
}

// (B) the moment form
__device__ __forceinline__ bool mf_certified_b(const MfCull &c, const MfPacket &p, const f3 W, float L, float nz, float Wn)
{
    const f3 a = mk(c.ax, c.ay, c.az), t1 = mk(c.t1x, c.t1y, c.t1z), t2 = mk(c.t2x, c.t2y, c.t2z);
    const float slack = (p.ro + c.Rc) + L * p.sigma + 1e-5f * (L + p.ro + c.Rc);
    const float Wa = dot3(W, a), W1 = fabsf(dot3(W, t1)), W2 = fabsf(dot3(W, t2));
    const float Wn_up = Wn * 1.0001f, Wn_lo = Wn * 0.9999f;
    // M >= max |W.n^|: the rectangle's side of the sign of W.a in full; the other side's numerator is at most X W1 + Y W2 - |W.a| and
    // its denominator at least 1
    const float far_side = fmaxf(0.0f, __builtin_fmaf(c.Y, W2, __builtin_fmaf(c.X, W1, -fabsf(Wa)))) * 1.0001f;
    const float M = fmaxf(mf_max_dot(c, fabsf(Wa), W1, W2, Wn_up), far_side);
    const float wp = __builtin_amdgcn_sqrtf(fmaxf(0.0f, __builtin_fmaf(Wn_lo, Wn_lo, -(M * M)))) * 0.9999f - slack;
    const float dist = Wn_lo - slack;
    const float lhs_b = (c.hmin * wp) * (0.5f - (c.lmax * 0.33334f) * __builtin_amdgcn_rcpf(dist) * 1.0001f) * 0.99f;
    return (wp > 0.0f) && (dist > 1.3334f * c.lmax) && (M < Wn_lo) && (lhs_b > 0.0f) && (lhs_b >= nz);
}

// all three for one lane (the culling kernel evaluates (B) only where a wave has a lane that (K) and (A) left open)
__device__ __forceinline__ bool mf_certified(const MfCull &c, const MfPacket &p)
{
    const f3 g = p.O - mk(c.cx, c.cy, c.cz);                               // O - C
    const float L = __builtin_amdgcn_sqrtf(dot3(g, g)) * 1.0001f;
    const float nz = 9.5367431640625e-07f * __builtin_fmaf(c.lmax, p.On, c.Pw) * 1.01f;   // 2^-20 (E |o| + Pw): the reference's own rounding (mf_margin); lmax = E
    const f3 W = cross3(p.D, g);                                          // moment of the axis about the tile's centre
    const float Wn = __builtin_amdgcn_sqrtf(dot3(W, W));
    return (c.Nmin > 0.0f) && (mf_certified_ka(c, p, g, L, nz, Wn) || mf_certified_b(c, p, W, L, nz, Wn));       // any NaN: false
}

}  // namespace rt
