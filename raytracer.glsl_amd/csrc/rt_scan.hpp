// rt_scan.hpp -- kernel variant 4 (default): the bf16 matrix-core broad phase of rt_mfma.hpp as a hand-ordered instruction stream, with
// one or two waves per SIMD, plus the kernels around it (packet culling, work items, narrow phase).
//
// The broad phase of find_closest_mesh (:331-361) is the contraction described in rt_mfma.hpp (A tiles, B operand, threshold, margin
// proof): F~[edge row][ray] per (tile of 10 triangles, set of 32 rays) from ONE v_mfma_f32_32x32x16_bf16, then "does any triangle of
// this lane survive" = 5 v_min3 + 2 v_max3 + 1 v_cmp on the 16 results.  What this file adds is the schedule.  An MFMA occupies the
// matrix pipe for 32 cycles but holds the vector issue port for 8 only; a VALU instruction costs a wave 4 (MI355X_MICROARCH.md):
//
//     tile t:   mfma  X0 <- A_t B_0     8 VALU examining Y0 (tile t-1, set 0)
//               mfma  X1 <- A_t B_1     8 VALU examining Y1
//               ...                      (4 ray sets; then ONE group of scalar instructions per TWO tiles)
//     tile t+1: the same with X and Y swapped
//
// i.e. every examination reads an accumulator whose matrix instruction was issued 4 products (>= 160 cycles) earlier, and every
// matrix instruction is followed by exactly the vector instructions that fit beside it.  The compiler's own order for the same source
// was "4 products back to back, then 32 VALU" (nothing overlaps: 125 cycles per product in round 1); here the order is written out in
// inline asm whose operands are bound to physical registers (the compiler sees ordinary dataflow and keeps out of the way).
//
// Around the stream: a block stages the A tiles of a chunk of <= 32 quads in LDS and its waves work through (granule of 128 rays,
// chunk) items -- handed out in fixed turns or claimed from counters (scan_solo_kernel below); survivors go to a per-wave LDS queue
// and from there, as dense records, to the wave's own region of the candidate buffer for narrow_phase_kernel (what does not fit is
// tested in place); packet_cull_kernel certifies, per granule and quad, that every ray misses every triangle by the reference's own
// arithmetic (rt_mfma.hpp, MfCull) so that the scan can skip the quad.
#pragma once
#include <type_traits>
#include "rt_mfma.hpp"

#pragma clang fp contract(off)

namespace rt {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x6 __attribute__((ext_vector_type(6)));

// max / min over the 32 columns of a wave whose two lane halves hold the same values: four DPP steps inside each row of 16 lanes,
// then the two rows through scalar registers (ds_bpermute-based shuffles cost an LDS round trip per step: 14 reductions of them
// were 70 % of the camera-ray bounce once culling had shortened the scan)
template <bool kMax> __device__ __forceinline__ float half_reduce(float x)
{
    auto op = [](float a, float b) { return kMax ? __builtin_fmaxf(a, b) : __builtin_fminf(a, b); };
    auto dpp = [](float v, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, false)); };
    x = op(x, dpp(x, std::integral_constant<int, 0xB1>{}));      // quad_perm [1,0,3,2]
    x = op(x, dpp(x, std::integral_constant<int, 0x4E>{}));      // quad_perm [2,3,0,1]
    x = op(x, dpp(x, std::integral_constant<int, 0x141>{}));     // row_half_mirror
    x = op(x, dpp(x, std::integral_constant<int, 0x140>{}));     // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
    return op(r0, r1);
}
__device__ __forceinline__ float half_max(float x) { return half_reduce<true>(x); }
__device__ __forceinline__ float half_min(float x) { return half_reduce<false>(x); }
// the same over all 64 lanes of a wave (distinct values in both halves)
template <bool kMax> __device__ __forceinline__ float full_reduce(float x)
{
    auto op = [](float a, float b) { return kMax ? __builtin_fmaxf(a, b) : __builtin_fminf(a, b); };
    auto dpp = [](float v, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, false)); };
    x = op(x, dpp(x, std::integral_constant<int, 0xB1>{}));
    x = op(x, dpp(x, std::integral_constant<int, 0x4E>{}));
    x = op(x, dpp(x, std::integral_constant<int, 0x141>{}));
    x = op(x, dpp(x, std::integral_constant<int, 0x140>{}));
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };
    return op(op(rl(0), rl(16)), op(rl(32), rl(48)));
}

// ---- packet culling (rt_mfma.hpp, MfCull): for every granule of 128 consecutive rays of a queue (= the rays one scan wave handles
// per trip) an origin sphere (O, ro), a direction cone (unit D, sigma = max |d^ - D|) and On >= max |o|, then the certificates against
// every tile of the mesh: bit t of the granule's row of wb.keep is CLEAR when all 128 rays are certified rejections for all 10
// triangles of tile t.  One wave per granule, two rays per lane for the bounds, one tile per lane for the test.  A granule with a
// non-finite origin, a direction that cannot be normalised, or directions spread too widely to have an axis keeps every tile.
// (Round 2 first evaluated the certificate inside the scan, per (wave, chunk) item: ~5,000 cycles per item for a scalar load of the
// bounds, a conflict-ridden LDS read of the chunk's records and the test itself, 40 % of the camera-ray bounce.)
__global__ void __launch_bounds__(256) packet_cull_kernel(WaveBuffers wb, const MfCull *__restrict__ cull, uint32_t n_tiles, uint32_t bounce, float ro_add, float sigma_add)
{
    const uint32_t n_rays = wb.counts[bounce];
    const RayQueue qin = (bounce & 1u) ? wb.q[1] : wb.q[0];
    const uint32_t lane = threadIdx.x & 63u, n_gran = (n_rays + 127u) / 128u;
    const float inf = __builtin_inff();
    for (uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6); g < n_gran; g += gridDim.x * 4u) {
        f3 o[2], dh[2]; bool valid[2]; bool usable = true;
        f3 olo = mk(inf, inf, inf), ohi = mk(-inf, -inf, -inf), dlo = olo, dhi = ohi;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t slot = g * 128u + (uint32_t)k * 64u + lane;
            valid[k] = slot < n_rays;
            o[k] = mk(0.0f, 0.0f, 0.0f); dh[k] = o[k];
            if (valid[k]) {
                const float4 a = qin.a[slot], b = qin.b[slot];
                o[k] = mk(a.x, a.y, a.z);
                const f3 d = mk(a.w, b.x, b.y);
                // (v_rsq_f32, 1 ulp: this feeds bounds that carry 1e-4 relative and 1e-6 absolute slack)
                const float dd = dot3(d, d), il = __builtin_amdgcn_rsqf(dd), dl = dd * il;
                dh[k] = mk(d.x * il, d.y * il, d.z * il);
                usable &= (dl > 0.0f) && (dl < inf) && (fabsf(o[k].x) < 1e18f) && (fabsf(o[k].y) < 1e18f) && (fabsf(o[k].z) < 1e18f);
                olo = mk(fminf(olo.x, o[k].x), fminf(olo.y, o[k].y), fminf(olo.z, o[k].z)); ohi = mk(fmaxf(ohi.x, o[k].x), fmaxf(ohi.y, o[k].y), fmaxf(ohi.z, o[k].z));
                dlo = mk(fminf(dlo.x, dh[k].x), fminf(dlo.y, dh[k].y), fminf(dlo.z, dh[k].z)); dhi = mk(fmaxf(dhi.x, dh[k].x), fmaxf(dhi.y, dh[k].y), fmaxf(dhi.z, dh[k].z));
            }
        }
        olo = mk(full_reduce<false>(olo.x), full_reduce<false>(olo.y), full_reduce<false>(olo.z)); ohi = mk(full_reduce<true>(ohi.x), full_reduce<true>(ohi.y), full_reduce<true>(ohi.z));
        dlo = mk(full_reduce<false>(dlo.x), full_reduce<false>(dlo.y), full_reduce<false>(dlo.z)); dhi = mk(full_reduce<true>(dhi.x), full_reduce<true>(dhi.y), full_reduce<true>(dhi.z));
        MfPacket pk;
        pk.O = mk(0.5f * olo.x + 0.5f * ohi.x, 0.5f * olo.y + 0.5f * ohi.y, 0.5f * olo.z + 0.5f * ohi.z);
        f3 D = mk(0.5f * dlo.x + 0.5f * dhi.x, 0.5f * dlo.y + 0.5f * dhi.y, 0.5f * dlo.z + 0.5f * dhi.z);
        const float DD = dot3(D, D), iDl = __builtin_amdgcn_rsqf(DD), Dl = DD * iDl;
        pk.D = mk(D.x * iDl, D.y * iDl, D.z * iDl);                           // |D| = 1 +- 3e-7: covered by the slack of sigma
        float ro = 0.0f, sigma = 0.0f;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (valid[k]) {
                const f3 eo = o[k] - pk.O, ed = dh[k] - pk.D;
                ro = fmaxf(ro, __builtin_amdgcn_sqrtf(dot3(eo, eo))); sigma = fmaxf(sigma, __builtin_amdgcn_sqrtf(dot3(ed, ed)));
            }
        // (ro_add, sigma_add: the camera-ray bits are kept for the frames that follow while the camera stands still; their rays differ from
        // this frame's by the depth-of-field jitter only, which the host bounds -- rtgl_amd.hip, camera_keep_valid)
        pk.ro = full_reduce<true>(ro) * 1.0001f + 1e-30f + ro_add; pk.sigma = full_reduce<true>(sigma) * 1.0001f + 2e-6f + sigma_add;
        usable = !__any(!usable) && (Dl > 0.25f);
        pk.On = __builtin_amdgcn_sqrtf(dot3(pk.O, pk.O)) * 1.0001f + pk.ro;
        uint32_t *const row = wb.keep + (size_t)g * wb.keep_words;
        // (the record of the next pass travels while this one is evaluated)
        MfCull c_next = {};
        bool have_next = usable && lane < n_tiles;
        if (have_next) c_next = cull[lane];
        for (uint32_t t0 = 0; t0 < n_tiles; t0 += 64u) {
            const MfCull c = c_next;
            const bool have_this = have_next;
            have_next = usable && t0 + 64u + lane < n_tiles;
            if (have_next) c_next = cull[t0 + 64u + lane];
            // (K) and (A) first: ~60 instructions that settle most tiles of a coherent granule; (B), twice that, only where a lane of the
            // wave is still open (64 consecutive tiles are neighbours in space: far from the granule they are settled together)
            bool skip = false, open = false;
            f3 W = mk(0.0f, 0.0f, 0.0f); float L = 0.0f, nz = 0.0f, Wn = 0.0f;
            if (have_this && c.Nmin > 0.0f) {
                const f3 gq = pk.O - mk(c.cx, c.cy, c.cz);
                L = __builtin_amdgcn_sqrtf(dot3(gq, gq)) * 1.0001f;
                nz = 9.5367431640625e-07f * __builtin_fmaf(c.lmax, pk.On, c.Pw) * 1.01f;
                W = cross3(pk.D, gq);
                Wn = __builtin_amdgcn_sqrtf(dot3(W, W));
                skip = mf_certified_ka(c, pk, gq, L, nz, Wn);
                open = !skip;
            }
            if (__any(open)) { if (open) skip = mf_certified_b(c, pk, W, L, nz, Wn); }
            const unsigned long long have = (n_tiles - t0 >= 64u) ? ~0ull : ((1ull << (n_tiles - t0)) - 1ull);
            const unsigned long long keep = ~(unsigned long long)__builtin_amdgcn_ballot_w64(skip) & have;
            if (lane == 0u) {
                store_through(row + (t0 >> 5), (uint32_t)keep);                     // (read by later kernels: rt_wavefront.hpp, store_through)
                if ((t0 >> 5) + 1u < wb.keep_words) store_through(row + ((t0 >> 5) + 1u), (uint32_t)(keep >> 32));
            }
        }
    }
}

// exact reference-order test (:243-249) of the triangle at storage position `pos` for the ray in queue slot `slot`; the hit key carries the
// VISIT index (the reference's first-visited-wins tie rule, :349), read beside the records, not before them
__device__ __forceinline__ void exact_and_merge_at(const MfView &mf, const RayQueue &qin, unsigned long long *best, uint32_t slot, uint32_t pos)
{
    const float4 a = qin.a[slot], b = qin.b[slot];
    const uint32_t v = mf.order[pos];
    TriRay tr; tr.o = mk(a.x, a.y, a.z); tr.d = mk(a.w, b.x, b.y); tr.cv = cross3(tr.d, tr.o); tr.ncv = tr.nd = 0.0f;
    const float t = tri_exact(mf.edges_s[pos], mf.planes_s[pos], tr);
    if (kEps < t && t < kInf) atomicMin(&best[slot], ((unsigned long long)__float_as_uint(t) << 32) | v);
}

constexpr int kSoloSets = 4;                                      // 32-ray sets per wave: 512 rays per block of four waves
struct SoloCfg {
    static constexpr uint32_t kStepMax = 2u * 64u;                    // a trip examines two tiles; a tile adds at most one entry per lane
    static constexpr uint32_t kDrain = 192u;                          // the queue is handed over once it holds this many
    static constexpr uint32_t kQueue = kStepMax + kDrain + 1u;        // entries per wave (+ one scratch entry lanes without a survivor write to)
    static constexpr uint32_t kRaysPerWave = (uint32_t)kSoloSets * 32u;
};

// ---- the hand-ordered instruction stream ---------------------------------------------------------------------------------------
// Register map of the hot loop (one wave per SIMD owns the whole file; VALU instructions address v0..v255 only, so everything
// the examination reads has to sit there):
//     X0..X3 = v[128:143] v[144:159] v[160:175] v[176:191]      accumulators of the tile in flight / pending, by ray set
//     Y0..Y3 = v[192:207] v[208:223] v[224:239] v[240:255]
//     MA0..3 = v[96:101] v[104:109] v[112:117] v[120:125]       minima (5 per set, slot 5 = their maximum) written by a Y stage
//     MB0..3 = v[64:69]  v[72:77]   v[80:85]   v[88:93]         ... by an X stage   (the two registers behind each block are the compiler's)
//     KA = s[36:43], KB = s[44:51]                              "lanes with a survivor" per ray set of the first / second stage of a trip
//                                                               (s32..s34 are the ABI's stack / frame registers: kept clear of)
// The blocks are bound with physical-register constraints, so the compiler sees ordinary dataflow (it keeps its own values out of
// the way and orders loads and waits), while the ORDER inside a statement is exactly what is written.
//
// One statement = one trip of the steady loop = two pipeline stages of 4 x (one matrix instruction, the 8 vector instructions
// examining the block the previous stage produced for the same ray set), then ONE group of scalar instructions that folds the
// eight "lanes with a survivor" masks into two scalars.  Measured with tools/scan_stage_rate.hip on MI355X (one wave per SIMD):
// the pure vector stream runs at 44-46 cycles per product; every place where scalar instructions interrupt it costs ~40 cycles,
// however few they are, and a scalar instruction that reads a mask a v_cmp has just written stalls for another ~30.  Hence one
// scalar spot per TWO stages (the fold, directly followed by the compiler's compare-and-branch, loop control and the LDS reads of
// the next two tiles), with the freshest mask folded last; the rare "park the survivors" path runs behind the statement, off the
// minima the statement leaves in MA (first stage) and MB (second stage).
// Hazards the compiler cannot see (it does not look into inline asm):
//   * MFMA result -> VALU read of the same block: a block is read one stage after it was written, i.e. at least 3 MFMAs and 27
//     VALU instructions later (>= 12 wait states needed for this 8-pass MFMA, tools/mfma_hazard_probe.hip); prologue and epilogue
//     of a segment put a full s_nop 15 in between;
//   * VALU write -> MFMA overwrite of a block (the M registers are outside the blocks): no wait states required on gfx950;
//   * MFMA SOURCE operands written by a VALU instruction: ONE wait state -- the hardware does not interlock this.  With a v_mov of
//     the A rows directly in front of the matrix instruction every lane multiplies a half-updated operand; one independent
//     instruction in between and none does (tools/mfma_src_hazard_probe.hip, profiles/r3_mfma_src_hazard_probe.txt).  The compiler
//     inserts that state for its own code, but a statement that BEGINS with a matrix instruction may find the compiler's copy of
//     the tile rows right in front of it: every such statement starts with RT_HEAD (round 3: the list-driven tile loop lost hits
//     in every culled configuration until it did).
#define RT_HEAD "s_nop 1\n\t"
#define RT_EXAM(PL, ML, KL, TH) \
    "v_min3_f32 v[" #ML "+0], v[" #PL "+0], v[" #PL "+1], v[" #PL "+2]\n\t" \
    "v_min3_f32 v[" #ML "+1], v[" #PL "+3], v[" #PL "+4], v[" #PL "+5]\n\t" \
    "v_min3_f32 v[" #ML "+2], v[" #PL "+6], v[" #PL "+7], v[" #PL "+8]\n\t" \
    "v_min3_f32 v[" #ML "+3], v[" #PL "+9], v[" #PL "+10], v[" #PL "+11]\n\t" \
    "v_min3_f32 v[" #ML "+4], v[" #PL "+12], v[" #PL "+13], v[" #PL "+14]\n\t" \
    "v_max3_f32 v[" #ML "+5], v[" #ML "+0], v[" #ML "+1], v[" #ML "+2]\n\t" \
    "v_max3_f32 v[" #ML "+5], v[" #ML "+5], v[" #ML "+3], v[" #ML "+4]\n\t" \
    "v_cmp_nle_f32_e64 s[" #KL ":" #KL "+1], v[" #ML "+5], " TH "\n\t"
#define RT_MFMA(NL, AOP, B) "v_mfma_f32_32x32x16_bf16 v[" #NL ":" #NL "+15], " AOP ", " B ", 0\n\t"
#define RT_FOLD(ANY, KL) "s_or_b64 " ANY ", s[" #KL ":" #KL "+1], s[" #KL "+2:" #KL "+3]\n\ts_or_b64 " ANY ", " ANY ", s[" #KL "+4:" #KL "+5]\n\ts_or_b64 " ANY ", " ANY ", s[" #KL "+6:" #KL "+7]\n\t"
#define RT_STAGE_Y_TEXT(AOP) RT_MFMA(192, AOP, "%[b0]") RT_EXAM(128, 96, 36, "%[t0]") RT_MFMA(208, AOP, "%[b1]") RT_EXAM(144, 104, 38, "%[t1]") \
                             RT_MFMA(224, AOP, "%[b2]") RT_EXAM(160, 112, 40, "%[t2]") RT_MFMA(240, AOP, "%[b3]") RT_EXAM(176, 120, 42, "%[t3]")
#define RT_STAGE_X_TEXT(AOP) RT_MFMA(128, AOP, "%[b0]") RT_EXAM(192, 64, 44, "%[t0]") RT_MFMA(144, AOP, "%[b1]") RT_EXAM(208, 72, 46, "%[t1]") \
                             RT_MFMA(160, AOP, "%[b2]") RT_EXAM(224, 80, 48, "%[t2]") RT_MFMA(176, AOP, "%[b3]") RT_EXAM(240, 88, 50, "%[t3]")
#define RT_K_CLOBBERS "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51"
// two stages: tile t (operand AY) -> Y while X (tile t-1) is examined into MA; tile t+1 (operand AX) -> X while Y is examined into MB
#define RT_TRIP(AY, AX) \
    asm volatile(RT_HEAD RT_STAGE_Y_TEXT("%[ay]") RT_STAGE_X_TEXT("%[ax]") RT_FOLD("%[anya]", 36) RT_FOLD("%[anyb]", 44) \
                 : "+{v[128:143]}"(X0), "+{v[144:159]}"(X1), "+{v[160:175]}"(X2), "+{v[176:191]}"(X3), \
                   "=&{v[192:207]}"(Y0), "=&{v[208:223]}"(Y1), "=&{v[224:239]}"(Y2), "=&{v[240:255]}"(Y3), \
                   "=&{v[96:101]}"(MA0), "=&{v[104:109]}"(MA1), "=&{v[112:117]}"(MA2), "=&{v[120:125]}"(MA3), \
                   "=&{v[64:69]}"(MB0), "=&{v[72:77]}"(MB1), "=&{v[80:85]}"(MB2), "=&{v[88:93]}"(MB3), [anya] "=&s"(any_a), [anyb] "=&s"(any_b) \
                 : [ay] "v"(AY), [ax] "v"(AX), [b0] "v"(B0), [b1] "v"(B1), [b2] "v"(B2), [b3] "v"(B3), [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3) \
                 : RT_K_CLOBBERS)
// the same trip, which also issues the LDS reads of the tile rows two trips ahead (into NY, NX; `addr` = LDS byte address of the
// next row to read, advanced by two rows) right behind its first matrix instruction and waits for them at its end, when they have
// long landed: nothing of the tile fetch is left for the scalar spot between two trips
#define RT_TRIP_L(AY, AX, NY, NX, ADDR) \
    asm volatile(RT_HEAD RT_MFMA(192, "%[ay]", "%[b0]") "ds_read_b128 %[ny], %[addr]\n\tds_read_b128 %[nx], %[addr] offset:1024\n\tv_add_u32_e32 %[addr], 0x800, %[addr]\n\t" \
                 RT_EXAM(128, 96, 36, "%[t0]") RT_MFMA(208, "%[ay]", "%[b1]") RT_EXAM(144, 104, 38, "%[t1]") \
                 RT_MFMA(224, "%[ay]", "%[b2]") RT_EXAM(160, 112, 40, "%[t2]") RT_MFMA(240, "%[ay]", "%[b3]") RT_EXAM(176, 120, 42, "%[t3]") \
                 RT_STAGE_X_TEXT("%[ax]") "s_waitcnt lgkmcnt(0)\n\t" RT_FOLD("%[anya]", 36) RT_FOLD("%[anyb]", 44) \
                 : "+{v[128:143]}"(X0), "+{v[144:159]}"(X1), "+{v[160:175]}"(X2), "+{v[176:191]}"(X3), \
                   "=&{v[192:207]}"(Y0), "=&{v[208:223]}"(Y1), "=&{v[224:239]}"(Y2), "=&{v[240:255]}"(Y3), \
                   "=&{v[96:101]}"(MA0), "=&{v[104:109]}"(MA1), "=&{v[112:117]}"(MA2), "=&{v[120:125]}"(MA3), \
                   "=&{v[64:69]}"(MB0), "=&{v[72:77]}"(MB1), "=&{v[80:85]}"(MB2), "=&{v[88:93]}"(MB3), [anya] "=&s"(any_a), [anyb] "=&s"(any_b), \
                   [ny] "=&v"(NY), [nx] "=&v"(NX), [addr] "+v"(ADDR) \
                 : [ay] "v"(AY), [ax] "v"(AX), [b0] "v"(B0), [b1] "v"(B1), [b2] "v"(B2), [b3] "v"(B3), [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3) \
                 : RT_K_CLOBBERS, "memory")
// the same trip for a LIST of tiles (culled bounces scan the kept tiles of a segment, wherever they are): the rows of the two tiles the
// NEXT trip multiplies are read from two explicit LDS addresses
#define RT_TRIP_G(AY, AX, NY, NX, ADDRY, ADDRX) \
    asm volatile(RT_HEAD RT_MFMA(192, "%[ay]", "%[b0]") "ds_read_b128 %[ny], %[addry]\n\tds_read_b128 %[nx], %[addrx]\n\t" \
                 RT_EXAM(128, 96, 36, "%[t0]") RT_MFMA(208, "%[ay]", "%[b1]") RT_EXAM(144, 104, 38, "%[t1]") \
                 RT_MFMA(224, "%[ay]", "%[b2]") RT_EXAM(160, 112, 40, "%[t2]") RT_MFMA(240, "%[ay]", "%[b3]") RT_EXAM(176, 120, 42, "%[t3]") \
                 RT_STAGE_X_TEXT("%[ax]") "s_waitcnt lgkmcnt(0)\n\t" RT_FOLD("%[anya]", 36) RT_FOLD("%[anyb]", 44) \
                 : "+{v[128:143]}"(X0), "+{v[144:159]}"(X1), "+{v[160:175]}"(X2), "+{v[176:191]}"(X3), \
                   "=&{v[192:207]}"(Y0), "=&{v[208:223]}"(Y1), "=&{v[224:239]}"(Y2), "=&{v[240:255]}"(Y3), \
                   "=&{v[96:101]}"(MA0), "=&{v[104:109]}"(MA1), "=&{v[112:117]}"(MA2), "=&{v[120:125]}"(MA3), \
                   "=&{v[64:69]}"(MB0), "=&{v[72:77]}"(MB1), "=&{v[80:85]}"(MB2), "=&{v[88:93]}"(MB3), [anya] "=&s"(any_a), [anyb] "=&s"(any_b), \
                   [ny] "=&v"(NY), [nx] "=&v"(NX) \
                 : [ay] "v"(AY), [ax] "v"(AX), [b0] "v"(B0), [b1] "v"(B1), [b2] "v"(B2), [b3] "v"(B3), [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3), \
                   [addry] "v"(ADDRY), [addrx] "v"(ADDRX) \
                 : RT_K_CLOBBERS, "memory")
// a single stage (a segment with an even number of tiles ends with one), the products of a segment's first tile, and the
// examination of its last one (nothing to overlap with)
#define RT_STAGE_Y(AY) \
    asm volatile(RT_HEAD RT_STAGE_Y_TEXT("%[ay]") RT_FOLD("%[anya]", 36) \
                 : "=&{v[192:207]}"(Y0), "=&{v[208:223]}"(Y1), "=&{v[224:239]}"(Y2), "=&{v[240:255]}"(Y3), \
                   "=&{v[96:101]}"(MA0), "=&{v[104:109]}"(MA1), "=&{v[112:117]}"(MA2), "=&{v[120:125]}"(MA3), [anya] "=&s"(any_a) \
                 : "{v[128:143]}"(X0), "{v[144:159]}"(X1), "{v[160:175]}"(X2), "{v[176:191]}"(X3), \
                   [ay] "v"(AY), [b0] "v"(B0), [b1] "v"(B1), [b2] "v"(B2), [b3] "v"(B3), [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3) \
                 : RT_K_CLOBBERS)
#define RT_PRODUCTS_X(AOP) \
    asm volatile(RT_HEAD RT_MFMA(128, "%[a]", "%[b0]") RT_MFMA(144, "%[a]", "%[b1]") RT_MFMA(160, "%[a]", "%[b2]") RT_MFMA(176, "%[a]", "%[b3]") "s_nop 15" \
                 : "=&{v[128:143]}"(X0), "=&{v[144:159]}"(X1), "=&{v[160:175]}"(X2), "=&{v[176:191]}"(X3) \
                 : [a] "v"(AOP), [b0] "v"(B0), [b1] "v"(B1), [b2] "v"(B2), [b3] "v"(B3))
#define RT_EXAMINE_X() \
    asm volatile("s_nop 15\n\t" RT_EXAM(128, 96, 36, "%[t0]") RT_EXAM(144, 104, 38, "%[t1]") RT_EXAM(160, 112, 40, "%[t2]") RT_EXAM(176, 120, 42, "%[t3]") "s_nop 7\n\t" RT_FOLD("%[anya]", 36) \
                 : "=&{v[96:101]}"(MA0), "=&{v[104:109]}"(MA1), "=&{v[112:117]}"(MA2), "=&{v[120:125]}"(MA3), [anya] "=&s"(any_a) \
                 : "{v[128:143]}"(X0), "{v[144:159]}"(X1), "{v[160:175]}"(X2), "{v[176:191]}"(X3), [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3) \
                 : RT_K_CLOBBERS)
#define RT_EXAMINE_Y() \
    asm volatile("s_nop 15\n\t" RT_EXAM(192, 64, 44, "%[t0]") RT_EXAM(208, 72, 46, "%[t1]") RT_EXAM(224, 80, 48, "%[t2]") RT_EXAM(240, 88, 50, "%[t3]") "s_nop 7\n\t" RT_FOLD("%[anyb]", 44) \
                 : "=&{v[64:69]}"(MB0), "=&{v[72:77]}"(MB1), "=&{v[80:85]}"(MB2), "=&{v[88:93]}"(MB3), [anyb] "=&s"(any_b) \
                 : "{v[192:207]}"(Y0), "{v[208:223]}"(Y1), "{v[224:239]}"(Y2), "{v[240:255]}"(Y3), [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3) \
                 : RT_K_CLOBBERS)

// the 20-bit survivor mask of a lane from a minima block set: bit 5 s + u = !(minimum u of ray set s <= threshold s).  Built with the
// carry chain (v_cmp -> vcc, v_addc_co: mask = 2 mask + vcc), highest bit first: 40 vector instructions, no scalar ones.
#define RT_MASK_BIT(ML, U, TH) "v_cmp_nle_f32_e32 vcc, v[" #ML "+" #U "], " TH "\n\tv_addc_co_u32_e32 %[m], vcc, %[m], %[m], vcc\n\t"
#define RT_MASK_SET(ML, TH) RT_MASK_BIT(ML, 4, TH) RT_MASK_BIT(ML, 3, TH) RT_MASK_BIT(ML, 2, TH) RT_MASK_BIT(ML, 1, TH) RT_MASK_BIT(ML, 0, TH)
#define RT_MASK_A(MASK) \
    asm volatile("v_mov_b32_e32 %[m], 0\n\t" RT_MASK_SET(120, "%[t3]") RT_MASK_SET(112, "%[t2]") RT_MASK_SET(104, "%[t1]") RT_MASK_SET(96, "%[t0]") \
                 : [m] "=&v"(MASK) : "{v[96:101]}"(MA0), "{v[104:109]}"(MA1), "{v[112:117]}"(MA2), "{v[120:125]}"(MA3), \
                   [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3) : "vcc")
#define RT_MASK_B(MASK) \
    asm volatile("v_mov_b32_e32 %[m], 0\n\t" RT_MASK_SET(88, "%[t3]") RT_MASK_SET(80, "%[t2]") RT_MASK_SET(72, "%[t1]") RT_MASK_SET(64, "%[t0]") \
                 : [m] "=&v"(MASK) : "{v[64:69]}"(MB0), "{v[72:77]}"(MB1), "{v[80:85]}"(MB2), "{v[88:93]}"(MB3), \
                   [t0] "v"(th0), [t1] "v"(th1), [t2] "v"(th2), [t3] "v"(th3) : "vcc")

// diagnostics build (-DRT_SOLO_STAMPS): s_memtime stamps around the phases of a wave's life, summed into mf.dbg_log
// (-DRT_SOLO_STAMPS=2: only the begin and the end of every wave, on the constant 100 MHz clock: mean and slowest wave per bounce in units
// of 10 ns, to set against the launch durations of a profile -- what the distribution of work over the waves loses)
#if defined(RT_SOLO_STAMPS) && RT_SOLO_STAMPS == 3
// (-DRT_SOLO_STAMPS=3: begin and end of every wave on BOTH clocks: shader cycles / 100 MHz ticks = the clock the chip holds in this kernel)
#define RT_STAMP(var) const unsigned long long var = 0
#define RT_STAMP_ADD(acc, from, to)
#define RT_STAMP_NOW() __builtin_amdgcn_s_memtime()
#elif defined(RT_SOLO_STAMPS) && RT_SOLO_STAMPS == 2
#define RT_STAMP(var) const unsigned long long var = 0
#define RT_STAMP_ADD(acc, from, to)
#define RT_STAMP_NOW() __builtin_amdgcn_s_memrealtime()
#elif defined(RT_SOLO_STAMPS)
#define RT_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RT_STAMP_ADD(acc, from, to) acc += (to) - (from)
#define RT_STAMP_NOW() __builtin_amdgcn_s_memtime()
#else
#define RT_STAMP(var)
#define RT_STAMP_ADD(acc, from, to)
#endif

// ---- the keep bits of one (granule, chunk): up to 128 tiles = bits [tile_begin, tile_begin + n_tiles) of the granule's row, as a
// 128-bit mask (wave-uniform in the scan: scalar loads and scalar arithmetic).  The words behind the end of a row belong to the next
// rows (the buffer ends 16 rows behind the last granule): masked out.
struct Mask128 { unsigned long long lo, hi; };
__device__ __forceinline__ Mask128 m128_shr(Mask128 m, uint32_t s)
{
    Mask128 r;
    if (s >= 128u) { r.lo = 0ull; r.hi = 0ull; }
    else if (s >= 64u) { r.lo = m.hi >> (s - 64u); r.hi = 0ull; }
    else if (s == 0u) r = m;
    else { r.lo = (m.lo >> s) | (m.hi << (64u - s)); r.hi = m.hi >> s; }
    return r;
}
__device__ __forceinline__ Mask128 m128_low(uint32_t n)                 // the n lowest bits set, n <= 128
{
    Mask128 r;
    r.lo = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    r.hi = n >= 128u ? ~0ull : (n > 64u ? ((1ull << (n - 64u)) - 1ull) : 0ull);
    return r;
}
__device__ __forceinline__ Mask128 m128_and(Mask128 a, Mask128 b) { Mask128 r; r.lo = a.lo & b.lo; r.hi = a.hi & b.hi; return r; }
__device__ __forceinline__ bool m128_any(Mask128 m) { return (m.lo | m.hi) != 0ull; }
__device__ __forceinline__ uint32_t m128_popc(Mask128 m) { return (uint32_t)__builtin_popcountll(m.lo) + (uint32_t)__builtin_popcountll(m.hi); }
__device__ __forceinline__ uint32_t m128_ctz(Mask128 m)                 // index of the lowest set bit; 128: none
{
    return m.lo ? (uint32_t)__builtin_ctzll(m.lo) : (m.hi ? 64u + (uint32_t)__builtin_ctzll(m.hi) : 128u);
}
__device__ __forceinline__ uint32_t m128_cto(Mask128 m)                 // number of consecutive set bits from bit 0
{
    Mask128 n; n.lo = ~m.lo; n.hi = ~m.hi;
    return m128_ctz(n);
}
template <typename Words>
__device__ __forceinline__ Mask128 chunk_keep_bits(Words row, uint32_t tile_begin, uint32_t n_tiles)
{
    const Words w = row + (tile_begin >> 5);
    const uint32_t sh = tile_begin & 31u;
    const unsigned long long w01 = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32), w23 = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32), w4 = w[4];
    Mask128 m;
    m.lo = sh ? (w01 >> sh) | (w23 << (64u - sh)) : w01;
    m.hi = sh ? (w23 >> sh) | (w4 << (64u - sh)) : w23;
    return m128_and(m, m128_low(n_tiles));
}

// ---- culled bounces: the work items of a dynamic scan launch.  For every chunk of `chunk_quads` quads the granules whose keep bits are
// not all clear, compacted (one atomic per 64 granules; the order inside a chunk's list is whatever the atomics made it: it only
// decides which wave scans what).  Granules with nothing to scan never become an item: with 95 % of the camera rays' tests culled
// two thirds of the (granule, chunk) pairs of C2 are empty.  grid = (ceil(granules / 256), chunks).
template <bool kCount>
__global__ void __launch_bounds__(256) cull_items_kernel(WaveBuffers wb, uint32_t bounce, uint32_t chunk_quads, uint32_t n_quads, Counters *__restrict__ counters)
{
    const uint32_t n_rays = wb.counts[bounce], n_gran = (n_rays + 127u) >> 7;
    const uint32_t c = blockIdx.y, q_begin = c * chunk_quads, q_end = min(q_begin + chunk_quads, n_quads);
    const uint32_t lane = threadIdx.x & 63u;
    if (q_begin >= q_end) return;
    const uint32_t n_tiles = (q_end - q_begin) * kMfQuadTiles;
    unsigned long long culled = 0;
    for (uint32_t g0 = blockIdx.x * 256u + (threadIdx.x & ~63u); g0 < n_gran; g0 += gridDim.x * 256u) {      // (wave-uniform bound)
        const uint32_t g = g0 + lane;
        bool any = false;
        if (g < n_gran) {
            const Mask128 bits = chunk_keep_bits(wb.keep + (size_t)g * wb.keep_words, q_begin * kMfQuadTiles, n_tiles);
            any = m128_any(bits);
            if (kCount) culled += (unsigned long long)(n_tiles - m128_popc(bits)) * kMfTileTris * min(128u, n_rays - g * 128u);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(any);
        if (m == 0ull) continue;
        uint32_t at = 0u;
        if (lane == 0u) at = atomicAdd(wb.item_counts + c, (uint32_t)__popcll(m));
        at = __builtin_amdgcn_readfirstlane(at) + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (any) store_through(wb.items + ((size_t)c * wb.items_stride + at), g);
    }
    if (kCount && culled) atomicAdd(&counters->culled_tests, culled);
}

// ---- culled bounces, planned work distribution: what every (granule, chunk) item will cost, as an inclusive prefix sum over the
// granules of each chunk (plan_prefix[chunk * stride + granule]) and the chunks' totals.  Cost, in quarters of a tile: 4 x (tiles the
// granule keeps in the chunk + ~6 tiles' worth for the item's set-up: ray fetch, four ray set-ups, pipeline fill and drain) when it keeps
// any, 1 when it keeps none (reading its keep bits and stepping on is not free: with every empty item at cost 0 a wave of the
// camera-ray bounce, 99.7 % culled, could be handed thousands of them in a row -- 5 ms for a launch that takes 0.07).  The scan
// cuts the line of all items (chunk-major) into one equal-cost interval per block and each block's part of a chunk into one per wave:
// with culling, items differ between nothing and 128 tiles, and a launch whose waves took their items in fixed turns lasted 1.5 x (bounce 1
// of C2) to 2.4 x (bounce 4) its mean wave.  One block per chunk; also counts the culled tests (option "counters").
constexpr uint32_t kPlanItemFixed = 6u;
template <bool kCount>
__global__ void __launch_bounds__(256) scan_plan_kernel(WaveBuffers wb, uint32_t bounce, uint32_t chunk_quads, uint32_t n_quads, Counters *__restrict__ counters)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry;
    const uint32_t n_rays = wb.counts[bounce], n_gran = (n_rays + 127u) >> 7;
    const uint32_t c = blockIdx.x, q_begin = c * chunk_quads, q_end = min(q_begin + chunk_quads, n_quads);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_tiles = q_begin < q_end ? (q_end - q_begin) * kMfQuadTiles : 0u;
    uint32_t *const P = wb.plan_prefix + (size_t)c * wb.plan_stride;
    unsigned long long culled = 0;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (uint32_t g0 = 0; g0 < n_gran; g0 += 256u) {
        const uint32_t g = g0 + threadIdx.x;
        uint32_t cost = 0u;
        if (g < n_gran && n_tiles) {
            const uint32_t kept = m128_popc(chunk_keep_bits(wb.keep + (size_t)g * wb.keep_words, q_begin * kMfQuadTiles, n_tiles));
            cost = kept ? 4u * (kept + kPlanItemFixed) : 1u;
            if (kCount) culled += (unsigned long long)(n_tiles - kept) * kMfTileTris * min(128u, n_rays - g * 128u);
        }
        uint32_t inc = cost;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint32_t base = carry;
        for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
        if (g < n_gran) store_through(P + g, base + inc);
        __syncthreads();
        if (threadIdx.x == 255u) carry = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) store_through(wb.plan_total + c, carry);
    if (kCount && culled) atomicAdd(&counters->culled_tests, culled);
}
// exclusive prefix of the chunks' totals (plan_base[0 .. n_chunks], 64 bits each): one wave
__global__ void __launch_bounds__(64) scan_plan_base_kernel(WaveBuffers wb, uint32_t n_chunks)
{
    const uint32_t lane = threadIdx.x;
    unsigned long long run = 0ull;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64u) {
        const uint32_t c = c0 + lane;
        const unsigned long long t = c < n_chunks ? (unsigned long long)wb.plan_total[c] : 0ull;
        unsigned long long inc = t;
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long u = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += u; }
        if (c < n_chunks) store_through(wb.plan_base + c, run + inc - t);
        run += __shfl(inc, 63);
    }
    if (lane == 0) store_through(wb.plan_base + n_chunks, run);
}

// W = waves per SIMD.  W = 1: the wave owns the register file (256 + AGPRs) and hides its own latencies (the rays of its next item
// travel while it scans).  W = 2: two waves share a SIMD, 256 registers each; a wave alone can issue one vector instruction per 4 cycles
// while the SIMD executes one per 2, so the second wave's VALU work runs beside the first one's and the stream becomes bound by
// the matrix pipe itself (tools/scan_stage_rate.hip: 33 cycles per product against 46); the second wave also fills the holes that
// scalar instructions, branches and the parking path tear into the stream.  The W = 2 form keeps nothing ray-related live across
// the tile loop (rays are re-read from the queue at every segment: the partner wave hides the round trip).
//
// Work distribution.  A work item = (granule of 128 rays, chunk of <= 32 quads whose A tiles sit in LDS).  A block starts on chunk
// blockIdx.x mod n_chunks.
//   * static (`dynamic` = 0): the block stays there; its waves and those of the other blocks of the chunk take the chunk's items in
//     turn (wave r of n takes items r, r + n, ...).  Nothing is synchronised, every wave knows its next item (whose record and rays
//     travel while it scans).  Best when the chunks cost about the same and there are many blocks per chunk: C2, C5.
//   * dynamic (`dynamic` = 1): the waves CLAIM the chunk's items from a counter in global memory (sched[chunk]), in batches, each at
//     its own pace; when the chunk has no unclaimed item left the block moves on, cyclically, to the next chunk that has, stages its
//     tiles and carries on there, until no chunk has.  Costs a round trip per batch and a block-wide wait per move; pays when
//     a launch has few blocks per chunk and the chunks differ (C4: 79 chunks for 256 CUs, camera rays culled for some chunks and not
//     for others: the static launch lasted 1.33x its mean wave; dynamic: 29.9 -> 33.9 Mpaths/s.  On C2 the static form is 14 %
//     faster: 8 chunks, 32 blocks each).
template <bool kCount, int W, int kDist>
__global__ void __launch_bounds__(256 * W) __attribute__((amdgpu_waves_per_eu(W, W)))
scan_solo_kernel(SceneView sc, WaveBuffers wb, MfView mf, uint32_t bounce, uint32_t chunk_quads, uint32_t n_chunks, Counters *__restrict__ counters, int debug_skip_exact, int cull)
{
    using Cfg = SoloCfg;
    constexpr bool dynamic = kDist == 1, planned = kDist == 2, hybrid = kDist == 3;  // (a template parameter: the static form carries none of the claiming state)
    constexpr int S = kSoloSets;
    constexpr uint32_t kWaves = 4u * (uint32_t)W, kThreads = 256u * (uint32_t)W;
    extern __shared__ uint4 lds_tiles[];                      // the chunk's A tiles, [quad][tile][panel][row]
    __shared__ uint2 lds_queue[kWaves * Cfg::kQueue];        // per-wave survivor queue, entry = (lane | tile in chunk << 8, 20-bit mask: bit 5 s + u = triangle u of the lane's half survived for ray set s)
    __shared__ uint32_t lds_pick;                             // the chunk the block scans next
    __shared__ uint32_t lds_ring[kWaves * 16u];              // per wave: the granules of its last 16 items (queue entries name their item by its turn in this ring)
#ifdef RT_SOLO_STAMPS
#if RT_SOLO_STAMPS == 3
    const unsigned long long tr_wave_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned long long ts_wave_begin = RT_STAMP_NOW();
    unsigned long long tt_stage = 0, tt_rays = 0, tt_group = 0, tt_steady = 0, tt_park = 0, tt_flush = 0, tt_iters = 0, tt_culled = 0, tt_culled_n = 0, tt_tiles = 0;
#endif
    // With one wave per SIMD the wave claims the whole register file of its SIMD (as two waves do: 2 x 256), so that nothing else runs
    // beside it and the compiler has AGPRs to spare instead of spilling.  History: DESIGN.md 5.2 -- builds whose one-wave scan used ~360
    // registers, kept prefetched rays in AGPRs and spilled four values lost 16 rays of a launch every few dozen runs when other
    // path-tracing pipelines ran on the device at the same time; the build with this line went 0 of 240 where its predecessor went
    // 19 of 120 on the same box.  Why is not established.
    if constexpr (W == 1) asm volatile("" ::: "a255");
    const uint32_t n_rays = wb.counts[bounce], n_gran = (n_rays + 127u) >> 7;
    const RayQueue qin = (bounce & 1u) ? wb.q[1] : wb.q[0];
    unsigned long long *best = (bounce & 1u) ? wb.best[1] : wb.best[0];
    // (`wave` through readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all lanes, and everything the item loop
    // derives from it -- item index, granule, keep bits, loop control -- would live in vector registers under EXEC masks)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), col = lane & 31, half = lane >> 5;
    // only quads that hold triangles; the padding rows inside the last one carry a -3e38 bias and never survive
    const uint32_t real_quads = min(mf.n_quads, (sc.n_tri_visits + kMfQuadTris - 1u) / kMfQuadTris);
    // survivors of this wave go to ITS region of the candidate buffer: no atomic, no round trip the single wave of a SIMD would wait for
    const uint32_t region = blockIdx.x * kWaves + (uint32_t)wave;
    uint2 *const cand = wb.cand + (size_t)region * wb.cand_region;
    unsigned long long appended = 0;                           // wave-uniform; pairs beyond the region's capacity are tested in place
    uint32_t *const sched = wb.sched + (size_t)bounce * wb.sched_stride;
    uint32_t opaque_zero = 0u;
    asm volatile("" : "+v"(opaque_zero));
    const uint32_t group_shift = (uint32_t)__builtin_ctz(mf.group_quads);
    constexpr uint32_t kQuadBytes = kMfQuadTiles * 1024;
    uint2 *queue = lds_queue + wave * Cfg::kQueue;
    unsigned long long c_cand_lane = 0;                        // kCount: surviving pairs this lane handed over
    unsigned long long c_culled = 0;                           // kCount: ray x triangle pairs the keep bits spared this wave (static launches)
    typedef const float __attribute__((address_space(4))) *ConstFloats;       // group records: uniform index => s_load
    typedef const uint32_t __attribute__((address_space(4))) *ConstWords;
    const ConstFloats groups_k = (ConstFloats)(uintptr_t)mf.groups;
    const uint32_t l_lane = (uint32_t)half * 32u + (uint32_t)col;              // this lane's row inside a tile (uint4 index)

    // rays of one granule as they sit in the queue: both lane halves hold the same ray
    float4 nxt_a[S], nxt_b[S];
    auto fetch_rays = [&](uint32_t g, float4 (&da)[S], float4 (&db)[S]) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t slot = g * 128u + (uint32_t)s * 32u + (uint32_t)col;
            da[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); db[s] = da[s];
            if (slot < n_rays) { da[s] = qin.a[slot]; db[s] = qin.b[slot]; }
        }
    };
    // the constants of a ray over the scan, from its queue record
    auto prepare_ray = [&](MfRay &r, const float4 &a, const float4 &b, bool valid) {
        r.valid = valid;
        r.o = mk(a.x, a.y, a.z); r.d = mk(a.w, b.x, b.y);
        r.wd = __builtin_amdgcn_sqrtf(dot3(r.d, r.d)) * 1.001f;               // (1 ulp square roots: these are bounds, inflated by 1.001)
        r.wod = (__builtin_amdgcn_sqrtf(dot3(r.o, r.o)) * 1.001f) * r.wd;
        const uint32_t dxy = pack_bf16(r.d.x, r.d.y);
        r.dyz = pack_bf16(r.d.y, r.d.z);
        r.dx_hi = dxy << 16;
        const f3 dl = mk(r.d.x - __uint_as_float(dxy << 16), r.d.y - __uint_as_float(dxy & 0xffff0000u), r.d.z - __uint_as_float(r.dyz & 0xffff0000u));
        r.tail = half ? pack_bf16(dl.y, dl.z) : pack_bf16(1.0f, dl.x);
    };

    // Which chunk a block starts on.  Consecutive blocks go to consecutive XCDs (8, each with its own L2): with chunk = block mod chunks
    // all blocks of a chunk would sit on one XCD and every XCD would pull every ray through its L2 (measured: 130 MB of HBM traffic per C2
    // launch against 58 MB).  So, when there are enough blocks, eight consecutive blocks share a chunk: the waves that take the same items
    // of different chunks then sit on the same XCD and share the rays in its L2.
    constexpr uint32_t kXcds = 8;
    const bool by_xcd = gridDim.x >= kXcds * n_chunks;
    const uint32_t c_first = by_xcd ? (blockIdx.x / kXcds) % n_chunks : blockIdx.x % n_chunks;
    // ---- planned: the block's interval (p_lo, p_hi] of the cost line of all items (chunk-major; plan_base = where each chunk starts).  An
    // unculled launch needs no plan: every item costs the same, chunk c starts at c x granules.
    typedef const unsigned long long __attribute__((address_space(4))) *ConstU64;
    const ConstU64 pbase_k = (ConstU64)(uintptr_t)wb.plan_base;
    auto chunk_base = [&](uint32_t cc) -> unsigned long long { return cull ? pbase_k[cc] : (unsigned long long)cc * n_gran; };
    unsigned long long p_lo = 0ull, p_hi = 0ull;
    uint32_t p_next = 0u;                                      // planned: the next chunk to look at
    if constexpr (planned) {
        const unsigned long long total = chunk_base(n_chunks);
        // (128-bit product: the total can exceed 2^32 and there can be thousands of blocks)
        p_lo = (unsigned long long)(((unsigned __int128)total * blockIdx.x) / gridDim.x);
        p_hi = (unsigned long long)(((unsigned __int128)total * (blockIdx.x + 1u)) / gridDim.x);
        uint32_t lo = 0u, hi = n_chunks;                       // last chunk that starts at or before p_lo
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (chunk_base(mid) <= p_lo) lo = mid; else hi = mid; }
        p_next = lo;
    }
    bool first_chunk = true;
    for (uint32_t c_from = c_first;;) {
        // ---- dynamic: the block picks the next chunk (cyclically from c_from) that has unclaimed items; none: done
        RT_STAMP(ts_pick);
        uint32_t c = c_from;
        if constexpr (planned) {
            // the next chunk that has a part of the block's interval; none: done
            while (p_next < n_chunks && chunk_base(p_next + 1u) <= p_lo) ++p_next;
            if (p_lo == p_hi || p_next >= n_chunks || chunk_base(p_next) >= p_hi) break;
            c = p_next++;
            if (!first_chunk) __syncthreads();                 // every wave is done with the tiles of the previous chunk
            first_chunk = false;
        }
        if (dynamic) {
            __syncthreads();                                   // every wave is done with the tiles (and the pick) of the previous chunk
            if (wave == 0) {
                uint32_t dist = 0xFFFFFFFFu;
                for (uint32_t cc = (uint32_t)lane; cc < n_chunks; cc += 64u) {
                    const uint32_t have = cull ? wb.item_counts[cc] : n_gran;      // (this loop runs in dynamic launches only)
                    if (__hip_atomic_load(sched + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < have) dist = min(dist, (cc + n_chunks - c_from) % n_chunks);
                }
                for (int off = 32; off > 0; off >>= 1) dist = min(dist, (uint32_t)__shfl_xor((int)dist, off));
                if (lane == 0) lds_pick = dist == 0xFFFFFFFFu ? 0xFFFFFFFFu : (c_from + dist) % n_chunks;
            }
            __syncthreads();
            c = __builtin_amdgcn_readfirstlane(lds_pick);            // (uniform, and the compiler should know)
            if (c == 0xFFFFFFFFu) break;
        }
        c_from = (c + 1u) % n_chunks;
        const uint32_t q_begin = c * chunk_quads, q_end = min(q_begin + chunk_quads, real_quads);
        const uint32_t v_chunk_begin = q_begin * kMfQuadTris, v_chunk_end = min(q_end * (uint32_t)kMfQuadTris, sc.n_tri_visits);
        const uint32_t n_tiles = (q_end - q_begin) * kMfQuadTiles, tile_begin = q_begin * kMfQuadTiles;
        const Mask128 keep_all = m128_low(n_tiles);
        {
            const uint32_t n16 = (q_end - q_begin) * (kQuadBytes / 16u);
            const uint4 *src = mf.A + (size_t)q_begin * (kQuadBytes / 16u);
            // eight loads in flight per thread (a plain copy loop waits for every load: 32 exposed L2 round trips per staging)
            for (uint32_t i0 = threadIdx.x; i0 < n16; i0 += 8u * kThreads) {
                uint4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const uint32_t i = i0 + (uint32_t)k * kThreads; v[k] = src[min(i, n16 - 1u)]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) { const uint32_t i = i0 + (uint32_t)k * kThreads; if (i < n16) lds_tiles[i] = v[k]; }
            }
            __syncthreads();
        }
        RT_STAMP(ts_staged);
        RT_STAMP_ADD(tt_stage, ts_pick, ts_staged);

        // ---- the wave's items of this chunk.  Static: item (rank of the wave among the waves that start on this chunk) + i x (their
        // number).  Dynamic: claimed from the chunk's counter in guided batches -- a quarter of an even share of what is left, at most
        // 16 items, at least one (a single address takes some ten million atomics per second: one claim per item and wave made the
        // claims queue up behind each other, ~5 us each); the wave waits for a claim, once per batch.
        uint32_t *const counter = sched + c;
        const uint32_t n_items = __builtin_amdgcn_readfirstlane((cull && dynamic) ? wb.item_counts[c] : n_gran);
        const ConstWords items_k = (ConstWords)(uintptr_t)(wb.items + (size_t)c * wb.items_stride);
        // blocks that start on this chunk, and this block's rank among them
        uint32_t blocks_here, rank_here;
        if (by_xcd) {
            const uint32_t rest = gridDim.x % (kXcds * n_chunks);
            blocks_here = (gridDim.x / (kXcds * n_chunks)) * kXcds + min(kXcds, rest > kXcds * c ? rest - kXcds * c : 0u);
            rank_here = (blockIdx.x / (kXcds * n_chunks)) * kXcds + blockIdx.x % kXcds;
        } else { blocks_here = (gridDim.x - c + n_chunks - 1u) / n_chunks; rank_here = blockIdx.x / n_chunks; }
        constexpr uint32_t kNone = 0xFFFFFFFFu;
        // (the address is made to look divergent: for a uniform one LLVM's atomic optimizer rewrites the operation into its wave-aggregated
        // form, which is no faster here and longer)
        // hybrid: two of three granules (hybrid_div - 1 of hybrid_div) in fixed turns (nothing to wait for, every wave knows its next item), every third -- spread evenly
        // over the queue, so that the tail looks like the rest -- claimed from the chunk's counter when the wave is through with its
        // turns: with culling an item costs anything between nothing and 128 tiles, and fixed turns alone left the launch waiting for
        // its unluckiest wave (1.5 x the mean wave on bounce 1 of C2, 2.4 x on bounce 4)
        const uint32_t hdiv = max(wb.hybrid_div, 1u), hturn = max(hdiv - 1u, 1u);      // hybrid: every hdiv-th granule is claimed (1: all of them)
        const uint32_t n_tail = hybrid ? n_gran / hdiv : 0u, n_turns = n_gran - n_tail;
        const uint32_t n_claimable = hybrid ? n_tail : n_items;
        auto claim = [&](uint32_t seen, uint32_t &lo, uint32_t &end) {
            const uint32_t rem = n_claimable > seen ? n_claimable - seen : 0u;
            const uint32_t n = hybrid ? min(max(rem / (2u * max(blocks_here, 1u) * kWaves), 1u), 4u) : min(max(rem / (4u * max(blocks_here, 1u) * kWaves), 1u), 16u);
            uint32_t v = 0u;
            if (lane == 0) v = atomicAdd(counter + opaque_zero, n);
            const uint32_t at = __builtin_amdgcn_readfirstlane(v);
            lo = min(at, n_claimable); end = min(at + n, n_claimable);
        };
        bool tail = false;                                     // hybrid: the wave has finished its turns and claims
        // item k of the chunk: the granule and which of the chunk's tiles its rays cannot be rejected for (bit t: tile t must be
        // scanned).  Culled bounces: dynamic launches take the granule from the compacted list of cull_items_kernel; the keep bits come
        // from the granule's row (chunk_keep_bits), through the scalar cache.
        const ConstWords keep_k = (ConstWords)(uintptr_t)wb.keep;
        auto item_of = [&](uint32_t k, uint32_t &g, Mask128 &bits) {
            g = hybrid ? (tail ? hdiv * k + (hdiv - 1u) : (k / hturn) * hdiv + k % hturn) : k; bits = keep_all;
            if (cull) {
                if (dynamic) g = items_k[k];
                bits = chunk_keep_bits(keep_k + (size_t)g * wb.keep_words, tile_begin, n_tiles);
            }
        };
        uint32_t k, hi, step;
        if (dynamic) { step = 1u; claim(0u, k, hi); }
        else if (planned) {
            // the block's part (a, e] of this chunk's cost, an eighth (or a quarter) of it per wave, and the granules whose prefix
            // sum falls into the wave's part: consecutive granules, no claim, no turn
            const unsigned long long cb = chunk_base(c), ce = chunk_base(c + 1u);
            const uint32_t a = (uint32_t)((p_lo > cb ? p_lo : cb) - cb), e = (uint32_t)((p_hi < ce ? p_hi : ce) - cb);
            const uint32_t w_lo = a + (uint32_t)(((unsigned long long)(e - a) * (uint32_t)wave) / kWaves), w_hi = a + (uint32_t)(((unsigned long long)(e - a) * ((uint32_t)wave + 1u)) / kWaves);
            const ConstWords P = (ConstWords)(uintptr_t)(wb.plan_prefix + (size_t)c * wb.plan_stride);
            auto first_above = [&](uint32_t x) -> uint32_t {           // first granule whose inclusive prefix sum exceeds x
                if (!cull) return min(x, n_gran);                        // (every item costs 1: the prefix of granule g is g + 1)
                uint32_t lo = 0u, up = n_gran;
                while (lo < up) { const uint32_t mid = (lo + up) >> 1; if (P[mid] > x) up = mid; else lo = mid + 1u; }
                return lo;
            };
            step = 1u; k = first_above(w_lo); hi = first_above(w_hi);
        }
        else {
            step = blocks_here * kWaves; k = rank_here * kWaves + (uint32_t)wave; hi = hybrid ? n_turns : n_items;
            if (hybrid && k >= hi) { tail = true; step = 1u; claim(0u, k, hi); }
        }
        uint32_t rec_k = kNone, rec_g = 0u;                    // the record of an item read ahead
        Mask128 rec_keep = {0ull, 0ull};
        uint32_t ray_k = kNone;                                // W = 1: the item whose rays are in (or on their way to) nxt_a / nxt_b
        auto advance = [&]() {
            k += step;
            if (dynamic && k >= hi) claim(hi, k, hi);          // (dynamic: the batch is used up)
            if (hybrid && k >= hi) { if (tail) claim(hi, k, hi); else { tail = true; step = 1u; claim(0u, k, hi); } }
        };
        // The wave's survivor queue lives across its items of this chunk: an entry names its item by the item's turn in a ring of 16
        // granules, and the queue is handed over when it is full, when the ring is about to wrap, and behind the chunk's last item --
        // not behind every item (that was 7 % of a wave's time on bounce 1 of C2 and 20-40 % on the late bounces, whose items keep a
        // handful of tiles each).
        uint32_t qn = 0, seq = 0, seq_flushed = 0;              // wave-uniform
        uint32_t *const ring = lds_ring + wave * 16;
        auto flush = [&]() {
            // Every (queue entry, ray set) with a non-empty 5-bit mask becomes one record (queue slot of the ray, storage position of
            // the lane's first triangle << 5 | mask), appended densely to the wave's region (ballot + prefix count per ray set), fire
            // and forget; the narrow phase expands the masks and maps storage position -> visit index.  What does not fit gets its
            // exact tests right here, so the result never depends on the buffer size.  `appended` is 64 bits wide: a degenerate
            // scene (NaN bounds: every pair survives) can exceed 2^32 records per wave.
            for (uint32_t i0 = 0; i0 < qn; i0 += 64u) {
                const uint32_t i = i0 + (uint32_t)lane;
                const uint2 e = i < qn ? queue[i] : make_uint2(0u, 0u);
                const uint32_t ln = e.x & 63u, pos5 = v_chunk_begin + ((e.x >> 8) & 127u) * kMfTileTris + 5u * (ln >> 5);
                const uint32_t slot0 = ring[(e.x >> 16) & 15u] * 128u;
                if (kCount) c_cand_lane += (unsigned long long)__popc(e.y);
                const uint32_t bits = (debug_skip_exact == 0 || debug_skip_exact == 3) ? e.y : 0u;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    uint32_t um = (bits >> (5 * s)) & 31u;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(um != 0u);
                    const unsigned long long at = appended + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    const uint32_t slot = slot0 + (uint32_t)(s * 32) + (ln & 31u);
                    if (um != 0u) {
                        if (at < (unsigned long long)wb.cand_region) store_through(reinterpret_cast<unsigned long long *>(cand + at), (unsigned long long)slot | ((unsigned long long)((pos5 << 5) | um) << 32));
                        else
                            while (um) {
                                const uint32_t pos = pos5 + (uint32_t)__builtin_ctz(um);
                                um &= um - 1u;
                                if (pos < v_chunk_end) exact_and_merge_at(mf, qin, best, slot, pos);
                            }
                    }
                    appended += (unsigned long long)__popcll(m);
                }
            }
            qn = 0; seq_flushed = seq;
        };
        while (k < hi) {
            RT_STAMP(ts_iter);
            uint32_t g; Mask128 keep;
            if (rec_k == k) { g = rec_g; keep = rec_keep; } else item_of(k, g, keep);
            const uint32_t wave_slot0 = g * 128u;
            // the item after this one, if it is known already: its record (and with one wave per SIMD its rays) travel during the scan
            const uint32_t succ = k + step < hi ? k + step : kNone;
            if (succ != kNone) { item_of(succ, rec_g, rec_keep); rec_k = succ; }
            if (kCount && cull && !dynamic && !planned && lane == 0)          // (dynamic launches count these in cull_items_kernel, planned ones in scan_plan_kernel)
                c_culled += (unsigned long long)(n_tiles - m128_popc(keep)) * kMfTileTris * min(128u, n_rays - wave_slot0);
            if (!m128_any(keep)) {                             // (static, culled bounce) nothing of this chunk can be hit by this granule's rays
                if constexpr (W == 1) { if (succ != kNone && m128_any(rec_keep)) { fetch_rays(rec_g, nxt_a, nxt_b); ray_k = succ; } }
                advance();
                continue;
            }
            if (seq - seq_flushed >= 15u) flush();              // (the ring is about to wrap)
            if (lane == 0) ring[seq & 15u] = g;
            const uint32_t item_tag = (seq & 15u) << 16;
            ++seq;
            MfRay ray[S];
            if constexpr (W == 1) {
                if (ray_k != k) fetch_rays(g, nxt_a, nxt_b);
#pragma unroll
                for (int s = 0; s < S; ++s) prepare_ray(ray[s], nxt_a[s], nxt_b[s], wave_slot0 + (uint32_t)s * 32u + (uint32_t)col < n_rays);
                // the next item's rays travel while this one is scanned (one wave per SIMD: nothing else would hide the round trip)
                if (succ != kNone && m128_any(rec_keep)) { fetch_rays(rec_g, nxt_a, nxt_b); ray_k = succ; }
            }

            u32x4 B0, B1, B2, B3;                               // B operands (K layout: see MfView) and thresholds of the four ray sets
            float th0, th1, th2, th3;
            f32x16 X0, X1, X2, X3, Y0, Y1, Y2, Y3;
            f32x6 MA0, MA1, MA2, MA3, MB0, MB1, MB2, MB3;        // [0..4]: minima of the five triangles of the lane's half, [5]: their maximum
            unsigned long long any_a, any_b;                    // lanes with a survivor in the tile examined into MA / MB
            // rare path: some lane has a survivor in `tile` (index inside the chunk).  Branch-free: every lane builds the 20-bit mask of
            // its surviving (ray set, triangle) pairs from the minima the stage left behind, lanes with a non-empty mask append ONE
            // entry to the wave's LDS queue (the others write to a scratch entry).  (The first form of this path tested set by set and
            // triangle by triangle with a scalar branch each: ~40 cycles per branch, 350-400 per parking event, 17 % of a bounce.)  A
            // finite threshold means finite operands and edge values below 2^7 * 1e30 in magnitude, hence finite minima; a NaN
            // threshold passes all.
            auto append = [&](uint32_t tile, uint32_t mask) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(mask != 0u);
                uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                asm volatile("" : "+v"(pre));                   // (keeps the prefix count out of a branch on "any lane active")
                queue[mask ? qn + pre : Cfg::kQueue - 1u] = make_uint2((uint32_t)lane | (tile << 8) | item_tag, mask);
                qn += (uint32_t)__popcll(m);
                if (qn >= Cfg::kDrain) flush();
            };
            auto park_a = [&](uint32_t tile) { RT_STAMP(ts_p0); uint32_t mask; RT_MASK_A(mask); append(tile, mask); RT_STAMP(ts_p1); RT_STAMP_ADD(tt_park, ts_p0, ts_p1); };
            auto park_b = [&](uint32_t tile) { RT_STAMP(ts_p0); uint32_t mask; RT_MASK_B(mask); append(tile, mask); RT_STAMP(ts_p1); RT_STAMP_ADD(tt_park, ts_p0, ts_p1); };
            RT_STAMP(ts_rays);
            RT_STAMP_ADD(tt_rays, ts_iter, ts_rays);

            // segments = runs of tiles that share one group (local origin + bounds); a chunk may start or end inside a group
            for (uint32_t ts0 = 0; ts0 < n_tiles;) {
                RT_STAMP(ts_g0);
                const uint32_t q = q_begin + ts0 / kMfQuadTiles;
                const uint32_t grp = q >> group_shift;
                const uint32_t ts1 = min(n_tiles, (((grp + 1u) << group_shift) - q_begin) * kMfQuadTiles);
                // tiles of this segment the wave still has to scan (bit i: tile ts0 + i); a segment that is culled altogether costs
                // nothing, not even its setup
                Mask128 seg = m128_and(m128_shr(keep, ts0), m128_low(ts1 - ts0));
                if (!m128_any(seg)) { ts0 = ts1; continue; }
                if constexpr (W == 2) {                          // nothing ray-related stays live across the tile loop: re-read the queue here
                    fetch_rays(g, nxt_a, nxt_b);
#pragma unroll
                    for (int s = 0; s < S; ++s) prepare_ray(ray[s], nxt_a[s], nxt_b[s], wave_slot0 + (uint32_t)s * 32u + (uint32_t)col < n_rays);
                }
                const ConstFloats gp = groups_k + (size_t)grp * (sizeof(MfGroup) / 4);
                MfGroup G;
                G.cx = gp[0]; G.cy = gp[1]; G.cz = gp[2]; G.E = gp[3]; G.Ml = gp[4]; G.Pw = gp[5]; G.P = gp[6]; G.pad1 = 0.0f;
                u32x4 Bs[S]; float ths[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const MfRay &r = ray[s];
                    const f3 ol = r.o - mk(G.cx, G.cy, G.cz);
                    const f3 cvl = cross3(r.d, ol);
                    // v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence: these are bounds, inflated by 1.001
                    const float ncv = __builtin_amdgcn_sqrtf(dot3(cvl, cvl)) * 1.001f, no = __builtin_amdgcn_sqrtf(dot3(ol, ol)) * 1.001f;
                    const float margin = mf_margin(G, ncv, no, r);
                    // empty slot: nothing survives.  Margin not finite or so large that the bf16 products could overflow (bounds NaN for
                    // non-finite vertices, huge coordinates): NaN threshold, everything survives.
                    ths[s] = (!r.valid || debug_skip_exact == 2) ? __builtin_inff() : (margin < 1.0e30f ? -margin : __builtin_nanf(""));
                    Bs[s].x = pack_bf16(cvl.x, cvl.y);
                    Bs[s].y = (pack_bf16(cvl.z, 0.0f) & 0xffffu) | r.dx_hi;
                    Bs[s].z = r.dyz; Bs[s].w = r.tail;
                }
                B0 = Bs[0]; B1 = Bs[1]; B2 = Bs[2]; B3 = Bs[3];
                th0 = ths[0]; th1 = ths[1]; th2 = ths[2]; th3 = ths[3];
                typedef const uint4 __attribute__((address_space(3))) *LdsRow;
                const uint32_t n_kept = m128_popc(seg);
                if (n_kept == ts1 - ts0 && debug_skip_exact != 3) {     // (debug_skip_exact = 3: timing diagnostics, every segment through the list loop)
                    // ---- the whole segment: consecutive tiles [t0, t1), rows read two trips ahead by address increment
                    const uint32_t t0 = ts0, t1 = ts1;
                    // prologue: the products of the first tile.  The four rows behind the chunk's last tile are allocated (and never used).
                    const uint4 *row = lds_tiles + (size_t)t0 * 64u + l_lane;
                    auto tile_row = [&]() { const uint4 r = *row; row += 64; return u32x4{r.x, r.y, r.z, r.w}; };
                    u32x4 ap = tile_row(), ay = tile_row(), ax = tile_row(), by, bx;
                    uint32_t addr = (uint32_t)(uintptr_t)(LdsRow)row;                       // LDS byte address of tile t0 + 3's row
                    RT_PRODUCTS_X(ap);
                    // steady state: two stages per trip (tile t -> Y beside the examination of tile t-1, tile t+1 -> X beside the
                    // examination of tile t), so that X and Y swap roles without moves; two trips per loop iteration, so that the tile
                    // rows do, too
                    uint32_t t = t0 + 1u;
                    for (; t + 3u < t1; t += 4u) {
                        RT_TRIP_L(ay, ax, by, bx, addr);
                        if (__builtin_expect((any_a | any_b) != 0ull, 0)) {
                            if (any_a) park_a(t - 1u);
                            if (any_b) park_b(t);
                        }
                        RT_TRIP_L(by, bx, ay, ax, addr);
                        if (__builtin_expect((any_a | any_b) != 0ull, 0)) {
                            if (any_a) park_a(t + 1u);
                            if (any_b) park_b(t + 2u);
                        }
                    }
                    if (t + 1u < t1) {                                                       // two or three tiles left: one more trip
                        RT_TRIP_L(ay, ax, by, bx, addr);
                        if ((any_a | any_b) != 0ull) {
                            if (any_a) park_a(t - 1u);
                            if (any_b) park_b(t);
                        }
                        ay = by; t += 2u;
                    }
                    // epilogue: a last single stage if the tile count is even, then the examination of the last tile
                    if (t < t1) {
                        RT_STAGE_Y(ay);
                        if (any_a) park_a(t - 1u);
                        RT_EXAMINE_Y();
                        if (any_b) park_b(t);
                    } else {
                        RT_EXAMINE_X();
                        if (any_a) park_a(t - 1u);
                    }
                } else {
                    // ---- a culled segment: the same stream over the LIST of its kept tiles, in storage order; the pipeline fills and
                    // drains once per segment however the kept tiles are scattered (walking them run by run cost a prologue, an
                    // epilogue and a parking pass per run: 2.8 steady tiles for a run of one)
                    unsigned long long cur = seg.lo, nxt = seg.hi;
                    uint32_t base = ts0;                                                    // tile of bit 0 of `cur`
                    auto next_tile = [&]() -> uint32_t {                                    // (callers never ask for more tiles than are kept)
                        if (cur == 0ull) { cur = nxt; nxt = 0ull; base += 64u; }
                        const uint32_t bit = (uint32_t)__builtin_ctzll(cur);
                        cur &= ~(1ull << bit);
                        return base + bit;
                    };
                    const uint32_t lds_lane = (uint32_t)(uintptr_t)(LdsRow)(lds_tiles + l_lane);
                    auto row_addr = [&](uint32_t tile) { return lds_lane + tile * 1024u; };
                    auto load_row = [&](uint32_t tile) { const uint4 r = lds_tiles[(size_t)tile * 64u + l_lane]; return u32x4{r.x, r.y, r.z, r.w}; };
                    uint32_t left = n_kept;                                                 // tiles not yet handed to the matrix pipe
                    uint32_t t_x = next_tile(); --left;                                     // the tile whose products sit in X
                    u32x4 ap = load_row(t_x), ay, ax, by, bx;
                    uint32_t t_y = t_x, t_n = t_x;                                          // tiles of the rows in (ay | by), (ax | bx); dummies when none is left
                    if (left > 0u) t_y = next_tile();
                    if (left > 1u) t_n = next_tile();
                    ay = load_row(t_y); ax = load_row(t_n);
                    RT_PRODUCTS_X(ap);
                    // a trip: t_y -> Y beside the examination of X (t_x), t_n -> X beside the examination of Y (t_y), and the rows of the
                    // two tiles behind them on their way; two trips per iteration, so that the row registers swap roles without moves
#define RT_LIST_TRIP(AY, AX, NY, NX) do { \
                        left -= 2u; \
                        const uint32_t u_y = left > 0u ? next_tile() : t_x, u_n = left > 1u ? next_tile() : t_x; \
                        const uint32_t a_y = row_addr(u_y), a_n = row_addr(u_n); \
                        RT_TRIP_G(AY, AX, NY, NX, a_y, a_n); \
                        if (__builtin_expect((any_a | any_b) != 0ull, 0)) { \
                            if (any_a) park_a(t_x); \
                            if (any_b) park_b(t_y); \
                        } \
                        t_x = t_n; t_y = u_y; t_n = u_n; \
                    } while (0)
                    while (left >= 4u) { RT_LIST_TRIP(ay, ax, by, bx); RT_LIST_TRIP(by, bx, ay, ax); }
                    if (left >= 2u) { RT_LIST_TRIP(ay, ax, by, bx); ay = by; }
#undef RT_LIST_TRIP
                    if (left == 1u) {
                        RT_STAGE_Y(ay);
                        if (any_a) park_a(t_x);
                        RT_EXAMINE_Y();
                        if (any_b) park_b(t_y);
                    } else {
                        RT_EXAMINE_X();
                        if (any_a) park_a(t_x);
                    }
                }
                ts0 = ts1;
                RT_STAMP(ts_g2);
                RT_STAMP_ADD(tt_steady, ts_g0, ts_g2);
            }
            RT_STAMP(ts_f0);
            advance();
            RT_STAMP(ts_f1);
            RT_STAMP_ADD(tt_flush, ts_f0, ts_f1);
#ifdef RT_SOLO_STAMPS
            tt_iters++; tt_tiles += m128_popc(keep);          // (tiles the item actually scanned)
#endif
        }
        flush();                                               // what the wave's last items of this chunk left in its queue
        if (!dynamic && !planned) break;                       // static: a block stays with the chunk it started on
    }
#ifdef RT_SOLO_STAMPS
    if (lane == 0 && mf.dbg_log) {
        unsigned long long *d = reinterpret_cast<unsigned long long *>(mf.dbg_log) + 16ull * bounce;
        const unsigned long long ts_end = RT_STAMP_NOW();
#if RT_SOLO_STAMPS == 3
        const unsigned long long tr_end = __builtin_amdgcn_s_memrealtime();          // (both clocks read before the atomics below, which queue up for microseconds)
#endif
        atomicAdd(d + 0, ts_end - ts_wave_begin); atomicAdd(d + 1, tt_stage); atomicAdd(d + 2, tt_rays); atomicAdd(d + 3, tt_group);
        atomicAdd(d + 4, tt_steady); atomicAdd(d + 5, tt_park); atomicAdd(d + 6, tt_flush); atomicAdd(d + 7, tt_iters); atomicAdd(d + 8, 1ull);
        atomicAdd(d + 9, tt_tiles);
        if (blockIdx.x == 0 && wave == 0) d[15] = (unsigned long long)gridDim.x * kWaves;
        atomicMax(d + 10, ts_end - ts_wave_begin);                // slowest wave of this launch: summed per bounce by the narrow phase
        if (kDist == 0 && bounce < 16u) {                              // static launches: wave time per chunk (the block's only one)
            unsigned long long *pc = reinterpret_cast<unsigned long long *>(mf.dbg_log) + 2048ull + (16ull * bounce) * 64ull * 2ull + 2ull * (c_first % 512u);
            atomicAdd(pc, ts_end - ts_wave_begin); atomicAdd(pc + 1, 1ull);
        }
        atomicAdd(d + 11, tt_culled); atomicAdd(d + 12, tt_culled_n);
#if RT_SOLO_STAMPS == 3
        atomicAdd(d + 14, tr_end - tr_wave_begin);
#endif
    }
#endif
    if (lane == 0) store_through(wb.cand_counts + region, (uint32_t)(appended < (unsigned long long)wb.cand_region ? appended : (unsigned long long)wb.cand_region));
    if (lane == 0 && appended > (unsigned long long)*wb.cand_peak) atomicMax(wb.cand_peak, (uint32_t)(appended < 0xFFFFFFF0ull ? appended : 0xFFFFFFF0ull));   // (racy pre-check: only saves atomics)
    if (kCount) {
        if (c_cand_lane) atomicAdd(&counters->candidates, c_cand_lane);
        if (c_culled) atomicAdd(&counters->culled_tests, c_culled);
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters->tri_tests, (unsigned long long)n_rays * sc.n_tri_visits);
    }
}

// ---- narrow phase: one lane per record (ray, first storage position << 5 | 5-bit survivor mask): exact reference-order test
// (:243-249) of every marked triangle, atomicMin merge.  grid = (wave regions of the scan launch that preceded it, kNarrowSplit): the
// blocks of a region take its records in turns of 256 -- the kernel is a chain of four dependent round trips per record, and with one
// block per region its duration was that of the fullest region (three to four turns where the mean is under two).
constexpr uint32_t kNarrowSplit = 4;
__global__ void __launch_bounds__(256) narrow_phase_kernel(SceneView sc, WaveBuffers wb, MfView mf, uint32_t bounce, uint32_t n_regions)
{
#ifdef RT_SOLO_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && mf.dbg_log) {          // the slowest wave of the scan launch that just ended: summed per bounce
        unsigned long long *d = reinterpret_cast<unsigned long long *>(mf.dbg_log) + 16ull * bounce;
        d[13] += d[10]; d[10] = 0ull;
    }
#endif
    const RayQueue qin = (bounce & 1u) ? wb.q[1] : wb.q[0];
    unsigned long long *best = (bounce & 1u) ? wb.best[1] : wb.best[0];
    for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
        const uint2 *cand = wb.cand + (size_t)r * wb.cand_region;
        const uint32_t i0 = blockIdx.y * 256u + threadIdx.x;
        uint2 c = i0 < wb.cand_region ? cand[i0] : make_uint2(0u, 0u);      // (read together with the count: one round trip less)
        const uint32_t n = min(wb.cand_counts[r], wb.cand_region);
        for (uint32_t i = i0; i < n; i += 256u * gridDim.y) {
            if (i != i0) c = cand[i];
            uint32_t um = c.y & 31u;
            const uint32_t pos5 = c.y >> 5;
            while (um) {
                const uint32_t pos = pos5 + (uint32_t)__builtin_ctz(um);
                um &= um - 1u;
                if (pos < sc.n_tri_visits) exact_and_merge_at(mf, qin, best, c.x, pos);     // (padding rows behind the last triangle)
            }
        }
    }
}

}  // namespace rt
