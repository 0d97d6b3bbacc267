// rt_device.hpp -- device-side building blocks of the path tracer (gfx950 / CDNA4 only).
//
// Numerics contract: every float operation below is an IEEE-754 binary32
// operation in a fixed order; the translation unit is compiled with -ffp-contract=off and the only
// fused operations are the __builtin_fmaf calls written out here.  Division and sqrt are the
// correctly rounded forms (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).  The order
// is the one the reference shader has when executed on Mesa llvmpipe (the pinned oracle), so the
// images are designed to be bit-identical to that run; see tests/ for the measured agreement.
//
// Cited reference lines are in /root/reference/shaders/raytracer.glsl unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace rt {

constexpr float kInf = 1e5f;    // INF      (:5)
constexpr float kEps = 0.005f;  // EPSILON  (:4)
constexpr uint32_t kNoSphere = 0xFFFFFFFFu;

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
// dot(): z and y products summed first, then the x product (order of the pinned oracle)
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// normalize(): v * (1 / sqrt(dot(v, v)))
__device__ __forceinline__ f3 normalize3(f3 a)
{
    float inv = 1.0f / __builtin_sqrtf(dot3(a, a));
    return a * inv;
}

// ---- sin / cos: Cephes single-precision range reduction + polynomials, with exactly the fused
// multiply-adds of the pinned oracle.
template <bool kCos>
__device__ __forceinline__ float sincos_poly(float a)
{
    uint32_t ai = __float_as_uint(a);
    float x = __uint_as_float(ai & 0x7fffffffu);
    float ys = x * 1.27323954473516f;
    int32_t j = (int32_t)ys;
    int32_t jadd = j + 1;
    int32_t jj = jadd & ~1;
    float y = (float)jj;
    int32_t sel = kCos ? jj - 2 : jj;
    uint32_t sign = kCos ? ((4u & ~(uint32_t)sel) << 29) : ((ai ^ ((uint32_t)jadd << 29)) & 0x80000000u);
    bool use_sin_poly = ((sel & 2) == 0);
    float x1 = __builtin_fmaf(y, -0.78515625f, x);
    float x2 = __builtin_fmaf(y, -2.4187564849853515625e-4f, x1);
    float x3 = __builtin_fmaf(y, -3.77489497744594108e-8f, x2);
    float z = x3 * x3;
    float c = __builtin_fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = __builtin_fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    float s = __builtin_fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s = __builtin_fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = __builtin_fmaf(s, x3, x3);
    float r = use_sin_poly ? s : c;
    r = __uint_as_float(__float_as_uint(r) ^ sign);
    if (r < -1.0f) r = -1.0f;
    if (r > 1.0f) r = 1.0f;
    if ((ai & 0x7f800000u) == 0x7f800000u) r = __uint_as_float(0x7fc00000u);
    return r;
}
__device__ __forceinline__ float sin_rt(float a) { return sincos_poly<false>(a); }
__device__ __forceinline__ float cos_rt(float a) { return sincos_poly<true>(a); }

// ---- PCG4D hash RNG (:131-152).  All arithmetic is uint32 wrap-around.
struct Rng { uint32_t x, y, z, w; };

__device__ __forceinline__ void pcg4d(Rng &v)
{
    v.x = v.x * 1664525u + 1013904223u;
    v.y = v.y * 1664525u + 1013904223u;
    v.z = v.z * 1664525u + 1013904223u;
    v.w = v.w * 1664525u + 1013904223u;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
    v.x ^= v.x >> 16; v.y ^= v.y >> 16; v.z ^= v.z >> 16; v.w ^= v.w >> 16;
    v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
}
// rand() (:148-152): float(seed.x) / float(0xffffffffu); the divisor rounds to 2^32
__device__ __forceinline__ float rand01(Rng &r)
{
    pcg4d(r);
    return (float)r.x / 4294967296.0f;
}
// random_in_sphere (:154-162)
__device__ __forceinline__ f3 random_in_sphere(Rng &r)
{
    float z = rand01(r) * 2.0f + -1.0f;
    float a = 6.283185482025146484375f * rand01(r);
    float rr = __builtin_sqrtf(1.0f + -(z * z));
    return mk(rr * cos_rt(a), rr * sin_rt(a), z);
}

// ---- scene records -------------------------------------------------------------------------
// Reference GPU layouts, read through bounds-checked accessors: a read past
// the end of a buffer returns zeros, like llvmpipe's SSBO loads.
struct SphereRec { float cx, cy, cz, radius; int32_t material; int32_t pad[3]; };   // 32 B
struct MaterialRec { float ar, ag, ab, smoothness, er, eg, eb; uint32_t type; };    // 32 B

// Per-triangle records precomputed at upload (ray-independent terms of triangle_intersect
// :230-239, evaluated with the same operations the shader would use per ray).
struct alignas(16) TriEdges {   // 80 B: what the per-ray edge test reads
    float e0x, e0y, e0z;        // v1 - v0
    float e1x, e1y, e1z;        // v2 - v1
    float e2x, e2y, e2z;        // v0 - v2
    float m0x, m0y, m0z;        // cross(v1, v0)
    float m1x, m1y, m1z;        // cross(v2, v1)
    float m2x, m2y, m2z;        // cross(v0, v2)
    float bound_e, bound_m;     // >= max_k |e_k|, >= max_k |m_k|  (for the conservative filter)
};
struct alignas(16) TriPlane {   // 32 B: read only for candidates / the winning hit
    float nx, ny, nz;           // normalize(cross(v1 - v0, v2 - v0))
    float v0x, v0y, v0z;
    int32_t material;           // int(vertices[3v].w) (:353), -1 when not representable
    int32_t pad;
};

struct FrameParams {            // mirrors rtgl_frame_params (include/rtgl_amd.h)
    int32_t frames; uint32_t samples; uint32_t max_bounce; float time;
    float background[3]; int32_t reset_flag; int32_t use_envmap; int32_t use_dof; int32_t random;
    float cam_pos[3]; float cam_fov; float cam_aperture; float cam_focal;
    float cam_forward[3]; float cam_up[3]; float cam_right[3];
};

struct SceneView {
    const SphereRec *spheres; uint32_t n_spheres;
    const uint32_t *sphere_visits; uint32_t n_sphere_visits;   // flattened node walk (host side)
    const MaterialRec *materials; uint32_t n_materials;
    const TriEdges *tri_edges; const TriPlane *tri_planes; uint32_t n_tri_visits;
    const uint8_t *env; int32_t env_w, env_h, env_c, env_faces;
};

__device__ __forceinline__ MaterialRec load_material(const SceneView &sc, int32_t i)
{
    MaterialRec m;
    if (i >= 0 && (uint32_t)i < sc.n_materials) m = sc.materials[i];
    else { m.ar = m.ag = m.ab = m.smoothness = m.er = m.eg = m.eb = 0.0f; m.type = 0u; }
    return m;
}

// ---- ray / hit --------------------------------------------------------------------------------
struct Hit { float t; f3 point, normal; int32_t material; };

// sphere_intersect (:200-220)
__device__ __forceinline__ float sphere_intersect(f3 o, f3 d, f3 c, float radius)
{
    f3 op = c - o;
    float b = dot3(op, d);
    float det = (b * b - dot3(op, op)) + radius * radius;
    float sq = __builtin_sqrtf(det);
    float t1 = b - sq, t2 = b + sq;
    float t = (0.001f < t1) ? t1 : ((0.001f < t2) ? t2 : kInf);
    return (det < 0.0f) ? kInf : t;
}

// traverse (:272-329).  The walk over the node buffer does not depend on the ray (the AABB cull
// is compiled out in the reference, :288-292), so the host flattens it once per scene into
// `sphere_visits`; the per-ray work is the closest-hit scan in that order.
__device__ __forceinline__ bool sphere_pass(const SceneView &sc, f3 o, f3 d, Hit &hit)
{
    bool any = false;
    for (uint32_t k = 0; k < sc.n_sphere_visits; ++k) {
        uint32_t i = sc.sphere_visits[k];
        f3 c = mk(0.0f, 0.0f, 0.0f);
        float radius = 0.0f;
        int32_t material = 0;
        if (i != kNoSphere) {
            const SphereRec &s = sc.spheres[i];
            c = mk(s.cx, s.cy, s.cz); radius = s.radius; material = s.material;
        }
        float t = sphere_intersect(o, d, c, radius);
        if (kEps < t && t < hit.t) {
            hit.t = t;
            hit.point = o + d * t;
            f3 pc = hit.point - c;
            hit.normal = mk(pc.x / radius, pc.y / radius, pc.z / radius);
            hit.material = material;
            any = true;
        }
    }
    return any;
}

// Per-ray constants of the triangle pass.
struct TriRay {
    f3 o, d;        // ray
    f3 cv;          // cross(d, o)  (center_v :227)
    float ncv, nd;  // 2^-19 * (upper bounds of |cv|, |d|): scale of the filter margin
};
__device__ __forceinline__ TriRay make_tri_ray(f3 o, f3 d)
{
    TriRay r;
    r.o = o; r.d = d;
    r.cv = cross3(d, o);
    r.ncv = (__builtin_sqrtf(dot3(r.cv, r.cv)) * 1.0001f) * 1.9073486328125e-06f;
    r.nd = (__builtin_sqrtf(dot3(d, d)) * 1.0001f) * 1.9073486328125e-06f;
    return r;
}

// Conservative edge filter.  The reference accepts a triangle when, for all three edges,
//     dot(e_k, cv) + dot(m_k, d) > 0                                  (:243-245)
// evaluated with separately rounded products and sums.  F_k below is the same six-term sum as one
// fma chain; it differs from the reference's value by less than 9 * 2^-24 * sum|terms|, and
// sum|terms| <= |e_k||cv| + |m_k||d| <= bound_e*|cv| + bound_m*|d|.  A triangle is rejected only
// when some F_k <= -margin with margin = 2^-19 * that bound (3.5x the worst case), so every triangle
// the reference would accept survives; survivors are re-evaluated exactly by tri_exact().
// NaNs fail the `<=` and therefore survive to the exact test.
__device__ __forceinline__ bool tri_filter(const TriEdges &T, const TriRay &r)
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
    float margin = __builtin_fmaf(T.bound_e, r.ncv, T.bound_m * r.nd) + 1e-30f;
    float mn = __builtin_fminf(__builtin_fminf(f0, f1), f2);   // v_min3_f32: ignores NaN operands
    return !(mn <= -margin);
}

// Exact re-evaluation in the reference's operation order: -dot(e_k, cv) < dot(m_k, d) for k=0,1,2,
// then t = -dot(o - v0, n) / dot(d, n) (:248) with the `t < INF` filter (:249).
// Returns kInf when the triangle is not hit.
__device__ __forceinline__ float tri_exact(const TriEdges &T, const TriPlane &P, const TriRay &r)
{
    f3 e0 = mk(T.e0x, T.e0y, T.e0z), m0 = mk(T.m0x, T.m0y, T.m0z);
    if (!(-dot3(e0, r.cv) < dot3(m0, r.d))) return kInf;
    f3 e1 = mk(T.e1x, T.e1y, T.e1z), m1 = mk(T.m1x, T.m1y, T.m1z);
    if (!(-dot3(e1, r.cv) < dot3(m1, r.d))) return kInf;
    f3 e2 = mk(T.e2x, T.e2y, T.e2z), m2 = mk(T.m2x, T.m2y, T.m2z);
    if (!(-dot3(e2, r.cv) < dot3(m2, r.d))) return kInf;
    f3 n = mk(P.nx, P.ny, P.nz), v0 = mk(P.v0x, P.v0y, P.v0z);
    float q = dot3(r.d, n);
    float t = -dot3(r.o - v0, n) / q;
    return (t < kInf) ? t : kInf;
}

// ---- environment cube map (texture(u_envmap, dir), :442) ----------------------------------------
// Face selection and (s,t) per the OpenGL 4.3 cube map table; bilinear filter as the pinned oracle
// does it for 8-bit UNORM texels: fixed-point texel coordinates with 8 fractional bits (round to
// nearest even), clamp-to-edge inside the selected face (non-seamless), three 8-bit lerps rounded
// like pmulhrsw, result * float(1/255).
__device__ __forceinline__ int32_t lerp8(int32_t w, int32_t v0, int32_t v1)
{
    int32_t p = w * ((v1 - v0) * 128);
    int32_t r = (((p >> 14) + 1) >> 1) & 0xff;
    return (v0 + r) & 0xff;
}
__device__ __forceinline__ f3 env_lookup(const SceneView &sc, f3 dir)
{
    if (sc.env == nullptr || sc.env_faces < 6) return mk(0.0f, 0.0f, 0.0f);
    float ax = __builtin_fabsf(dir.x), ay = __builtin_fabsf(dir.y), az = __builtin_fabsf(dir.z);
    bool x_over_y = ax > ay;
    float mxy = x_over_y ? ax : ay;
    bool z_major = az >= mxy;
    bool sx = __float_as_uint(dir.x) >> 31, sy = __float_as_uint(dir.y) >> 31, sz = __float_as_uint(dir.z) >> 31;
    int face; float ma, fs, ft;
    if (z_major) { face = 4; ma = dir.z; fs = sz ? -dir.x : dir.x; ft = -dir.y; }
    else if (x_over_y) { face = 0; ma = dir.x; fs = sx ? dir.z : -dir.z; ft = -dir.y; }
    else { face = 2; ma = dir.y; fs = dir.x; ft = sy ? -dir.z : dir.z; }
    if (__float_as_uint(ma) >> 31) face += 1;
    float ima = 0.5f / __builtin_fabsf(ma);
    float s = fs * ima + 0.5f;
    float t = ft * ima + 0.5f;
    const int W = sc.env_w, H = sc.env_h, C = sc.env_c;
    float sf = (s * (float)W) * 256.0f, tf = (t * (float)H) * 256.0f;
    if (!(sf > -1e9f)) sf = -1e9f;
    if (sf > 1e9f) sf = 1e9f;
    if (!(tf > -1e9f)) tf = -1e9f;
    if (tf > 1e9f) tf = 1e9f;
    int32_t si = (int32_t)__builtin_rintf(sf) - 128, ti = (int32_t)__builtin_rintf(tf) - 128;
    int32_t s0 = si >> 8, sw = si & 0xff, t0 = ti >> 8, tw = ti & 0xff;
    int32_t s1 = s0 + 1, t1 = t0 + 1;
    s0 = min(max(s0, 0), W - 1); s1 = min(max(s1, 0), W - 1);
    t0 = min(max(t0, 0), H - 1); t1 = min(max(t1, 0), H - 1);
    const uint8_t *f = sc.env + (size_t)face * W * H * C;
    const uint8_t *p00 = f + ((size_t)t0 * W + s0) * C, *p01 = f + ((size_t)t0 * W + s1) * C;
    const uint8_t *p10 = f + ((size_t)t1 * W + s0) * C, *p11 = f + ((size_t)t1 * W + s1) * C;
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int32_t r0 = lerp8(sw, p00[c], p01[c]), r1 = lerp8(sw, p10[c], p11[c]);
        out[c] = (float)lerp8(tw, r0, r1) * 0.0039215688593685626983642578125f;
    }
    return mk(out[0], out[1], out[2]);
}

// ---- camera_ray (:168-198) with the NDC of main() (:540-552) -----------------------------------
__device__ __forceinline__ void camera_ray(const FrameParams &P, int px, int py, int W, int H, Rng &rng, f3 &origin, f3 &dir)
{
    float fw = (float)W, fh = (float)H;
    float aspect = fh / fw;
    float ndx = ((float)px / fw) * 2.0f + -1.0f;
    float ndy = ((float)py / fh) * 2.0f + -1.0f;
    f3 pos = mk(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    f3 fwd = mk(P.cam_forward[0], P.cam_forward[1], P.cam_forward[2]);
    f3 up = mk(P.cam_up[0], P.cam_up[1], P.cam_up[2]);
    f3 right = mk(P.cam_right[0], P.cam_right[1], P.cam_right[2]);
    float half = P.cam_fov / 2.0f;
    float tn = sin_rt(half) / cos_rt(half);
    f3 target = pos + fwd;
    f3 vp = target + (right * (2.0f * tn)) * ndx;
    vp = vp + (up * (2.0f * (tn * aspect))) * ndy;
    dir = normalize3(vp - pos);
    origin = pos;
    if (P.use_dof) {
        f3 s = random_in_sphere(rng);
        f3 jitter = s * P.cam_aperture;
        origin = pos + jitter;
        f3 focal_point = pos + dir * P.cam_focal;
        dir = normalize3(focal_point - origin);
    }
}

// ---- material response of one bounce (:447-525) ------------------------------------------------
// In: the chosen hit, the incoming direction d, throughput.  Out: new origin/direction/throughput,
// radiance increment already added.  Returns false when the path ends here (total internal
// reflection, :490-494).
__device__ __forceinline__ bool shade_hit(const SceneView &sc, const Hit &h, Rng &rng, f3 &o, f3 &d, f3 &thr, f3 &radiance)
{
    MaterialRec m = load_material(sc, h.material);
    f3 albedo = mk(m.ar, m.ag, m.ab), emission = mk(m.er, m.eg, m.eb);
    f3 n = h.normal;
    bool inside = (-(d.z * n.z) + -(d.y * n.y)) < d.x * n.x;   // dot(-d, n) < 0 (:455)
    o = h.point;
    if (m.type == 0u) {                       // diffuse (:462-466)
        d = normalize3(n + random_in_sphere(rng));
        thr = thr * albedo;
    } else if (m.type == 1u) {                // specular (:467-474): direction left un-normalised
        f3 diffuse = normalize3(n + random_in_sphere(rng));
        float dn2 = dot3(d, n) * 2.0f;
        f3 refl = d - n * dn2;
        d = mk(diffuse.x + m.smoothness * (refl.x - diffuse.x),
               diffuse.y + m.smoothness * (refl.y - diffuse.y),
               diffuse.z + m.smoothness * (refl.z - diffuse.z));
        thr = thr * albedo;
    } else if (m.type == 2u) {                // transmissive (:475-523)
        f3 nl = inside ? neg(n) : n;
        float nnt = inside ? 1.4f : 0.714285731315612793f;
        float ct = dot3(d, nl);
        float omc = 1.0f + -(ct * ct);
        if (1.0f < (nnt * nnt) * omc) return false;   // total internal reflection ends the path
        float k = 1.0f + -(nnt * (nnt * omc));
        f3 T = mk(0.0f, 0.0f, 0.0f);
        if (!(k < 0.0f)) {
            float f = nnt * ct + __builtin_sqrtf(k);
            T = d * nnt - nl * f;
        }
        float c2 = dot3(T, n);
        float tmp = inside ? c2 : -ct;
        float c = 1.0f + -tmp;
        float cc = c * c;
        float X = (cc * 0.97222220897674560546875f) * (cc * c);
        float Re = 0.02777777425944805145263671875f + X;
        float halfRe = 0.5f * Re;
        float Pp = 0.25f + halfRe;
        float RP = Re / Pp;
        float TP = (0.97222220897674560546875f + -X) / (0.75f + -halfRe);
        if (rand01(rng) < Pp) {
            thr = thr * (albedo * RP);
            float dn2 = dot3(d, n) * 2.0f;
            d = d - n * dn2;
        } else {
            thr = thr * (albedo * TP);
            d = T;
        }
    }
    radiance = radiance + emission * thr;     // :525
    return true;
}

// ---- image addressing, counters, running mean ---------------------------------------------------
struct ImageView {
    float4 *pixels;     // local RGBA32F rows
    int width, height;  // full image size
    int disp_w, disp_h; // dispatch footprint: width/8*8, height/8*8 (src/renderer.cpp:132-133)
    int local_rows;
    int rank, world, strip_rows;   // row-strip ownership (world == 1: everything)
};

__device__ __forceinline__ int local_to_global_row(const ImageView &im, int lr)
{
    if (im.world == 1) return lr;
    int ls = lr / im.strip_rows, within = lr - ls * im.strip_rows;
    return (ls * im.world + im.rank) * im.strip_rows + within;
}

struct Counters { unsigned long long paths, segments, tri_tests, candidates, env_lookups, culled_tests; };

// running mean of main() (:561-568)
__device__ __forceinline__ float4 accumulate_pixel(const FrameParams &P, f3 color, f3 prev)
{
    float ns = (float)P.samples;
    color = mk(color.x / ns, color.y / ns, color.z / ns);
    float fr = (float)P.frames, fr1 = (float)(P.frames + 1);
    return make_float4((color.x + prev.x * fr) / fr1, (color.y + prev.y * fr) / fr1, (color.z + prev.z * fr) / fr1, 1.0f);
}

}  // namespace rt
