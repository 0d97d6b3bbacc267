/*
 * pathtrace_oracle.c -- CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product (raytracer.glsl_amd/) never links it and has no CPU fallback.
 *
 * What it restates: the compute shader shaders/raytracer.glsl of the reference (cited below by
 * line) plus the host marshalling in src/renderer.cpp:89-134 that feeds it.  It is written from
 * the shader's *semantics*, in scalar C, and pinned two ways:
 *   (1) against golden images produced by running the reference shader itself on Mesa llvmpipe
 *       in the build container (tests/golden/, generator tests/golden/make_golden.py);
 *   (2) operation order follows what llvmpipe actually executes for that shader (Mesa 23.2.1
 *       NIR, dumped with ST_DEBUG=nir and read as study): 3-term dot products are summed
 *       (z*z' + y*y') + x*x', normalize is v * (1/sqrt(dot)), a/b is a true IEEE division,
 *       mix(a,b,t) = a + t*(b-a), sin/cos are the Cephes single-precision polynomials with the
 *       fused multiply-adds llvmpipe emits on an FMA-capable host, and the cube map is filtered
 *       in 8-bit fixed point.  With these choices the oracle is designed to be bit-identical to
 *       the llvmpipe run; the tests state the measured agreement.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -mfma (see oracle/Makefile).  -ffp-contract=off
 * is required: every fused operation below is spelled fmaf() explicitly.
 *
 * GL-undefined behaviour made explicit (same decisions as the HIP kernels):
 *   - reads past the end of any scene buffer return zeros (llvmpipe's bounds-checked SSBO loads);
 *   - an empty node buffer means "no spheres" (SURVEY A.9 item 14); the node walk is capped at
 *     65535 pops (llvmpipe's own loop cap) so cyclic child links terminate;
 *   - triangle ranges are clamped to the triangles that exist in the vertex buffer;
 *   - the first frame averages against whatever the caller put in the image (tests zero it).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_INF 1e5f      /* shaders/raytracer.glsl:5 */
#define ORACLE_EPS 0.005f    /* :4 */
#define ORACLE_INVALID 0xFFFFFFFFu /* :7 */
#define ORACLE_STACK 5       /* :98 */
#define ORACLE_NODE_POP_CAP 65535

typedef struct { float x, y, z; } v3;

/* mirrors the 17 uniforms of shaders/raytracer.glsl:62-81 as uploaded by src/renderer.cpp:96-123 */
typedef struct oracle_params {
    int32_t frames;
    uint32_t samples;
    uint32_t max_bounce;
    float time;
    float background[3];
    int32_t reset_flag;
    int32_t use_envmap;
    int32_t use_dof;
    int32_t random;
    float camera_position[3];
    float camera_fov; /* radians (src/renderer.cpp:116) */
    float camera_aperture;
    float camera_focal_length;
    float camera_forward[3];
    float camera_up[3];
    float camera_right[3];
} oracle_params;

/* scene buffers in the layouts of SURVEY Appendix C (shaders/raytracer.glsl:11-60) */
typedef struct oracle_scene {
    const uint8_t *spheres;   uint32_t n_spheres;   /* stride 32 */
    const uint8_t *materials; uint32_t n_materials; /* stride 32 */
    const uint8_t *meshes;    uint32_t n_meshes;    /* stride 16 */
    const uint8_t *vertices;  uint32_t n_vertices;  /* vec4 count, stride 16 */
    const uint8_t *nodes;     uint32_t n_nodes;     /* stride 48 */
    const uint8_t *env;       /* faces +X,-X,+Y,-Y,+Z,-Z, 8-bit, tightly packed */
    int32_t env_w, env_h, env_channels, env_faces;
} oracle_scene;

typedef struct oracle_counters {
    uint64_t paths, segments, sphere_tests, triangle_tests, rand_calls, env_lookups;
} oracle_counters;

/* ---------------------------------------------------------------- scalar helpers */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* dot(): llvmpipe sums the z and y products first, then adds the x product */
static inline float vdot(v3 a, v3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
static inline v3 vcross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* normalize(): v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt, both IEEE */
static inline v3 vnormalize(v3 a)
{
    float inv = 1.0f / sqrtf(vdot(a, a));
    return vscale(a, inv);
}

/* sin/cos: Cephes sinf/cosf range reduction + polynomials as llvmpipe emits them
 * (lp_bld_arit.c lp_build_sin_or_cos, algorithm from Cephes / sse_mathfun); fmuladd is fused */
static float oracle_sincos(float a, int want_cos)
{
    uint32_t ai = f2u(a);
    float x = u2f(ai & 0x7fffffffu);
    float ys = x * 1.27323954473516f; /* 4/pi */
    int32_t j = (int32_t)ys;
    int32_t jadd = j + 1;
    int32_t jj = jadd & ~1;
    float y = (float)jj;
    int32_t sel = want_cos ? jj - 2 : jj;
    uint32_t sign = want_cos ? ((4u & ~(uint32_t)sel) << 29)
                             : ((ai ^ ((uint32_t)jadd << 29)) & 0x80000000u);
    int use_sin_poly = ((sel & 2) == 0);
    float x1 = fmaf(y, -0.78515625f, x);
    float x2 = fmaf(y, -2.4187564849853515625e-4f, x1);
    float x3 = fmaf(y, -3.77489497744594108e-8f, x2);
    float z = x3 * x3;
    float c = fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    float s = fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s = fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = fmaf(s, x3, x3);
    float r = use_sin_poly ? s : c;
    r = u2f(f2u(r) ^ sign);
    if (r < -1.0f) r = -1.0f;
    if (r > 1.0f) r = 1.0f;
    if (!isfinite(a)) r = NAN;
    return r;
}
static inline float osin(float a) { return oracle_sincos(a, 0); }
static inline float ocos(float a) { return oracle_sincos(a, 1); }

/* ---------------------------------------------------------------- RNG (shaders/raytracer.glsl:131-152) */

typedef struct { uint32_t x, y, z, w; } u4;

static inline void pcg4d(u4 *v) /* :140-146 */
{
    v->x = v->x * 1664525u + 1013904223u;
    v->y = v->y * 1664525u + 1013904223u;
    v->z = v->z * 1664525u + 1013904223u;
    v->w = v->w * 1664525u + 1013904223u;
    v->x += v->y * v->w; v->y += v->z * v->x; v->z += v->x * v->y; v->w += v->y * v->z;
    v->x ^= v->x >> 16; v->y ^= v->y >> 16; v->z ^= v->z >> 16; v->w ^= v->w >> 16;
    v->x += v->y * v->w; v->y += v->z * v->x; v->z += v->x * v->y; v->w += v->y * v->z;
}

typedef struct {
    u4 seed;
    oracle_counters *cnt;
    size_t pix;              /* py * W + px of the pixel being traced (ray dump only) */
} rng_t;

/* Diagnostics (tools/diagnostics/cull_emulation.py): when set, every path that reaches bounce g_dump_bounce leaves the ray it enters
 * that bounce with (o.xyz, d.xyz) in g_dump[6 * pixel]; the caller pre-fills the buffer with NaN.  Not used by any test. */
static float *g_dump = NULL;
static uint32_t g_dump_bounce = 0;
void oracle_set_ray_dump(float *buf, uint32_t bounce) { g_dump = buf; g_dump_bounce = bounce; }

static inline float orand(rng_t *r) /* :148-152; float(0xffffffffu) rounds to 2^32 */
{
    pcg4d(&r->seed);
    r->cnt->rand_calls++;
    return (float)r->seed.x / 4294967296.0f;
}

/* random_in_sphere (:154-162): z first, then the angle */
static inline v3 random_in_sphere(rng_t *r)
{
    float z = orand(r) * 2.0f + -1.0f;
    float a = 6.283185482025146484375f * orand(r); /* rand()*2.0*PI folded: 2*float(PI) */
    float rr = sqrtf(1.0f + -(z * z));
    float x = rr * ocos(a);
    float y = rr * osin(a);
    return V(x, y, z);
}

/* ---------------------------------------------------------------- bounds-checked buffer reads */

static inline float rdf(const uint8_t *p) { float f; memcpy(&f, p, 4); return f; }
static inline uint32_t rdu(const uint8_t *p) { uint32_t u; memcpy(&u, p, 4); return u; }

typedef struct { v3 center; float radius; int32_t material; } sphere_t;
static inline sphere_t load_sphere(const oracle_scene *sc, uint32_t i)
{
    sphere_t s = { { 0, 0, 0 }, 0, 0 };
    if (i < sc->n_spheres) {
        const uint8_t *p = sc->spheres + (size_t)i * 32;
        s.center = V(rdf(p), rdf(p + 4), rdf(p + 8));
        s.radius = rdf(p + 12);
        s.material = (int32_t)rdu(p + 16);
    }
    return s;
}
typedef struct { v3 albedo; float smoothness; v3 emission; uint32_t type; } material_t;
static inline material_t load_material(const oracle_scene *sc, int32_t i)
{
    material_t m = { { 0, 0, 0 }, 0, { 0, 0, 0 }, 0 };
    if (i >= 0 && (uint32_t)i < sc->n_materials) {
        const uint8_t *p = sc->materials + (size_t)i * 32;
        m.albedo = V(rdf(p), rdf(p + 4), rdf(p + 8));
        m.smoothness = rdf(p + 12);
        m.emission = V(rdf(p + 16), rdf(p + 20), rdf(p + 24));
        m.type = rdu(p + 28);
    }
    return m;
}
typedef struct { uint32_t left, right, offset, count; } node_t;
static inline node_t load_node(const oracle_scene *sc, uint32_t i)
{
    /* out-of-range node ids are treated as childless empty nodes (see header) */
    node_t n = { ORACLE_INVALID, ORACLE_INVALID, 0, 0 };
    if (i < sc->n_nodes) {
        const uint8_t *p = sc->nodes + (size_t)i * 48;
        n.left = rdu(p + 32); n.right = rdu(p + 36); n.offset = rdu(p + 40); n.count = rdu(p + 44);
    }
    return n;
}

/* material id carried in vertices[3v].w (shaders/raytracer.glsl:353): int(w), with anything that
 * does not fit an int32 mapped to -1 (= "no such material", reads as the zero material) */
static inline int32_t material_from_w(float w)
{
    if (!(w > -2147483648.0f && w < 2147483648.0f)) return -1;
    return (int32_t)w;
}

/* ---------------------------------------------------------------- intersection */

typedef struct { float t; v3 point, normal; int32_t material; } hit_t;

/* sphere_intersect (:200-220) */
static inline float sphere_intersect(v3 o, v3 d, const sphere_t *s)
{
    v3 op = vsub(s->center, o);
    float b = vdot(op, d);
    float det = (b * b - vdot(op, op)) + s->radius * s->radius;
    if (det < 0.0f) return ORACLE_INF;
    det = sqrtf(det);
    float t1 = b - det;
    if (0.001f < t1) return t1;
    float t2 = b + det;
    if (0.001f < t2) return t2;
    return ORACLE_INF;
}

/* traverse (:272-329): LIFO walk over all nodes, AABB cull disabled in the reference (:288-292) */
static int traverse(const oracle_scene *sc, v3 o, v3 d, hit_t *hit, oracle_counters *cnt)
{
    int closest = -1;
    if (sc->n_nodes == 0) return closest;
    uint32_t items[ORACLE_STACK] = { 0, 0, 0, 0, 0 };
    int top = 0; /* push(s, 0) */
    int pops = 0;
    while (top != -1 && pops < ORACLE_NODE_POP_CAP) {
        uint32_t id = items[top--];
        pops++;
        node_t n = load_node(sc, id);
        if (n.left != ORACLE_INVALID && top != ORACLE_STACK - 1) items[++top] = n.left;
        if (n.right != ORACLE_INVALID && top != ORACLE_STACK - 1) items[++top] = n.right;
        if (n.count > 0) {
            uint32_t end = n.offset + n.count;
            for (uint32_t i = n.offset; i < end; i++) {
                sphere_t s = load_sphere(sc, i);
                cnt->sphere_tests++;
                float t = sphere_intersect(o, d, &s);
                if (ORACLE_EPS < t && t < hit->t) {
                    hit->t = t;
                    hit->point = vadd(o, vscale(d, t));
                    v3 pc = vsub(hit->point, s.center);
                    hit->normal = V(pc.x / s.radius, pc.y / s.radius, pc.z / s.radius);
                    hit->material = s.material;
                    closest = (int)i;
                }
            }
        }
    }
    return closest;
}

/* find_closest_mesh (:331-361) with triangle_intersect (:223-256) inlined.
 * Edge tests: dot(e_k, cv) + dot(m_k, cu) > 0 is evaluated as -dot(e_k, cv) < dot(m_k, cu)
 * (what llvmpipe executes; identical outcome to the rounded sum for all finite inputs). */
static int find_closest_mesh(const oracle_scene *sc, v3 o, v3 d, hit_t *hit, oracle_counters *cnt)
{
    float max_t = ORACLE_INF;
    int closest = -1;
    uint32_t n_tris = sc->n_vertices / 3;
    v3 cu = d;
    v3 cv = vcross(d, o);
    for (uint32_t i = 0; i < sc->n_meshes; i++) {
        uint32_t start = rdu(sc->meshes + (size_t)i * 16);
        uint32_t size = rdu(sc->meshes + (size_t)i * 16 + 4);
        uint64_t end64 = (uint64_t)start + size;
        uint32_t end = end64 > n_tris ? n_tris : (uint32_t)end64;
        for (uint32_t v = start; v < end; v++) {
            const uint8_t *p = sc->vertices + (size_t)v * 48;
            v3 v0 = V(rdf(p), rdf(p + 4), rdf(p + 8));
            v3 v1 = V(rdf(p + 16), rdf(p + 20), rdf(p + 24));
            v3 v2 = V(rdf(p + 32), rdf(p + 36), rdf(p + 40));
            cnt->triangle_tests++;
            v3 e0 = vsub(v1, v0), m0 = vcross(v1, v0);
            if (!(-vdot(e0, cv) < vdot(m0, cu))) continue;
            v3 e1 = vsub(v2, v1), m1 = vcross(v2, v1);
            if (!(-vdot(e1, cv) < vdot(m1, cu))) continue;
            v3 e2 = vsub(v0, v2), m2 = vcross(v0, v2);
            if (!(-vdot(e2, cv) < vdot(m2, cu))) continue;
            v3 n = vnormalize(vcross(vsub(v1, v0), vsub(v2, v0)));
            float q = vdot(d, n);
            float t = -vdot(vsub(o, v0), n) / q;
            if (!(t < ORACLE_INF)) continue;
            if (ORACLE_EPS < t && t < max_t) {
                hit->t = t;
                hit->point = vadd(o, vscale(d, t));
                hit->normal = n;
                hit->material = material_from_w(rdf(p + 12));
                max_t = t;
                closest = (int)i;
            }
        }
    }
    return closest;
}

/* ---------------------------------------------------------------- cube map (texture(u_envmap, dir), :442)
 * Face selection / projection per the OpenGL 4.3 cube map table; filtering as llvmpipe does it
 * for 8-bit UNORM formats: texel coordinates in fixed point with 8 fractional bits (round to
 * nearest even), clamp-to-edge inside the face (non-seamless), 8-bit lerps rounded like
 * pmulhrsw, result * float(1/255). */
static inline int64_t lerp8(int64_t w, int64_t v0, int64_t v1)
{
    int64_t p = w * ((v1 - v0) * 128);
    int64_t r = (((p >> 14) + 1) >> 1) & 0xff;
    return (v0 + r) & 0xff;
}
static v3 env_lookup(const oracle_scene *sc, v3 dir)
{
    if (!sc->env || sc->env_faces < 6 || sc->env_w <= 0 || sc->env_h <= 0) return V(0, 0, 0); /* incomplete cube: black */
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    int x_over_y = ax > ay;
    float mxy = ax > ay ? ax : ay;
    int z_major = az >= mxy;
    int face; float ma, fs, ft;
    if (z_major) { face = 4; ma = dir.z; fs = signbit(dir.z) ? -dir.x : dir.x; ft = -dir.y; }
    else if (x_over_y) { face = 0; ma = dir.x; fs = signbit(dir.x) ? dir.z : -dir.z; ft = -dir.y; }
    else { face = 2; ma = dir.y; fs = dir.x; ft = signbit(dir.y) ? -dir.z : dir.z; }
    if (signbit(ma)) face += 1;
    float ima = 0.5f / fabsf(ma);
    float s = fs * ima + 0.5f;
    float t = ft * ima + 0.5f;
    int W = sc->env_w, H = sc->env_h, C = sc->env_channels;
    float sf = (s * (float)W) * 256.0f, tf = (t * (float)H) * 256.0f;
    /* NaN / out-of-range directions: GL-undefined; clamp so the lookup stays inside the face */
    if (!(sf > -1e9f)) sf = -1e9f; if (sf > 1e9f) sf = 1e9f;
    if (!(tf > -1e9f)) tf = -1e9f; if (tf > 1e9f) tf = 1e9f;
    int64_t si = (int64_t)rintf(sf) - 128, ti = (int64_t)rintf(tf) - 128;
    int64_t s0 = si >> 8, sw = si & 0xff, t0 = ti >> 8, tw = ti & 0xff;
    int64_t s1 = s0 + 1, t1 = t0 + 1;
    if (s0 < 0) s0 = 0; if (s0 > W - 1) s0 = W - 1; if (s1 < 0) s1 = 0; if (s1 > W - 1) s1 = W - 1;
    if (t0 < 0) t0 = 0; if (t0 > H - 1) t0 = H - 1; if (t1 < 0) t1 = 0; if (t1 > H - 1) t1 = H - 1;
    const uint8_t *f = sc->env + (size_t)face * W * H * C;
    float out[3];
    for (int c = 0; c < 3; c++) {
        int64_t v00 = f[((size_t)t0 * W + s0) * C + c], v01 = f[((size_t)t0 * W + s1) * C + c];
        int64_t v10 = f[((size_t)t1 * W + s0) * C + c], v11 = f[((size_t)t1 * W + s1) * C + c];
        int64_t r0 = lerp8(sw, v00, v01), r1 = lerp8(sw, v10, v11);
        out[c] = (float)lerp8(tw, r0, r1) * 0.0039215688593685626983642578125f; /* float(1/255) */
    }
    return V(out[0], out[1], out[2]);
}

/* ---------------------------------------------------------------- trace_path (:420-529) */

static v3 trace_path(const oracle_scene *sc, const oracle_params *P, v3 o, v3 d, rng_t *rng)
{
    oracle_counters *cnt = rng->cnt;
    v3 radiance = V(0, 0, 0), thr = V(1, 1, 1);
    cnt->paths++;
    for (uint32_t bounce = 0; bounce < P->max_bounce; bounce++) {
        cnt->segments++;
        if (g_dump && bounce == g_dump_bounce) { float *q = g_dump + 6 * rng->pix; q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = d.x; q[4] = d.y; q[5] = d.z; }
        hit_t h1, h2;
        memset(&h1, 0, sizeof h1); memset(&h2, 0, sizeof h2);
        h1.t = ORACLE_INF; h2.t = ORACLE_INF;
        int i = traverse(sc, o, d, &h1, cnt);
        int j = find_closest_mesh(sc, o, d, &h2, cnt);
        if (i == -1 && j == -1) { /* :441-445 */
            v3 bg;
            if (P->use_envmap) { bg = env_lookup(sc, d); cnt->env_lookups++; }
            else bg = V(P->background[0], P->background[1], P->background[2]);
            radiance = vadd(radiance, vmul(bg, thr));
            break;
        }
        hit_t h = (h1.t < h2.t) ? h1 : h2; /* :447 */
        material_t m = load_material(sc, h.material);
        v3 n = h.normal;
        /* inside = dot(-d, n) < 0  (:455), evaluated as llvmpipe does: -(dz*nz) - (dy*ny) < dx*nx */
        int inside = (-(d.z * n.z) + -(d.y * n.y)) < d.x * n.x;
        o = h.point; /* :460 */
        if (m.type == 0) { /* diffuse :462-466 */
            d = vnormalize(vadd(n, random_in_sphere(rng)));
            thr = vmul(thr, m.albedo);
        } else if (m.type == 1) { /* specular :467-474, direction deliberately not normalised */
            v3 diffuse = vnormalize(vadd(n, random_in_sphere(rng)));
            float dn2 = vdot(d, n) * 2.0f;
            v3 refl = vsub(d, vscale(n, dn2));
            d = V(diffuse.x + m.smoothness * (refl.x - diffuse.x),
                  diffuse.y + m.smoothness * (refl.y - diffuse.y),
                  diffuse.z + m.smoothness * (refl.z - diffuse.z));
            thr = vmul(thr, m.albedo);
        } else if (m.type == 2) { /* transmissive :475-523 */
            v3 nl = inside ? vneg(n) : n;
            float nnt = inside ? 1.4f : 0.714285731315612793f; /* nt/nc : nc/nt */
            float ct = vdot(d, nl);
            float omc = 1.0f + -(ct * ct);
            if (1.0f < (nnt * nnt) * omc) break; /* total internal reflection ends the path (:490-494) */
            /* refract(d, nl, nnt): k = 1 - eta*(eta*(1 - c*c)) */
            float k = 1.0f + -(nnt * (nnt * omc));
            v3 T = V(0, 0, 0);
            if (!(k < 0.0f)) {
                float f = nnt * ct + sqrtf(k);
                T = vsub(vscale(d, nnt), vscale(nl, f));
            }
            float c2 = vdot(T, n);
            float tmp = inside ? c2 : -ct;
            /* fresnel_schlick(R0, tmp) (:414-418) in llvmpipe's association: R0 + ((c*c)*(1-R0))*((c*c)*c) */
            float c = 1.0f + -tmp;
            float cc = c * c;
            float X = (cc * 0.97222220897674560546875f) * (cc * c);
            float Re = 0.02777777425944805145263671875f + X;
            float halfRe = 0.5f * Re;
            float Pp = 0.25f + halfRe;
            float RP = Re / Pp;
            float TP = (0.97222220897674560546875f + -X) / (0.75f + -halfRe);
            if (orand(rng) < Pp) {
                thr = vmul(thr, vscale(m.albedo, RP));
                float dn2 = vdot(d, n) * 2.0f;
                d = vsub(d, vscale(n, dn2));
            } else {
                thr = vmul(thr, vscale(m.albedo, TP));
                d = T;
            }
        }
        /* any other type: direction and throughput pass through unchanged */
        radiance = vadd(radiance, vmul(m.emission, thr)); /* :525 */
    }
    return radiance;
}

/* ---------------------------------------------------------------- main() of the shader (:531-569) */

static void render_pixel(const oracle_scene *sc, const oracle_params *P, int W, int H, int px, int py,
                         float *image, uint32_t *seed_out, oracle_counters *cnt)
{
    rng_t rng;
    rng.cnt = cnt; rng.pix = (size_t)py * (size_t)W + (size_t)px;
    rng.seed.x = (uint32_t)px; rng.seed.y = (uint32_t)py; rng.seed.z = (uint32_t)P->random;
    rng.seed.w = (uint32_t)px + (uint32_t)py + (uint32_t)P->random; /* init_rand :135-138 */
    float fw = (float)W, fh = (float)H;
    float aspect = fh / fw; /* :540 */
    float *pix = image + ((size_t)py * W + px) * 4;
    v3 prev = P->reset_flag ? V(0, 0, 0) : V(pix[0], pix[1], pix[2]); /* :542-548 */
    float ndx = ((float)px / fw) * 2.0f + -1.0f; /* :550 */
    float ndy = ((float)py / fh) * 2.0f + -1.0f;
    /* camera_ray (:168-198) */
    v3 pos = V(P->camera_position[0], P->camera_position[1], P->camera_position[2]);
    v3 fwd = V(P->camera_forward[0], P->camera_forward[1], P->camera_forward[2]);
    v3 up = V(P->camera_up[0], P->camera_up[1], P->camera_up[2]);
    v3 right = V(P->camera_right[0], P->camera_right[1], P->camera_right[2]);
    float half = P->camera_fov / 2.0f;
    float tn = osin(half) / ocos(half);
    v3 target = vadd(pos, fwd);
    v3 vp = vadd(target, vscale(vscale(right, 2.0f * tn), ndx));
    vp = vadd(vp, vscale(vscale(up, 2.0f * (tn * aspect)), ndy));
    v3 dir = vnormalize(vsub(vp, pos));
    v3 origin = pos;
    if (P->use_dof) {
        v3 s = random_in_sphere(&rng);
        v3 jitter = vscale(s, P->camera_aperture);
        origin = vadd(pos, jitter);
        v3 focal_point = vadd(pos, vscale(dir, P->camera_focal_length));
        dir = vnormalize(vsub(focal_point, origin));
    }
    v3 color = V(0, 0, 0);
    for (uint32_t s = 0; s < P->samples; s++) /* :556-559: same ray, RNG state carried over */
        color = vadd(color, trace_path(sc, P, origin, dir, &rng));
    float ns = (float)P->samples;
    color = V(color.x / ns, color.y / ns, color.z / ns); /* :561 */
    float fr = (float)P->frames, fr1 = (float)(P->frames + 1);
    pix[0] = (color.x + prev.x * fr) / fr1; /* :563-564 */
    pix[1] = (color.y + prev.y * fr) / fr1;
    pix[2] = (color.z + prev.z * fr) / fr1;
    pix[3] = 1.0f;
    if (seed_out) {
        uint32_t *so = seed_out + ((size_t)py * W + px) * 4;
        so[0] = rng.seed.x; so[1] = rng.seed.y; so[2] = rng.seed.z; so[3] = rng.seed.w;
    }
}

typedef struct {
    const oracle_scene *sc; const oracle_params *P; int W, H, x0, y0, x1, y1;
    float *image; uint32_t *seed_out; volatile int next_row; oracle_counters cnt; pthread_mutex_t mu;
} job_t;

static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    oracle_counters c; memset(&c, 0, sizeof c);
    const int w = jb->x1 - jb->x0, h = jb->y1 - jb->y0;
    const int chunks_per_row = (w + 63) / 64, n_chunks = chunks_per_row * (h > 0 ? h : 0);
    for (;;) {
        int k = __sync_fetch_and_add(&jb->next_row, 1);   /* work item = 64 consecutive pixels of one row */
        if (k >= n_chunks) break;
        int y = jb->y0 + k / chunks_per_row, xa = jb->x0 + (k % chunks_per_row) * 64;
        int xb = xa + 64 < jb->x1 ? xa + 64 : jb->x1;
        for (int x = xa; x < xb; x++)
            render_pixel(jb->sc, jb->P, jb->W, jb->H, x, y, jb->image, jb->seed_out, &c);
    }
    pthread_mutex_lock(&jb->mu);
    jb->cnt.paths += c.paths; jb->cnt.segments += c.segments; jb->cnt.sphere_tests += c.sphere_tests;
    jb->cnt.triangle_tests += c.triangle_tests; jb->cnt.rand_calls += c.rand_calls; jb->cnt.env_lookups += c.env_lookups;
    pthread_mutex_unlock(&jb->mu);
    return NULL;
}

/* Render one frame over the pixel rectangle [x0,x1) x [y0,y1) of a W x H RGBA32F image (row 0 =
 * pixel y 0, like the GL image).  Like the reference dispatch (src/renderer.cpp:132-133) callers
 * normally pass x1 = W/8*8, y1 = H/8*8.  seed_out (optional, W*H*4 u32) receives each pixel's
 * final RNG state.  Returns 0. */
int oracle_render(const oracle_scene *sc, const oracle_params *P, int W, int H, float *image,
                  int x0, int y0, int x1, int y1, int nthreads, uint32_t *seed_out, oracle_counters *out_cnt)
{
    job_t jb;
    memset(&jb, 0, sizeof jb);
    jb.sc = sc; jb.P = P; jb.W = W; jb.H = H; jb.x0 = x0; jb.y0 = y0; jb.x1 = x1; jb.y1 = y1;
    jb.image = image; jb.seed_out = seed_out; jb.next_row = 0;
    pthread_mutex_init(&jb.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (int i = 1; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &jb);
    worker(&jb);
    for (int i = 1; i < nthreads; i++) pthread_join(th[i], NULL);
    if (out_cnt) *out_cnt = jb.cnt;
    return 0;
}

/* element-wise helpers exported so tests can pin the math against llvmpipe probes */
void oracle_sincos_array(const float *a, float *s, float *c, int n)
{
    for (int i = 0; i < n; i++) { s[i] = osin(a[i]); c[i] = ocos(a[i]); }
}
void oracle_env_lookup_array(const oracle_scene *sc, const float *dirs, float *rgb, int n)
{
    for (int i = 0; i < n; i++) {
        v3 r = env_lookup(sc, V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
        rgb[3 * i] = r.x; rgb[3 * i + 1] = r.y; rgb[3 * i + 2] = r.z;
    }
}
void oracle_pcg4d(uint32_t *v4, int rounds)
{
    u4 s = { v4[0], v4[1], v4[2], v4[3] };
    for (int i = 0; i < rounds; i++) pcg4d(&s);
    v4[0] = s.x; v4[1] = s.y; v4[2] = s.z; v4[3] = s.w;
}
