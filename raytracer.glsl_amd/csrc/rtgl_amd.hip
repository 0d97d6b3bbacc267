// rtgl_amd.hip -- HIP kernels + C ABI (include/rtgl_amd.h) of the MI355X-native path tracer.
// gfx950 only.  Build: see raytracer.glsl_amd/csrc/Makefile (hipcc --offload-arch=gfx950
// -ffp-contract=off).  There is no CPU fallback: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <memory>

#include "../../include/rtgl_amd.h"
#include "rt_device.hpp"
#include "rt_wavefront.hpp"
#include "rt_mfma.hpp"
#include "rt_scan.hpp"

#pragma clang fp contract(off)

using namespace rt;

// =================================================================================================
// Kernels
// =================================================================================================

// ---- upload-time preparation -------------------------------------------------------------------
// One thread per triangle visit.  visit_tri[k] = triangle index tested k-th by the reference's
// mesh loops (find_closest_mesh :336-341: meshes in order, triangles start..start+size-1 each).
__global__ void __launch_bounds__(256) prepare_triangles_kernel(const float4 *__restrict__ vertices,
                                                                const uint32_t *__restrict__ visit_tri, uint32_t n_visits,
                                                                TriEdges *__restrict__ edges, TriPlane *__restrict__ planes)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_visits) return;
    uint32_t tri = visit_tri[k];
    float4 a = vertices[3 * (size_t)tri + 0], b = vertices[3 * (size_t)tri + 1], c = vertices[3 * (size_t)tri + 2];
    f3 v0 = mk(a.x, a.y, a.z), v1 = mk(b.x, b.y, b.z), v2 = mk(c.x, c.y, c.z);
    f3 e0 = v1 - v0, e1 = v2 - v1, e2 = v0 - v2;                                   // :230,233,236
    f3 m0 = cross3(v1, v0), m1 = cross3(v2, v1), m2 = cross3(v0, v2);              // :231,234,237
    f3 n = normalize3(cross3(v1 - v0, v2 - v0));                                   // :239
    TriEdges E;
    E.e0x = e0.x; E.e0y = e0.y; E.e0z = e0.z; E.e1x = e1.x; E.e1y = e1.y; E.e1z = e1.z; E.e2x = e2.x; E.e2y = e2.y; E.e2z = e2.z;
    E.m0x = m0.x; E.m0y = m0.y; E.m0z = m0.z; E.m1x = m1.x; E.m1y = m1.y; E.m1z = m1.z; E.m2x = m2.x; E.m2y = m2.y; E.m2z = m2.z;
    float be = fmaxf(fmaxf(dot3(e0, e0), dot3(e1, e1)), dot3(e2, e2));
    float bm = fmaxf(fmaxf(dot3(m0, m0), dot3(m1, m1)), dot3(m2, m2));
    E.bound_e = __builtin_sqrtf(be) * 1.0001f;
    E.bound_m = __builtin_sqrtf(bm) * 1.0001f;
    edges[k] = E;
    TriPlane P;
    P.nx = n.x; P.ny = n.y; P.nz = n.z; P.v0x = v0.x; P.v0y = v0.y; P.v0z = v0.z;
    float w = a.w;                                                                 // int(vertices[3v].w) :353
    P.material = (w > -2147483648.0f && w < 2147483648.0f) ? (int32_t)w : -1;
    P.pad = 0;
    planes[k] = P;
}

// ---- variant 0: megakernel, one lane per pixel ----------------------------------------------------
// Triangle records are wave-uniform, so the compiler fetches them with scalar loads (SGPR operands
// feed the fma chain directly); no LDS traffic.  Baseline variant, kept as the A/B reference for the
// tiled / wavefront kernels.
template <bool kCount>
__global__ void __launch_bounds__(256) pathtrace_mega_kernel(SceneView sc, FrameParams P, ImageView im, uint4 *rng_out, Counters *counters)
{
    // 8x8 pixel block per wave (matches the reference work-group shape :38), 4 waves side by side
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = (blockIdx.x * 4 + wave) * 8 + (lane & 7);
    const int lrow = blockIdx.y * 8 + (lane >> 3);
    if (px >= im.disp_w || lrow >= im.local_rows) return;
    const int py = local_to_global_row(im, lrow);
    if (py >= im.disp_h) return;

    Rng rng; rng.x = (uint32_t)px; rng.y = (uint32_t)py; rng.z = (uint32_t)P.random;       // init_rand :135-138
    rng.w = (uint32_t)px + (uint32_t)py + (uint32_t)P.random;
    float4 *pix = im.pixels + (size_t)lrow * im.width + px;
    f3 prev = mk(0.0f, 0.0f, 0.0f);
    if (!P.reset_flag) { float4 q = *pix; prev = mk(q.x, q.y, q.z); }                       // :542-548
    f3 origin, dir;
    camera_ray(P, px, py, im.width, im.height, rng, origin, dir);

    unsigned long long c_seg = 0, c_tests = 0, c_cand = 0, c_env = 0;
    f3 color = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t s = 0; s < P.samples; ++s) {                                              // :556-559
        f3 o = origin, d = dir;
        f3 radiance = mk(0.0f, 0.0f, 0.0f), thr = mk(1.0f, 1.0f, 1.0f);
        for (uint32_t bounce = 0; bounce < P.max_bounce; ++bounce) {                        // :425
            if (kCount) c_seg++;
            Hit h1; h1.t = kInf; h1.material = 0; h1.point = h1.normal = mk(0.0f, 0.0f, 0.0f);
            bool hit_sphere = sphere_pass(sc, o, d, h1);                                    // :433
            // find_closest_mesh (:331-361)
            TriRay tr = make_tri_ray(o, d);
            float best_t = kInf; uint32_t best_v = 0xFFFFFFFFu;
            for (uint32_t v = 0; v < sc.n_tri_visits; ++v) {
                const TriEdges &T = sc.tri_edges[v];
                if (tri_filter(T, tr)) {
                    if (kCount) c_cand++;
                    float t = tri_exact(T, sc.tri_planes[v], tr);
                    if (kEps < t && t < best_t) { best_t = t; best_v = v; }
                }
            }
            if (kCount) c_tests += sc.n_tri_visits;
            bool hit_mesh = best_v != 0xFFFFFFFFu;
            if (!hit_sphere && !hit_mesh) {                                                 // :441-445
                f3 bg;
                if (P.use_envmap) { bg = env_lookup(sc, d); if (kCount) c_env++; }
                else bg = mk(P.background[0], P.background[1], P.background[2]);
                radiance = radiance + bg * thr;
                break;
            }
            Hit h = h1;
            if (!(h1.t < best_t)) {                                                         // :447
                const TriPlane &pl = sc.tri_planes[best_v];
                h.t = best_t; h.point = o + d * best_t; h.normal = mk(pl.nx, pl.ny, pl.nz); h.material = pl.material;
            }
            if (!shade_hit(sc, h, rng, o, d, thr, radiance)) break;
        }
        color = color + radiance;
    }
    { const float4 v = accumulate_pixel(P, color, prev); store_through(pix, v.x, v.y, v.z, v.w); }      // (rt_wavefront.hpp: read by the next frame's kernel, from whichever XCD)
    if (rng_out) rng_out[(size_t)lrow * im.width + px] = make_uint4(rng.x, rng.y, rng.z, rng.w);
    if (kCount) {
        atomicAdd(&counters->paths, (unsigned long long)P.samples);
        atomicAdd(&counters->segments, c_seg);
        atomicAdd(&counters->tri_tests, c_tests);
        atomicAdd(&counters->candidates, c_cand);
        atomicAdd(&counters->env_lookups, c_env);
    }
}

// ---- 8-bit readback (glGetTexImage GL_UNSIGNED_BYTE, src/renderer.cpp:223) -----------------------
__global__ void __launch_bounds__(256) image_to_u8_kernel(const float4 *__restrict__ src, uchar4 *__restrict__ dst, int width, int rows, int flip)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * rows) return;
    int y = i / width, x = i - y * width;
    float4 p = src[i];
    auto q = [](float v) -> unsigned char {
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);      // NaN -> 0 like a clamp-to-[0,1] conversion
        if (!(v == v)) v = 0.0f;
        return (unsigned char)__builtin_rintf(v * 255.0f);
    };
    int oy = flip ? rows - 1 - y : y;
    dst[(size_t)oy * width + x] = make_uchar4(q(p.x), q(p.y), q(p.z), q(p.w));
}

// =================================================================================================
// Host side: context + C ABI
// =================================================================================================

static thread_local std::string g_create_error;

// rtgl_create_multi: one submit thread per part.  A frame is ~27 launches = ~150 us of host time per device (measured,
// tools/diagnostics/host_enqueue.py) while a rank of 8 renders its strips of the 1080p benchmark frame in 0.68 ms: ONE thread submitting
// to 8 devices one after the other (1.2 ms) would be the limiter, 8 threads side by side are not.
struct PartWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    enum { kIdle, kJob, kDone, kQuit } state = kIdle;
    int rc = 0;
    rtgl_context *part = nullptr;
};

struct rtgl_context {
    int device = 0;
    int width = 0, height = 0;
    int rank = 0, world = 1, strip_rows = 8, local_rows = 0;
    hipStream_t own_stream = nullptr, stream = nullptr; bool own_stream_shared = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // kernel_timing: events since the last rtgl_timing_reset.  Per frame: [frame begin, (launch begin, launch end)*, frame end]
    std::vector<hipEvent_t> kev;
    uint32_t kev_used = 0;
    std::vector<uint32_t> kev_frame_start;   // index into kev of each recorded frame's first event
    std::string error;

    // raw scene copies (host) used to rebuild derived buffers
    std::vector<uint8_t> h_meshes, h_nodes, h_vertices;
    uint32_t n_meshes = 0, n_nodes = 0, n_vec4 = 0;

    // device buffers
    SphereRec *d_spheres = nullptr; uint32_t n_spheres = 0;
    MaterialRec *d_materials = nullptr; uint32_t n_materials = 0;
    float4 *d_vertices = nullptr;
    uint32_t *d_sphere_visits = nullptr; uint32_t n_sphere_visits = 0;
    TriEdges *d_edges = nullptr; TriPlane *d_planes = nullptr; uint32_t n_tri_visits = 0;
    TriEdges *d_edges_s = nullptr; TriPlane *d_planes_s = nullptr;          // ... in the storage order of the matrix-core scan (narrow phase)
    uint8_t *d_env = nullptr; int env_w = 0, env_h = 0, env_c = 0, env_faces = 0;
    float4 *d_image_own = nullptr, *d_image = nullptr;
    uint4 *d_rng = nullptr;
    Counters *d_counters = nullptr;
    uchar4 *d_u8 = nullptr;

    // bounce-wavefront pipeline buffers
    float2 *d_group_bounds = nullptr;
    MfGroup *d_mf_groups = nullptr; MfCull *d_mf_cull = nullptr; uint4 *d_mf_A = nullptr; uint32_t *d_mf_order = nullptr; uint32_t n_mf_groups = 0, mf_group_quads = 32; uint32_t *d_dbg_log = nullptr;   // bf16 matrix-core broad phase
    void *d_wave = nullptr; size_t wave_capacity = 0; bool wave_multi = false;   // queues (+ per-pixel state when u_samples > 1)
    uint32_t *d_counts = nullptr; uint32_t counts_capacity = 0;
    uint32_t *h_counts = nullptr;            // pinned: ray counts per bounce of the most recent finished frame
    hipEvent_t counts_ev = nullptr; bool counts_pending = false, counts_valid = false;
    bool timing_this_frame = false; uint32_t timing_frame_counter = 0;
    int kernel_in_use = -1;                  // variant the last frame actually ran
    int n_cus = 256;
    // kernel 4 candidate buffer: one region per wave of a scan launch.  Sized from what the scene needs, not from the image: it starts
    // at one record per ray and grows to 1.25 x the fullest region any finished frame reported (records that do not fit are tested
    // in place by the scan, so every size is correct; a too small one is only slower)
    uint2 *d_items = nullptr; size_t items_capacity = 0;          // packet culling: per chunk of a culled scan launch its work items + one count per chunk
    uint32_t *d_sched = nullptr; size_t sched_capacity = 0;       // kernel 4: next unclaimed item per (bounce, chunk)
    uint32_t *d_keep = nullptr; size_t keep_capacity = 0;         // packet culling: (granules of 128 rays) x (tiles / 32) words
    // the camera-ray bounce's keep bits, kept across frames while camera, image and scene stand still (a progressive render's normal state):
    // the camera rays of two frames differ by the depth-of-field jitter only, so bits certified for one frame's rays with the packet bounds
    // widened by that jitter hold for all of them and packet_cull_kernel is skipped on bounce 0 (a third of its work)
    uint32_t *d_keep0 = nullptr; size_t keep0_capacity = 0; bool keep0_valid = false; FrameParams keep0_params{}; uint32_t keep0_n0 = 0, keep0_words = 0; uint64_t keep0_scene = 0;
    uint64_t scene_version = 0;
    void *d_plan = nullptr; size_t plan_capacity = 0;             // planned work distribution of culled scan launches: cost prefix sums per chunk
    void *d_stage = nullptr; size_t stage_capacity = 0;           // ray binning: the staging queue + (key, rank) per slot
    uint32_t *d_sort_hist = nullptr; uint32_t sort_bits_alloc = 0;      // two sets of bin counters, used in turns
    int sort_set = 0; uint32_t sort_set_bits = 0; bool sort_sets_clean = false;      // the set the next binned bounce counts in; false: zero both first (fresh, or a frame was abandoned half way)
    float mesh_lo[3] = {0.0f, 0.0f, 0.0f}, mesh_hi[3] = {0.0f, 0.0f, 0.0f}, mesh_ext = 0.0f;       // box of the triangles' finite vertices (origin cells of the bin key)
    uint2 *d_cand = nullptr; uint32_t cand_regions = 0, cand_region_pairs = 0, cand_region_target = 0; bool cand_fixed = false;
    bool solo_attr_set = false;              // hipFuncAttributeMaxDynamicSharedMemorySize is per device: raised once per context (= per device binding)
    bool group_explicit = false;             // "mf_group_quads" was set through rtgl_set_option
    bool kernel_explicit = false;            // "kernel" was set through rtgl_set_option or RTGL_AMD_KERNEL
    uint32_t counts_n0 = 0, counts_len = 0;
    std::vector<uint32_t> est_counts;        // grid-size estimates for the next frame
    WaveBuffers wb{};

    // single-process multi-device mode (rtgl_create_multi): this context is the assembler -- it owns the full image on devices[0] --
    // and `parts` are the per-device tiled contexts that render the strips; every entry point fans out to them
    std::vector<rtgl_context *> parts;
    std::vector<hipEvent_t> part_done;
    hipEvent_t gather_done = nullptr;                     // recorded behind the gather's copies; every part waits for it before its next frame
    std::vector<std::unique_ptr<PartWorker>> workers;     // one per part when there is more than one (RTGL_AMD_MULTI_THREADS=0: none, the caller's thread submits)
    bool gathered = false, peer_copy = true;

    // frame batching (option "frame_batch" = B > 1): rtgl_render_frame only records the frame's uniforms until B frames are waiting (or
    // anything else is asked of the context); the B frames then travel through one set of launches (render_batch)
    std::vector<FrameParams> pending;
    float4 *d_batch_rad = nullptr; size_t batch_capacity = 0;
    uint32_t last_batch_frames = 1;
    bool tris_dirty = false, visits_dirty = false;
    FrameParams params{};
    bool have_params = false;
    int opt_kernel = RTGL_KERNEL_WAVEFRONT_MFMA_SOLO, opt_rng_state = 0, opt_counters = 0, opt_kernel_timing = 0, opt_wf_rays = 4, opt_wf_mode = kLds, opt_wf_chunk = 256, opt_wf_early = 0, opt_wf_packed = 0, opt_mf_chunk_quads = 32, opt_mf_group_quads = 32, opt_cull = 3, opt_sort_min_rays = 131072, opt_scan_waves = 0, opt_scan_dynamic = 0, opt_debug_skip_exact = 0, opt_frame_batch = 1;
};

static int fail(rtgl_context *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return code;
}
#define HIPCHK(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(ctx, RTGL_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

template <typename T>
static int realloc_upload(rtgl_context *ctx, T *&dptr, const void *src, size_t bytes)
{
    if (dptr) { HIPCHK(ctx, hipFree(dptr)); dptr = nullptr; }
    if (bytes == 0) return RTGL_OK;
    HIPCHK(ctx, hipMalloc((void **)&dptr, bytes));
    if (src) HIPCHK(ctx, hipMemcpyAsync(dptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // uploads are synchronous like glBufferData
    return RTGL_OK;
}

static int local_rows_for(int height, int rank, int world, int strip_rows)
{
    int n_strips = (height + strip_rows - 1) / strip_rows, rows = 0;
    for (int s = rank; s < n_strips; s += world) rows += std::min(strip_rows, height - s * strip_rows);
    return rows;
}

// Every context of a process on one device submits to ONE stream (unless the caller binds its own with rtgl_set_stream): the kernels
// of different contexts then never run beside each other.  Several path-tracing pipelines running CONCURRENTLY on a device have
// produced wrong frames (16 rays of a launch scanning stale data; DESIGN.md 5.2 -- rare with the shipped kernels, cause not
// established); a single pipeline at a time never has.  RTGL_AMD_PRIVATE_STREAMS=1 restores one stream per context (diagnostics:
// tools/diagnostics/flaky_tiled.py).
static std::mutex g_stream_mutex;
static std::map<int, std::pair<hipStream_t, int>> g_device_streams;
static hipError_t acquire_device_stream(int device, hipStream_t *out, bool *shared)
{
    const char *priv = getenv("RTGL_AMD_PRIVATE_STREAMS");
    if (priv && atoi(priv) != 0) { *shared = false; return hipStreamCreateWithFlags(out, hipStreamNonBlocking); }
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    auto it = g_device_streams.find(device);
    if (it == g_device_streams.end()) {
        hipStream_t st = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        it = g_device_streams.emplace(device, std::make_pair(st, 0)).first;
    }
    it->second.second++;
    *out = it->second.first; *shared = true;
    return hipSuccess;
}
// Which stream every live context of this process submits to, per device: rtgl_set_stream refuses a binding that would put two
// path-tracing pipelines on DIFFERENT streams of one device (they could then run concurrently: DESIGN.md 5.2) unless the caller takes
// that over explicitly (RTGL_AMD_ALLOW_CONCURRENT_PIPELINES=1; RTGL_AMD_PRIVATE_STREAMS=1 implies it).
static std::map<const rtgl_context *, std::pair<int, hipStream_t>> g_ctx_streams;
static void register_ctx_stream(const rtgl_context *ctx, int device, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    g_ctx_streams[ctx] = std::make_pair(device, st);
}
static void unregister_ctx_stream(const rtgl_context *ctx)
{
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    g_ctx_streams.erase(ctx);
}
static bool other_stream_in_use(const rtgl_context *ctx, int device, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    for (const auto &kv : g_ctx_streams) if (kv.first != ctx && kv.second.first == device && kv.second.second != st) return true;
    return false;
}
static bool concurrent_pipelines_allowed()
{
    const char *a = getenv("RTGL_AMD_ALLOW_CONCURRENT_PIPELINES"), *p = getenv("RTGL_AMD_PRIVATE_STREAMS");
    return (a && atoi(a) != 0) || (p && atoi(p) != 0);
}

static void release_device_stream(int device, hipStream_t st, bool shared)
{
    if (!shared) { (void)hipStreamDestroy(st); return; }
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    auto it = g_device_streams.find(device);
    if (it != g_device_streams.end() && --it->second.second == 0) { (void)hipStreamDestroy(it->second.first); g_device_streams.erase(it); }
}

extern "C" int rtgl_create_tiled(rtgl_context **out, int width, int height, int device, int rank, int world, int strip_rows)
{
    if (!out) return fail(nullptr, RTGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (width <= 0 || height <= 0 || world < 1 || rank < 0 || rank >= world || strip_rows <= 0 || (strip_rows % 8) != 0)
        return fail(nullptr, RTGL_ERR_INVALID, "rtgl_create: bad size / tiling arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RTGL_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(nullptr, RTGL_ERR_INVALID, "device ordinal out of range");
    rtgl_context *ctx = new rtgl_context();
    ctx->device = device; ctx->width = width; ctx->height = height;
    ctx->rank = rank; ctx->world = world; ctx->strip_rows = strip_rows;
    ctx->local_rows = local_rows_for(height, rank, world, strip_rows);
    auto bail = [&](int code) { g_create_error = ctx->error; rtgl_destroy(ctx); return code; };
#define CCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    ctx->error = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(RTGL_ERR_DEVICE); } } while (0)
    CCHK(hipSetDevice(device));
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cus = prop.multiProcessorCount; }
    CCHK(acquire_device_stream(device, &ctx->own_stream, &ctx->own_stream_shared));
    ctx->stream = ctx->own_stream;
    register_ctx_stream(ctx, device, ctx->stream);
    CCHK(hipEventCreate(&ctx->ev0));
    CCHK(hipEventCreate(&ctx->ev1));
    size_t img_bytes = (size_t)std::max(ctx->local_rows, 1) * width * sizeof(float4);
    CCHK(hipMalloc((void **)&ctx->d_image_own, img_bytes));
    CCHK(hipMemsetAsync(ctx->d_image_own, 0, img_bytes, ctx->stream));
    ctx->d_image = ctx->d_image_own;
    CCHK(hipMalloc((void **)&ctx->d_counters, 64));
    CCHK(hipMemsetAsync(ctx->d_counters, 0, 64, ctx->stream));
    CCHK(hipStreamSynchronize(ctx->stream));
#undef CCHK
    // operational override of the default scan without touching the caller: RTGL_AMD_KERNEL=0, 1, 2 or 4 (rtgl_set_option still wins)
    if (const char *k = getenv("RTGL_AMD_KERNEL")) { const int v = atoi(k); if (v >= RTGL_KERNEL_MEGA && v <= RTGL_KERNEL_WAVEFRONT_MFMA_SOLO && v != RTGL_KERNEL_REMOVED_3) { ctx->opt_kernel = v; ctx->kernel_explicit = true; } }
    if (const char *k = getenv("RTGL_AMD_SCAN_WAVES")) { const int v = atoi(k); if (v >= 0 && v <= 2) ctx->opt_scan_waves = v; }   // A/B of the scan's occupancy
    if (const char *k = getenv("RTGL_AMD_FRAME_BATCH")) { const int v = atoi(k); if (v >= 1 && v <= (int)kBatchMax) ctx->opt_frame_batch = v; }
    if (const char *k = getenv("RTGL_AMD_SCAN_DYNAMIC")) { const int v = atoi(k); if (v >= 0 && v <= 4) ctx->opt_scan_dynamic = v; }   // ... and of its work distribution
    *out = ctx;
    return RTGL_OK;
}

static int flush_pending_noexcept(rtgl_context *ctx);
extern "C" int rtgl_create(rtgl_context **out, int width, int height, int device)
{
    return rtgl_create_tiled(out, width, height, device, 0, 1, 8);
}

extern "C" void rtgl_destroy(rtgl_context *ctx)
{
    if (!ctx) return;
    // frames a batching context still holds back are submitted, not dropped (a caller that only ever called rtgl_render_frame and then
    // reads a bound device image after destroying the context would otherwise lose up to frame_batch - 1 frames)
    if (!ctx->pending.empty() && hipSetDevice(ctx->device) == hipSuccess && flush_pending_noexcept(ctx) != RTGL_OK)
        fprintf(stderr, "rtgl_destroy: %zu batched frame(s) could not be submitted: %s\n", ctx->pending.size(), ctx->error.c_str());
    unregister_ctx_stream(ctx);
    for (auto &w : ctx->workers) {
        { std::lock_guard<std::mutex> lk(w->m); w->state = PartWorker::kQuit; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    ctx->workers.clear();
    for (rtgl_context *part : ctx->parts) rtgl_destroy(part);
    ctx->parts.clear();
    (void)hipSetDevice(ctx->device);
    for (hipEvent_t e : ctx->part_done) (void)hipEventDestroy(e);
    if (ctx->gather_done) (void)hipEventDestroy(ctx->gather_done);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
#ifdef RT_SOLO_STAMPS
    if (ctx->d_dbg_log) {      // diagnostics build: where the waves of the solo scan spent their cycles, per bounce, summed over all frames
        unsigned long long h[16 * 64];
        if (hipMemcpy(h, ctx->d_dbg_log, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            for (int b = 0; b < 64 && h[16 * b + 8]; ++b) {
                const unsigned long long *d = h + 16 * b; const double tot = (double)d[0];
                fprintf(stderr, "rtgl stamps bounce %2d: waves %llu iters %llu  cycles/wave %.0f  staging %.1f%% rays %.1f%% group+prologue %.1f%% steady %.1f%% (park %.1f%%) flush %.1f%%  cycles per tile-stage in steady %.1f  slowest wave of a launch (mean over launches) %.0f  culled items %llu at %.0f cycles\n",
                        b, d[8], d[7], tot / d[8], 100.0 * d[1] / tot, 100.0 * d[2] / tot, 100.0 * d[3] / tot, 100.0 * d[4] / tot, 100.0 * d[5] / tot, 100.0 * d[6] / tot,
                        d[9] ? (double)d[4] / (double)d[9] : 0.0, (double)d[13] * (double)d[15] / (double)d[8], d[12], d[12] ? (double)d[11] / (double)d[12] : 0.0);
            }
#if RT_SOLO_STAMPS == 3
        if (hipMemcpy(h, ctx->d_dbg_log, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            double cyc = 0, ticks = 0;
            for (int b = 0; b < 64 && h[16 * b + 8]; ++b) {
                const unsigned long long *d = h + 16 * b;
                fprintf(stderr, "rtgl clock bounce %2d: %llu waves, %.0f shader cycles per wave over %.0f ticks of 10 ns: %.3f GHz\n", b, d[8], (double)d[0] / d[8], (double)d[14] / d[8], d[14] ? (double)d[0] / (double)d[14] * 0.1 : 0.0);
                cyc += (double)d[0]; ticks += (double)d[14];
            }
            if (ticks > 0) fprintf(stderr, "rtgl clock: in-kernel shader clock of scan_solo_kernel, wave-time weighted over all launches: %.3f GHz\n", cyc / ticks * 0.1);
        }
#endif
        // static launches: mean wave time per chunk, relative to the bounce's mean (what an uneven cost of the chunks loses)
        std::vector<unsigned long long> pc(16 * 64 * 2 * 16);
        if (hipMemcpy(pc.data(), reinterpret_cast<unsigned long long *>(ctx->d_dbg_log) + 2048, pc.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            for (int b = 0; b < 8; ++b) {
                double tot = 0, cnt = 0; int nch = 0;
                for (int c = 0; c < 64; ++c) { const unsigned long long *e = pc.data() + ((size_t)16 * b * 64 + c) * 2; if (e[1]) { tot += (double)e[0]; cnt += (double)e[1]; nch = c + 1; } }
                if (cnt == 0) continue;
                fprintf(stderr, "rtgl stamps bounce %2d: mean wave time per chunk / bounce mean:", b);
                for (int c = 0; c < nch; ++c) { const unsigned long long *e = pc.data() + ((size_t)16 * b * 64 + c) * 2; fprintf(stderr, " %.2f", e[1] ? ((double)e[0] / (double)e[1]) / (tot / cnt) : 0.0); }
                fprintf(stderr, "\n");
            }
    }
#endif
    void *ptrs[] = { ctx->d_spheres, ctx->d_materials, ctx->d_vertices, ctx->d_sphere_visits, ctx->d_edges, ctx->d_planes,
                     ctx->d_env, ctx->d_image_own, ctx->d_rng, ctx->d_counters, ctx->d_u8, ctx->d_group_bounds, ctx->d_wave, ctx->d_counts, ctx->d_mf_groups, ctx->d_mf_A, ctx->d_mf_order,
                     ctx->d_dbg_log, ctx->d_cand, ctx->d_keep0, ctx->d_plan, ctx->d_stage, ctx->d_sort_hist, ctx->d_mf_cull, ctx->d_keep, ctx->d_items, ctx->d_sched, ctx->d_edges_s, ctx->d_planes_s, ctx->d_batch_rad };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : ctx->kev) (void)hipEventDestroy(e);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->counts_ev) (void)hipEventDestroy(ctx->counts_ev);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) release_device_stream(ctx->device, ctx->own_stream, ctx->own_stream_shared);
    delete ctx;
}

extern "C" const char *rtgl_last_error(const rtgl_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

static int flush_pending(rtgl_context *ctx);
#define ENTER_NOFLUSH(ctx) do { if (!(ctx)) return RTGL_ERR_INVALID; HIPCHK(ctx, hipSetDevice((ctx)->device)); } while (0)
// every entry point but rtgl_render_frame / rtgl_set_frame_params first submits the frames a batching context is still holding back
#define ENTER(ctx) do { ENTER_NOFLUSH(ctx); if (!(ctx)->pending.empty()) { const int rcf_ = flush_pending(ctx); if (rcf_) return rcf_; } } while (0)
// multi-device context: run `call` on every part, report the first failure through the assembler
#define FANOUT(ctx, call) do { if (!(ctx)->parts.empty()) { (ctx)->gathered = false; \
    for (rtgl_context *part : (ctx)->parts) { const int rc_ = (call); if (rc_) return fail(ctx, rc_, std::string("device ") + std::to_string(part->device) + ": " + part->error); } \
    return RTGL_OK; } } while (0)

// ---- single-process multi-device context (create(w, h, devices[], n)) --------------------------------------------
// Replaces the ONE glDispatchCompute of the reference (src/renderer.cpp:129-134) by one strip-tiled dispatch per device; the tile
// buffers are gathered to devices[0] over xGMI (peer copies, one 2-D copy per part: strips of a part are `world` strips apart in
// the assembled image) when the image is consumed.  The same device may be listed more than once (several contexts on one GPU).
extern "C" int rtgl_create_multi(rtgl_context **out, int width, int height, const int *devices, int n_devices, int strip_rows)
{
    if (!out) return fail(nullptr, RTGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return fail(nullptr, RTGL_ERR_INVALID, "rtgl_create_multi: need 1..64 device ordinals");
    rtgl_context *ctx = nullptr;
    int rc = rtgl_create_tiled(&ctx, width, height, devices[0], 0, 1, strip_rows);         // the assembler: full image on devices[0]
    if (rc) return rc;
    for (int i = 0; i < n_devices; ++i) {
        rtgl_context *part = nullptr;
        rc = rtgl_create_tiled(&part, width, height, devices[i], i, n_devices, strip_rows);
        if (rc) { const std::string msg = g_create_error; rtgl_destroy(ctx); g_create_error = msg; return rc; }
        // (parts that share a device share its stream, like all contexts of the process: acquire_device_stream)
        ctx->parts.push_back(part);
        hipEvent_t ev = nullptr;
        (void)hipSetDevice(devices[i]);
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { rtgl_destroy(ctx); return fail(nullptr, RTGL_ERR_DEVICE, "hipEventCreate failed"); }
        ctx->part_done.push_back(ev);
        if (devices[i] != devices[0]) {           // xGMI peer copies; without peer access the gather goes through the host
            int can = 0;
            (void)hipSetDevice(devices[0]);
            if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) != hipSuccess || !can) ctx->peer_copy = false;
            else { hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ctx->peer_copy = false; }
            (void)hipGetLastError();
        }
    }
    const char *mt = getenv("RTGL_AMD_MULTI_THREADS");
    if (n_devices > 1 && !(mt && atoi(mt) == 0)) try {
        for (rtgl_context *part : ctx->parts) {
            ctx->workers.emplace_back(new PartWorker);
            PartWorker *w = ctx->workers.back().get();
            w->part = part;
            w->th = std::thread([w] {
                std::unique_lock<std::mutex> lk(w->m);
                for (;;) {
                    w->cv.wait(lk, [w] { return w->state == PartWorker::kJob || w->state == PartWorker::kQuit; });
                    if (w->state == PartWorker::kQuit) return;
                    lk.unlock();
                    const int rc = rtgl_render_frame(w->part);      // (sets the thread's device itself: ENTER)
                    lk.lock();
                    w->rc = rc; w->state = PartWorker::kDone;
                    w->cv.notify_all();
                }
            });
        }
    } catch (const std::exception &) {                   // no threads to be had: the caller's thread submits, as with RTGL_AMD_MULTI_THREADS=0
        for (auto &w : ctx->workers) {
            { std::lock_guard<std::mutex> lk(w->m); w->state = PartWorker::kQuit; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
        ctx->workers.clear();
    }
    *out = ctx;
    return RTGL_OK;
}

extern "C" int rtgl_device_count(const rtgl_context *ctx) { return ctx ? (ctx->parts.empty() ? 1 : (int)ctx->parts.size()) : RTGL_ERR_INVALID; }

// strips of every part -> the assembler's image (global row order), on the assembler's stream, after each part's queued frames
static int multi_gather(rtgl_context *ctx)
{
    if (ctx->gathered) return RTGL_OK;
    const int world = (int)ctx->parts.size(), sr = ctx->strip_rows;
    const size_t row_bytes = (size_t)ctx->width * 16, strip_bytes = row_bytes * sr;
    std::vector<float> host;
    for (int i = 0; i < world; ++i) {
        rtgl_context *part = ctx->parts[i];
        if (!part->pending.empty()) { HIPCHK(ctx, hipSetDevice(part->device)); const int rc = flush_pending(part); if (rc) return fail(ctx, rc, part->error); }
        const int full = part->local_rows / sr, tail_rows = part->local_rows - full * sr;       // only the owner of the last strip has a short one
        uint8_t *dst = reinterpret_cast<uint8_t *>(ctx->d_image) + (size_t)i * strip_bytes;
        const uint8_t *src = reinterpret_cast<const uint8_t *>(part->d_image);
        if (ctx->peer_copy) {
            HIPCHK(ctx, hipSetDevice(part->device));
            HIPCHK(ctx, hipEventRecord(ctx->part_done[i], part->stream));
            HIPCHK(ctx, hipSetDevice(ctx->device));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->part_done[i], 0));
            if (full) HIPCHK(ctx, hipMemcpy2DAsync(dst, (size_t)world * strip_bytes, src, strip_bytes, strip_bytes, (size_t)full, hipMemcpyDeviceToDevice, ctx->stream));
            if (tail_rows) HIPCHK(ctx, hipMemcpyAsync(dst + (size_t)full * world * strip_bytes, src + (size_t)full * strip_bytes, row_bytes * tail_rows, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            host.resize((size_t)part->local_rows * ctx->width * 4);
            int rc = rtgl_read_image_f32(part, host.data());
            if (rc) return fail(ctx, rc, part->error);
            HIPCHK(ctx, hipSetDevice(ctx->device));
            if (full) HIPCHK(ctx, hipMemcpy2DAsync(dst, (size_t)world * strip_bytes, host.data(), strip_bytes, strip_bytes, (size_t)full, hipMemcpyHostToDevice, ctx->stream));
            if (tail_rows) HIPCHK(ctx, hipMemcpyAsync(dst + (size_t)full * world * strip_bytes, reinterpret_cast<const uint8_t *>(host.data()) + (size_t)full * strip_bytes, row_bytes * tail_rows, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->peer_copy) {
        // the copies read every part's tile buffer on the ASSEMBLER's stream: the parts' next frames (their own streams) must not overwrite
        // the tiles before the copies are through
        if (!ctx->gather_done) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->gather_done, hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(ctx->gather_done, ctx->stream));
        for (rtgl_context *part : ctx->parts)
            if (part->stream != ctx->stream) {
                HIPCHK(ctx, hipSetDevice(part->device));
                HIPCHK(ctx, hipStreamWaitEvent(part->stream, ctx->gather_done, 0));
            }
        HIPCHK(ctx, hipSetDevice(ctx->device));
    }
    ctx->gathered = true;
    return RTGL_OK;
}

extern "C" int rtgl_gather_tiles(rtgl_context *ctx)
{
    ENTER(ctx);
    if (ctx->parts.empty()) return RTGL_OK;            // a single-device context holds its image already
    return multi_gather(ctx);
}

extern "C" int rtgl_upload_spheres(rtgl_context *ctx, const void *data, uint32_t count)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_upload_spheres(part, data, count));
    if (count && !data) return fail(ctx, RTGL_ERR_INVALID, "spheres is NULL");
    static_assert(sizeof(SphereRec) == 32, "Sphere stride (shaders/raytracer.glsl:11-15, std140)");
    int rc = realloc_upload(ctx, ctx->d_spheres, data, (size_t)count * 32);
    if (rc) return rc;
    ctx->n_spheres = count; ctx->visits_dirty = true;
    return RTGL_OK;
}

extern "C" int rtgl_upload_materials(rtgl_context *ctx, const void *data, uint32_t count)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_upload_materials(part, data, count));
    if (count && !data) return fail(ctx, RTGL_ERR_INVALID, "materials is NULL");
    static_assert(sizeof(MaterialRec) == 32, "Material stride (shaders/raytracer.glsl:17-21, std140)");
    int rc = realloc_upload(ctx, ctx->d_materials, data, (size_t)count * 32);
    if (rc) return rc;
    ctx->n_materials = count;
    return RTGL_OK;
}

extern "C" int rtgl_upload_meshes(rtgl_context *ctx, const void *data, uint32_t count)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_upload_meshes(part, data, count));
    if (count && !data) return fail(ctx, RTGL_ERR_INVALID, "meshes is NULL");
    ctx->h_meshes.assign((const uint8_t *)data, (const uint8_t *)data + (size_t)count * 16);
    ctx->n_meshes = count; ctx->tris_dirty = true;
    return RTGL_OK;
}

extern "C" int rtgl_upload_vertices(rtgl_context *ctx, const void *data, uint32_t vec4_count)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_upload_vertices(part, data, vec4_count));
    if (vec4_count && !data) return fail(ctx, RTGL_ERR_INVALID, "vertices is NULL");
    int rc = realloc_upload(ctx, ctx->d_vertices, data, (size_t)vec4_count * 16);
    if (rc) return rc;
    ctx->h_vertices.assign((const uint8_t *)data, (const uint8_t *)data + (size_t)vec4_count * 16);   // host copy: spatial ordering of the triangles
    ctx->n_vec4 = vec4_count; ctx->tris_dirty = true;
    return RTGL_OK;
}

extern "C" int rtgl_upload_nodes(rtgl_context *ctx, const void *data, uint32_t count)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_upload_nodes(part, data, count));
    if (count && !data) return fail(ctx, RTGL_ERR_INVALID, "nodes is NULL");
    ctx->h_nodes.assign((const uint8_t *)data, (const uint8_t *)data + (size_t)count * 48);
    ctx->n_nodes = count; ctx->visits_dirty = true;
    return RTGL_OK;
}

extern "C" int rtgl_upload_envmap(rtgl_context *ctx, const uint8_t *faces, int nfaces, int width, int height, int channels)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_upload_envmap(part, faces, nfaces, width, height, channels));
    if (nfaces < 0 || nfaces > 6 || width <= 0 || height <= 0 || (channels != 3 && channels != 4) || (nfaces && !faces))
        return fail(ctx, RTGL_ERR_INVALID, "envmap: need 0..6 faces, positive size, 3 or 4 channels");
    int rc = realloc_upload(ctx, ctx->d_env, faces, (size_t)nfaces * width * height * channels);
    if (rc) return rc;
    ctx->env_w = width; ctx->env_h = height; ctx->env_c = channels; ctx->env_faces = nfaces;
    return RTGL_OK;
}

// traverse() (:272-329) walks the node buffer identically for every ray; run that walk once here and
// keep the sphere indices in test order.  Stack of 5 with silently dropped pushes (:113-121), nodes
// outside the buffer are childless and empty, walk capped at 65535 pops.
static int rebuild_sphere_visits(rtgl_context *ctx)
{
    std::vector<uint32_t> visits;
    const size_t kMaxVisits = 1u << 20;
    if (ctx->n_nodes > 0) {
        uint32_t items[5] = { 0, 0, 0, 0, 0 };
        int top = 0, pops = 0;
        auto rd = [&](uint32_t node, int off) { uint32_t v; memcpy(&v, ctx->h_nodes.data() + (size_t)node * 48 + off, 4); return v; };
        while (top != -1 && pops < 65535) {
            uint32_t id = items[top--];
            pops++;
            uint32_t left = 0xFFFFFFFFu, right = 0xFFFFFFFFu, offset = 0, count = 0;
            if (id < ctx->n_nodes) { left = rd(id, 32); right = rd(id, 36); offset = rd(id, 40); count = rd(id, 44); }
            if (left != 0xFFFFFFFFu && top != 4) items[++top] = left;
            if (right != 0xFFFFFFFFu && top != 4) items[++top] = right;
            uint64_t end = (uint64_t)offset + count;
            bool zero_emitted = false;
            for (uint64_t i = offset; i < end; ++i) {
                if (i < ctx->n_spheres) visits.push_back((uint32_t)i);
                else {   // every out-of-range index reads the same all-zero sphere: one visit stands for the run
                    if (!zero_emitted) visits.push_back(kNoSphere);
                    zero_emitted = true;
                    break;
                }
                if (visits.size() > kMaxVisits) return fail(ctx, RTGL_ERR_INVALID, "node buffer expands to more than 2^20 sphere tests per ray");
            }
        }
    }
    int rc = realloc_upload(ctx, ctx->d_sphere_visits, visits.data(), visits.size() * sizeof(uint32_t));
    if (rc) return rc;
    ctx->n_sphere_visits = (uint32_t)visits.size();
    ctx->visits_dirty = false;
    return RTGL_OK;
}

// Storage order of the triangle visits for the matrix-core broad phase: the leaves of a k-d tree (median split along the longest axis),
// written out left to right.  A leaf is one MFMA tile (10 triangles); the split positions are multiples of the unit above them (tile ->
// quad of 4 tiles -> group of `group_tris` triangles), so every tile, quad and group of the storage order is one subtree.
//   * down to the groups the tree splits the triangle CENTROIDS: a group shares one local origin and one set of bounds in the bf16
//     broad phase (rt_mfma.hpp), whose margin grows with the group's extent;
//   * inside a group it splits the six-dimensional points (centroid, lambda x unit normal), lambda = 0.4 x the mesh's extent: a
//     tile's culling record (rt_mfma.hpp, MfCull) bounds its ten normals by a rectangle, and on a mesh that is coarse against its own
//     curvature (the 100,000-triangle benchmark field turns by 15 degrees from one cell to the next) ten NEIGHBOURS spread +-33 degrees --
//     a quarter of all far tiles then fail the certificates on their normals alone.  Ten triangles of similar slope from anywhere in
//     the group (3 units across there) spread +-9 degrees; that the tile's sphere grows from 0.5 to 1.5 units costs far less
//     (emulated on real bounce-1 rays, tools/diagnostics/cull_emulation.py: 49 -> 64 % of the tile tests certified at 100k triangles,
//     88 -> 97 % on its camera bounce; 67 -> 70 % and 99.6 -> 98.9 % at 10k).
// Only tightness depends on the order: hits merge by VISIT index.  Non-finite centroids and degenerate normals sort as 0.
// Deterministic: ties break by visit index.
static std::vector<uint32_t> kd_order(const rtgl_context *ctx, const std::vector<uint32_t> &visit_tri, uint32_t group_tris)
{
    const size_t n = visit_tri.size();
    const float *vx = reinterpret_cast<const float *>(ctx->h_vertices.data());
    std::vector<float> pts(6 * n);
    float blo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    auto finite = [](float c) { return c == c && c > -1.0e30f && c < 1.0e30f; };
    for (size_t v = 0; v < n; ++v) {
        const float *t = vx + (size_t)visit_tri[v] * 12;
        for (int a = 0; a < 3; ++a) {
            const float c = (t[a] + t[4 + a] + t[8 + a]) * (1.0f / 3.0f);
            pts[6 * v + a] = finite(c) ? c : 0.0f;
            if (finite(c)) { blo[a] = std::min(blo[a], c); bhi[a] = std::max(bhi[a], c); }
        }
        const float e1[3] = {t[4] - t[0], t[5] - t[1], t[6] - t[2]}, e2[3] = {t[8] - t[0], t[9] - t[1], t[10] - t[2]};
        const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        const float nl = std::sqrt(nx * nx + ny * ny + nz * nz);
        const bool ok = finite(nl) && nl > 0.0f;
        pts[6 * v + 3] = ok ? nx / nl : 0.0f; pts[6 * v + 4] = ok ? ny / nl : 0.0f; pts[6 * v + 5] = ok ? nz / nl : 0.0f;
    }
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) if (blo[a] <= bhi[a]) ext = std::max(ext, bhi[a] - blo[a]);
    float lambda = 0.4f * ext;
    if (const char *e = getenv("RTGL_AMD_KD_LAMBDA")) lambda = (float)atof(e) * ext;      // (tuning: fraction of the mesh's extent)
    for (size_t v = 0; v < n; ++v) for (int a = 3; a < 6; ++a) pts[6 * v + a] *= lambda;
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::vector<std::pair<size_t, size_t>> todo;
    if (n) todo.emplace_back(0, n);
    while (!todo.empty()) {
        const size_t lo = todo.back().first, hi = todo.back().second, m = hi - lo;
        todo.pop_back();
        if (m <= (size_t)kMfTileTris) continue;
        const size_t unit = m > group_tris ? group_tris : (m > (size_t)kMfQuadTris ? (size_t)kMfQuadTris : (size_t)kMfTileTris);
        const size_t nl = ((m / 2 + unit - 1) / unit) * unit;           // in [unit, m): m > unit
        const int dims = m > group_tris ? 3 : 6;
        float bl[6], bh[6];
        for (int a = 0; a < dims; ++a) { bl[a] = 3.0e38f; bh[a] = -3.0e38f; }
        for (size_t i = lo; i < hi; ++i)
            for (int a = 0; a < dims; ++a) { const float c = pts[6 * (size_t)order[i] + a]; bl[a] = std::min(bl[a], c); bh[a] = std::max(bh[a], c); }
        int ax = 0;
        for (int a = 1; a < dims; ++a) if (bh[a] - bl[a] > bh[ax] - bl[ax]) ax = a;
        std::nth_element(order.begin() + lo, order.begin() + lo + nl, order.begin() + hi, [&](uint32_t p, uint32_t q) {
            const float cp = pts[6 * (size_t)p + ax], cq = pts[6 * (size_t)q + ax];
            return cp < cq || (cp == cq && p < q);
        });
        todo.emplace_back(lo, lo + nl);
        todo.emplace_back(lo + nl, hi);
    }
    return order;
}

static int rebuild_triangles(rtgl_context *ctx)
{
    std::vector<uint32_t> visit_tri;
    uint32_t n_tris = ctx->n_vec4 / 3;   // to_triangles() drops a trailing partial triangle (src/renderer.h:34-48)
    for (uint32_t m = 0; m < ctx->n_meshes; ++m) {
        uint32_t start, size;
        memcpy(&start, ctx->h_meshes.data() + (size_t)m * 16, 4);
        memcpy(&size, ctx->h_meshes.data() + (size_t)m * 16 + 4, 4);
        uint64_t end = std::min<uint64_t>((uint64_t)start + size, n_tris);
        for (uint64_t t = start; t < end; ++t) visit_tri.push_back((uint32_t)t);
    }
    if (ctx->d_edges) { HIPCHK(ctx, hipFree(ctx->d_edges)); ctx->d_edges = nullptr; }
    if (ctx->d_planes) { HIPCHK(ctx, hipFree(ctx->d_planes)); ctx->d_planes = nullptr; }
    ctx->n_tri_visits = (uint32_t)visit_tri.size();
    ctx->scene_version++;
    {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        const float *vx = reinterpret_cast<const float *>(ctx->h_vertices.data());
        for (uint32_t t : visit_tri)
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 3; ++a) { const float c = vx[(size_t)t * 12 + 4 * k + a]; if (c == c && c > -1.0e30f && c < 1.0e30f) { lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); } }
        ctx->mesh_ext = 0.0f;
        for (int a = 0; a < 3; ++a) { ctx->mesh_lo[a] = lo[a] <= hi[a] ? lo[a] : 0.0f; ctx->mesh_hi[a] = lo[a] <= hi[a] ? hi[a] : 0.0f; if (lo[a] <= hi[a]) ctx->mesh_ext = std::max(ctx->mesh_ext, hi[a] - lo[a]); }
    }
    if (ctx->n_tri_visits) {
        uint32_t *d_visit = nullptr;
        HIPCHK(ctx, hipMalloc((void **)&d_visit, visit_tri.size() * 4));
        HIPCHK(ctx, hipMemcpyAsync(d_visit, visit_tri.data(), visit_tri.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_edges, (size_t)ctx->n_tri_visits * sizeof(TriEdges)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)ctx->n_tri_visits * sizeof(TriPlane)));
        dim3 grid((ctx->n_tri_visits + 255) / 256);
        hipLaunchKernelGGL(prepare_triangles_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_vertices, d_visit, ctx->n_tri_visits, ctx->d_edges, ctx->d_planes);
        HIPCHK(ctx, hipGetLastError());
        if (ctx->d_group_bounds) { HIPCHK(ctx, hipFree(ctx->d_group_bounds)); ctx->d_group_bounds = nullptr; }
        uint32_t n_groups = (ctx->n_tri_visits + kBoundGroup - 1) / kBoundGroup;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_group_bounds, (size_t)n_groups * sizeof(float2)));
        hipLaunchKernelGGL(group_bounds_kernel, dim3(n_groups), dim3(64), 0, ctx->stream, ctx->d_edges, ctx->n_tri_visits, ctx->d_group_bounds);
        HIPCHK(ctx, hipGetLastError());
        // bf16 broad-phase data: local origins, bounds, A matrices (rt_mfma.hpp)
        if (ctx->d_mf_groups) { HIPCHK(ctx, hipFree(ctx->d_mf_groups)); ctx->d_mf_groups = nullptr; }
        if (ctx->d_mf_A) { HIPCHK(ctx, hipFree(ctx->d_mf_A)); ctx->d_mf_A = nullptr; }
        if (ctx->d_mf_order) { HIPCHK(ctx, hipFree(ctx->d_mf_order)); ctx->d_mf_order = nullptr; }
        if (ctx->d_mf_cull) { HIPCHK(ctx, hipFree(ctx->d_mf_cull)); ctx->d_mf_cull = nullptr; }
        if (ctx->d_edges_s) { HIPCHK(ctx, hipFree(ctx->d_edges_s)); ctx->d_edges_s = nullptr; }
        if (ctx->d_planes_s) { HIPCHK(ctx, hipFree(ctx->d_planes_s)); ctx->d_planes_s = nullptr; }
        // quads sharing one local origin: 32 (= a chunk: one ray set-up per work item of the scan) unless the caller chose.  Smaller
        // groups have tighter bounds and fewer survivors (C4: 89 M per frame at 8 quads against 111 M at 32), but every group of a
        // chunk costs the scan a ray set-up and a pipeline fill of its own: 28.4 against 30.0 Mpaths/s
        ctx->mf_group_quads = ctx->group_explicit ? (uint32_t)ctx->opt_mf_group_quads : 32u;
        const uint32_t group_tris = ctx->mf_group_quads * kMfQuadTris;
        ctx->n_mf_groups = (ctx->n_tri_visits + group_tris - 1) / group_tris;
        const std::vector<uint32_t> order = kd_order(ctx, visit_tri, group_tris);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_mf_order, order.size() * 4));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_mf_order, order.data(), order.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_edges_s, (size_t)ctx->n_tri_visits * sizeof(TriEdges)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_planes_s, (size_t)ctx->n_tri_visits * sizeof(TriPlane)));
        hipLaunchKernelGGL(gather_storage_order_kernel, dim3((ctx->n_tri_visits + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_edges, ctx->d_planes, ctx->d_mf_order,
                           ctx->n_tri_visits, ctx->d_edges_s, ctx->d_planes_s);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_mf_groups, (size_t)ctx->n_mf_groups * sizeof(MfGroup)));
        const size_t a_bytes = ((size_t)ctx->n_mf_groups * ctx->mf_group_quads + 1) * kMfQuadTiles * 64 * sizeof(uint4);   // two K panels per tile; + one zero quad
        if (a_bytes > 0xFFFF0000ull) return fail(ctx, RTGL_ERR_INVALID, "mesh too large for the 32-bit tile offsets of the matrix-core scan");
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_mf_A, a_bytes));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_mf_A, 0, a_bytes, ctx->stream));
        hipLaunchKernelGGL(prepare_mfma_kernel, dim3(ctx->n_mf_groups), dim3(64), 0, ctx->stream, ctx->d_vertices, d_visit,
                           ctx->d_mf_order, ctx->n_tri_visits, ctx->n_mf_groups, ctx->mf_group_quads, ctx->d_mf_groups, ctx->d_mf_A, getenv("RTGL_AMD_ROW_GAMMA") ? (float)atof(getenv("RTGL_AMD_ROW_GAMMA")) : 1.220703125e-4f);
        HIPCHK(ctx, hipGetLastError());
        // packet-culling records, one per tile of the storage order (all-zero records -- unusable -- behind the last one)
        const uint32_t n_tiles_all = ctx->n_mf_groups * ctx->mf_group_quads * (uint32_t)kMfQuadTiles, n_tiles_alloc = n_tiles_all + 128u;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_mf_cull, (size_t)n_tiles_alloc * sizeof(MfCull)));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_mf_cull, 0, (size_t)n_tiles_alloc * sizeof(MfCull), ctx->stream));
        hipLaunchKernelGGL(prepare_cull_kernel, dim3(n_tiles_all), dim3(64), 0, ctx->stream, ctx->d_vertices, d_visit,
                           ctx->d_mf_order, ctx->n_tri_visits, n_tiles_all, ctx->d_mf_cull);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(d_visit));
    } else ctx->n_mf_groups = 0;
    ctx->tris_dirty = false;
    return RTGL_OK;
}

// event pair around one dominant-kernel launch (only with option kernel_timing)
static void kev_mark(rtgl_context *ctx)
{
    if (!ctx->timing_this_frame) return;
    if (ctx->kev_used == ctx->kev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ctx->kev.push_back(e); }
    (void)hipEventRecord(ctx->kev[ctx->kev_used++], ctx->stream);
}

// ---- bounce-wavefront pipeline: buffers + launches ------------------------------------------------
// kernel 4 launches at most max(CUs, chunks) blocks of four waves; each wave owns one region of the candidate buffer
static uint32_t solo_chunks(const rtgl_context *ctx);
static uint32_t solo_regions(const rtgl_context *ctx) { return std::max<uint32_t>((uint32_t)ctx->n_cus, solo_chunks(ctx)) * 8u; }   // (8 waves per block with two waves per SIMD)

static size_t counts_bytes(uint32_t capacity) { return (size_t)(capacity + 1u) * sizeof(uint32_t); }      // ray counts per bounce + the fullest candidate region

// The origin word of the bin key (rt_wavefront.hpp, ray_bin_key): 3 sort_ob - 1 cell bits behind the outside flag.  Inside: every bit goes to
// the axis whose cells are still the longest (a degenerate axis gets the last ones, which then carry nothing); outside: one bit less for
// the axis the mesh is thinnest on.
static void set_bin_cells(rtgl_context *ctx)
{
    WaveBuffers &wb = ctx->wb;
    const uint32_t T = wb.sort_T;
    const float ext_max = ctx->mesh_ext > 0.0f ? ctx->mesh_ext * 1.02f : 1.0f;
    float ext[3]; uint32_t bits[3] = {0, 0, 0}, order = 0;
    for (int a = 0; a < 3; ++a) ext[a] = std::max((ctx->mesh_hi[a] - ctx->mesh_lo[a]) + 0.02f * ext_max, 1.0e-6f * ext_max);
    for (uint32_t i = 0; i < T; ++i) {
        int best = 0;
        for (int a = 1; a < 3; ++a) if (ext[a] / (float)(1u << bits[a]) > ext[best] / (float)(1u << bits[best])) best = a;
        if (bits[best] >= 10u) for (int a = 0; a < 3; ++a) if (bits[a] < bits[best]) best = a;
        bits[best]++; order |= (uint32_t)best << (2u * i);
    }
    float cell_max = 0.0f;
    for (int a = 0; a < 3; ++a) {
        wb.sort_lo[a] = ctx->mesh_lo[a] - 0.01f * ext_max;
        wb.sort_inv_cell[a] = (float)(1u << bits[a]) / ext[a];
        if (bits[a]) cell_max = std::max(cell_max, ext[a] / (float)(1u << bits[a]));
    }
    for (int a = 0; a < 3; ++a) { wb.sort_in_lo[a] = wb.sort_lo[a] - cell_max; wb.sort_in_hi[a] = wb.sort_lo[a] + ext[a] + cell_max; wb.sort_cen[a] = wb.sort_lo[a] + 0.5f * ext[a]; }
    if (const char *e = getenv("RTGL_AMD_BIN_SPLIT")) if (atoi(e) == 0) for (int a = 0; a < 3; ++a) { wb.sort_in_lo[a] = -INFINITY; wb.sort_in_hi[a] = INFINITY; }      // (tuning: every origin clamped into the box's cells)
    wb.sort_in_bits = bits[0] | (bits[1] << 4) | (bits[2] << 8); wb.sort_in_order = order;
    wb.sort_inv_unit = 8.0f / ext_max;                            // the first outside cell ends a quarter of the half extent from the centre
    if (const char *e = getenv("RTGL_AMD_BIN_UNIT")) { const float v = (float)atof(e); if (v > 0.0f) wb.sort_inv_unit = v / ext_max; }      // (tuning)
    int thin = 0;
    for (int a = 1; a < 3; ++a) if (ext[a] < ext[thin]) thin = a;
    uint32_t ob[3] = {0, 0, 0}, oorder = 0;
    for (uint32_t i = 0; i < T; ++i) {                            // round robin, the thin axis last
        int best = -1;
        for (int k = 0; k < 3; ++k) { const int a = (thin + 1 + k) % 3; if (best < 0 || ob[a] < ob[best]) best = a; }
        if (ob[best] >= 4u) break;                                // (16 cells per axis at most: T <= 12 here)
        ob[best]++; oorder |= (uint32_t)best << (2u * i);
    }
    wb.sort_out_bits = ob[0] | (ob[1] << 4) | (ob[2] << 8); wb.sort_out_order = oorder;
}

static int ensure_wave_buffers(rtgl_context *ctx, uint32_t n0, uint32_t max_bounce, bool multi_sample)
{
    if (ctx->counts_capacity < max_bounce + 2) {
        if (ctx->d_counts) { HIPCHK(ctx, hipFree(ctx->d_counts)); ctx->d_counts = nullptr; }
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_counts, counts_bytes(max_bounce + 2)));                          // u32 ray counts per bounce
        if (ctx->h_counts) { HIPCHK(ctx, hipHostFree(ctx->h_counts)); ctx->h_counts = nullptr; }
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_counts, counts_bytes(max_bounce + 2), hipHostMallocDefault));
        if (!ctx->counts_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->counts_ev, hipEventDisableTiming));
        ctx->counts_capacity = max_bounce + 2; ctx->counts_pending = ctx->counts_valid = false; ctx->est_counts.clear();
    }
    const size_t local_px = (size_t)std::max(ctx->local_rows, 1) * ctx->width;
    if (ctx->opt_kernel == RTGL_KERNEL_WAVEFRONT_MFMA_SOLO) {
        const uint32_t need_regions = solo_regions(ctx);
        // (a record = one (ray, 5-triangle mask); every queue entry of the scan makes four of them)
        if (!ctx->cand_region_target) ctx->cand_region_target = std::max<uint32_t>(4096u, (uint32_t)std::min<uint64_t>(((uint64_t)n0 + need_regions - 1) / need_regions, 0xFFFFFFF0u));
        if (!ctx->d_cand || need_regions > ctx->cand_regions || ctx->cand_region_target > ctx->cand_region_pairs) {
            if (ctx->d_cand) { HIPCHK(ctx, hipFree(ctx->d_cand)); ctx->d_cand = nullptr; }                  // (hipFree waits for the frames in flight)
            ctx->cand_regions = need_regions; ctx->cand_region_pairs = ctx->cand_region_target;
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_cand, ((size_t)need_regions * ctx->cand_region_pairs) * sizeof(uint2) + (size_t)need_regions * sizeof(uint32_t) + 256));
        }
        {                                                    // packet culling: one bit per (granule of 128 rays, tile); + 16 granules read ahead of the last one
            const uint32_t real_quads = std::min(ctx->n_mf_groups * ctx->mf_group_quads, (ctx->n_tri_visits + (uint32_t)kMfQuadTris - 1) / (uint32_t)kMfQuadTris);
            ctx->wb.keep_words = std::max(1u, (real_quads * (uint32_t)kMfQuadTiles + 31u) / 32u);
            const size_t need = ((size_t)n0 / 128 + 16) * ctx->wb.keep_words;
            if (ctx->keep_capacity < need) {
                if (ctx->d_keep) { HIPCHK(ctx, hipFree(ctx->d_keep)); ctx->d_keep = nullptr; }
                HIPCHK(ctx, hipMalloc((void **)&ctx->d_keep, need * sizeof(uint32_t)));
                ctx->keep_capacity = need;
            }
        }
        ctx->wb.keep = ctx->d_keep;
        {                                                    // work distribution of the scan: one counter per (bounce, chunk); a launch has at most max(CUs, chunks) chunks
            ctx->wb.sched_stride = std::max<uint32_t>((uint32_t)ctx->n_cus, solo_chunks(ctx));
            const size_t need = (size_t)(max_bounce + 2) * ctx->wb.sched_stride;
            if (ctx->sched_capacity < need) {
                if (ctx->d_sched) { HIPCHK(ctx, hipFree(ctx->d_sched)); ctx->d_sched = nullptr; }
                HIPCHK(ctx, hipMalloc((void **)&ctx->d_sched, need * sizeof(uint32_t)));
                ctx->sched_capacity = need;
            }
            ctx->wb.sched = ctx->d_sched;
        }
        if (ctx->opt_cull == 3) {                            // ray binning: staging queue (a, b, c, rng: 4 x 16 B, pixel 4 B, key + rank 8 B) and the bin counters
            if (ctx->stage_capacity < n0) {
                if (ctx->d_stage) { HIPCHK(ctx, hipFree(ctx->d_stage)); ctx->d_stage = nullptr; }
                HIPCHK(ctx, hipMalloc(&ctx->d_stage, (size_t)n0 * 76 + 1024));
                ctx->stage_capacity = n0;
            }
            uint8_t *p = (uint8_t *)ctx->d_stage;
            const size_t cap = ctx->stage_capacity;
            ctx->wb.qt.a = (float4 *)p; p += cap * 16; ctx->wb.qt.b = (float4 *)p; p += cap * 16; ctx->wb.qt.c = (float4 *)p; p += cap * 16;
            ctx->wb.qt.rng = (uint4 *)p; p += cap * 16; ctx->wb.sort_kr = (uint2 *)p; p += cap * 8; ctx->wb.qt.pixel = (uint32_t *)p;
            // origin cell bits behind the flag: 11, 14 for more than twelve million rays
            ctx->wb.sort_ob = n0 > (12u << 20) ? 5u : 4u;            // (C5, 8.3 M rays: 917 Mpaths/s with 4, 909 with 5; C2 in batches of eight, 16.6 M: 2.12 against 2.10 ms per frame)
            if (const char *e = getenv("RTGL_AMD_SORT_OB")) { const int v = atoi(e); if (v >= 1 && v <= 5) ctx->wb.sort_ob = (uint32_t)v; }      // (tuning)
            ctx->wb.sort_db = 4u;
            if (const char *e = getenv("RTGL_AMD_SORT_DB")) { const int v = atoi(e); if (v >= 2 && v <= 6) ctx->wb.sort_db = (uint32_t)v; }      // (tuning: the bins stay as many)
            ctx->wb.sort_bits = 8u + 3u * ctx->wb.sort_ob;
            ctx->wb.sort_T = ctx->wb.sort_bits - 1u - 2u * ctx->wb.sort_db;
            if (ctx->sort_bits_alloc < ctx->wb.sort_bits) {
                if (ctx->d_sort_hist) { HIPCHK(ctx, hipFree(ctx->d_sort_hist)); ctx->d_sort_hist = nullptr; }
                const size_t bins = (size_t)1 << ctx->wb.sort_bits;
                HIPCHK(ctx, hipMalloc((void **)&ctx->d_sort_hist, 2 * (bins + bins / kSortSeg) * sizeof(uint32_t)));
                ctx->sort_bits_alloc = ctx->wb.sort_bits; ctx->sort_sets_clean = false;
            }
            ctx->wb.sort_hist = ctx->d_sort_hist; ctx->wb.sort_hist_other = ctx->d_sort_hist;      // (set per binned bounce: launch_wavefront)
            set_bin_cells(ctx);
        }
        ctx->wb.hybrid_div = 3u;          // (measured on C2 / C5: every 7th claimed 707 / 752 Mpaths/s, every 3rd 720 / 776, every 2nd 724 / 777, all of them 677 / 737)
        if (const char *e = getenv("RTGL_AMD_HYBRID_DIV")) { const int v = atoi(e); if (v >= 1 && v <= 64) ctx->wb.hybrid_div = (uint32_t)v; }      // (tuning)
        ctx->wb.items = reinterpret_cast<uint32_t *>(ctx->d_items); ctx->wb.item_counts = reinterpret_cast<uint32_t *>(ctx->d_items);     // (allocated by the first culled launch)
        ctx->wb.cand = ctx->d_cand;
        ctx->wb.cand_counts = reinterpret_cast<uint32_t *>(ctx->d_cand + (size_t)ctx->cand_regions * ctx->cand_region_pairs);
        ctx->wb.cand_region = ctx->cand_region_pairs;
        // diagnostics: RTGL_DEBUG_CAND_CAP=n pretends a wave's region holds n pairs only, so that the in-place fallback of the scan runs
        if (const char *cc = getenv("RTGL_DEBUG_CAND_CAP")) { ctx->wb.cand_region = std::min<uint32_t>(ctx->wb.cand_region, (uint32_t)atoi(cc)); ctx->cand_fixed = true; }
    }
    if (ctx->wave_capacity < n0 || (multi_sample && !ctx->wave_multi)) {
        if (ctx->d_wave) { HIPCHK(ctx, hipFree(ctx->d_wave)); ctx->d_wave = nullptr; }
        // per queue: 4 x 16 B + 4 B per ray; per-pixel state for u_samples > 1: 4 x 16 B
        size_t q_bytes = (size_t)n0 * (68 + 8), bytes = 2 * q_bytes + 1024 + (multi_sample ? local_px * 64 : 0);
        HIPCHK(ctx, hipMalloc(&ctx->d_wave, bytes));
        ctx->wave_capacity = n0; ctx->wave_multi = multi_sample;
        uint8_t *p = (uint8_t *)ctx->d_wave;
        for (int q = 0; q < 2; ++q) {
            ctx->wb.q[q].a = (float4 *)p; p += (size_t)n0 * 16;
            ctx->wb.q[q].b = (float4 *)p; p += (size_t)n0 * 16;
            ctx->wb.q[q].c = (float4 *)p; p += (size_t)n0 * 16;
            ctx->wb.q[q].rng = (uint4 *)p; p += (size_t)n0 * 16;
        }
        for (int q = 0; q < 2; ++q) { ctx->wb.best[q] = (unsigned long long *)p; p += (size_t)n0 * 8; }
        for (int q = 0; q < 2; ++q) { ctx->wb.q[q].pixel = (uint32_t *)p; p += (size_t)n0 * 4; }
        p = (uint8_t *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
        if (multi_sample) {
            ctx->wb.sums = (float4 *)p; p += local_px * 16;
            ctx->wb.cam_a = (float4 *)p; p += local_px * 16;
            ctx->wb.cam_b = (float4 *)p; p += local_px * 16;
            ctx->wb.pix_rng = (uint4 *)p; p += local_px * 16;
        } else ctx->wb.sums = ctx->wb.cam_a = ctx->wb.cam_b = nullptr, ctx->wb.pix_rng = nullptr;
    }
    ctx->wb.counts = ctx->d_counts;
    ctx->wb.cand_peak = ctx->d_counts + ctx->counts_capacity;
    ctx->wb.group_bounds = ctx->d_group_bounds;
    return RTGL_OK;
}

template <int R, int MODE>
static void launch_bounce(rtgl_context *ctx, const SceneView &sc, const FrameParams &P, const ImageView &im, uint32_t n0, uint32_t bounce, uint4 *rng_out)
{
    dim3 grid((n0 + 256u * R - 1) / (256u * R));
    if (ctx->opt_counters)
        hipLaunchKernelGGL((bounce_kernel<R, MODE, true>), grid, dim3(256), 0, ctx->stream, sc, P, im, ctx->wb, bounce, rng_out, ctx->d_counters);
    else
        hipLaunchKernelGGL((bounce_kernel<R, MODE, false>), grid, dim3(256), 0, ctx->stream, sc, P, im, ctx->wb, bounce, rng_out, ctx->d_counters);
}

// Upper estimate of the rays entering `bounce`, for grid sizing only (kernels grid-stride, so a low
// estimate costs time, never correctness): last finished frame's count + 10 % + 2048, capped by n0.
static uint32_t estimate_rays(const rtgl_context *ctx, uint32_t n0, uint32_t bounce)
{
    if (bounce == 0 || bounce >= ctx->est_counts.size()) return n0;
    uint64_t e = (uint64_t)ctx->est_counts[bounce] + ctx->est_counts[bounce] / 10 + 2048;
    return (uint32_t)std::min<uint64_t>(e, n0);
}

template <int R, int MODE>
static void launch_intersect(rtgl_context *ctx, const SceneView &sc, uint32_t n0, uint32_t bounce)
{
    const uint32_t chunk = (uint32_t)ctx->opt_wf_chunk;
    const uint32_t est = estimate_rays(ctx, n0, bounce);
    dim3 grid((est + 256u * R - 1) / (256u * R), (sc.n_tri_visits + chunk - 1) / chunk);
    const bool early = bounce < (uint32_t)ctx->opt_wf_early;     // wave-level edge short circuit on the coherent bounces
#define RTGL_LAUNCH_ISECT(E, P, C) hipLaunchKernelGGL((intersect_kernel<R, MODE, E, P, C>), grid, dim3(256), 0, ctx->stream, sc, ctx->wb, bounce, chunk, ctx->d_counters)
    if (early) { if (ctx->opt_counters) RTGL_LAUNCH_ISECT(true, false, true); else RTGL_LAUNCH_ISECT(true, false, false); }
    else if (ctx->opt_wf_packed) { if (ctx->opt_counters) RTGL_LAUNCH_ISECT(false, true, true); else RTGL_LAUNCH_ISECT(false, true, false); }
    else { if (ctx->opt_counters) RTGL_LAUNCH_ISECT(false, false, true); else RTGL_LAUNCH_ISECT(false, false, false); }
#undef RTGL_LAUNCH_ISECT
}

// kernel 4: one block per CU (forced by the LDS request), persistent over the ray blocks of its triangle chunk
static uint32_t solo_chunks(const rtgl_context *ctx)
{
    const uint32_t n_quads = ctx->n_mf_groups * ctx->mf_group_quads;
    const uint32_t real_quads = std::min(n_quads, (ctx->n_tri_visits + (uint32_t)kMfQuadTris - 1) / (uint32_t)kMfQuadTris);
    const uint32_t chunk_quads = std::min((uint32_t)ctx->opt_mf_chunk_quads, std::max(real_quads, 1u));
    return (real_quads + chunk_quads - 1) / chunk_quads;
}

// work distribution of the scan (rt_scan.hpp): "scan_dynamic" 0 = by the mesh (hybrid; dynamic from 1,024 quads = 41k triangles on: few
// blocks per chunk), 1 = static turns, 2 = dynamic claims, 3 = planned (equal-cost intervals of the item line, no atomics), 4 = hybrid
// (turns + a claimed tail).  Returns the kernel's kDist: 0 static, 1 dynamic, 2 planned, 3 hybrid.
static int solo_dynamic(const rtgl_context *ctx)
{
    const uint32_t real_quads = std::min(ctx->n_mf_groups * ctx->mf_group_quads, (ctx->n_tri_visits + (uint32_t)kMfQuadTris - 1) / (uint32_t)kMfQuadTris);
    return ctx->opt_scan_dynamic ? ctx->opt_scan_dynamic - 1 : (real_quads >= 1024u ? 1 : 3);
}

// may the camera-ray keep bits computed for frame `a` serve frame `b`?  Same camera; then the rays differ by the jitter of camera_ray (:187-195)
// only: origins by at most 2 |aperture|, unit directions by at most 2 |aperture| / (|focal| - |aperture|)
static bool same_camera(const FrameParams &a, const FrameParams &b)
{
    return a.use_dof == b.use_dof && a.cam_fov == b.cam_fov && a.cam_aperture == b.cam_aperture && a.cam_focal == b.cam_focal
           && memcmp(a.cam_pos, b.cam_pos, sizeof a.cam_pos) == 0 && memcmp(a.cam_forward, b.cam_forward, sizeof a.cam_forward) == 0
           && memcmp(a.cam_up, b.cam_up, sizeof a.cam_up) == 0 && memcmp(a.cam_right, b.cam_right, sizeof a.cam_right) == 0;
}
static bool camera_keep_widening(const FrameParams &P, float *ro_add, float *sigma_add)
{
    *ro_add = *sigma_add = 0.0f;
    if (!P.use_dof) return true;                         // the same ray every frame, bit for bit
    const float a = fabsf(P.cam_aperture), f = fabsf(P.cam_focal);
    const float pn = sqrtf(P.cam_pos[0] * P.cam_pos[0] + P.cam_pos[1] * P.cam_pos[1] + P.cam_pos[2] * P.cam_pos[2]);
    if (!(a < 0.25f * f) || !(f < 1.0e18f) || !(pn < 1.0e18f)) return false;     // (NaN included) no bound worth having: certify every frame's rays
    *ro_add = 2.0f * a * 1.001f + 1.0e-5f * (1.0f + pn);
    *sigma_add = 2.0f * a / (f - a) * 1.001f + 4.0e-6f;
    return true;
}

static int launch_intersect_solo(rtgl_context *ctx, const SceneView &sc, uint32_t n0, uint32_t bounce, bool binned, const FrameParams *cam)
{
    const uint32_t gq = ctx->mf_group_quads, n_quads = ctx->n_mf_groups * gq;
    const uint32_t real_quads = std::min(n_quads, (ctx->n_tri_visits + (uint32_t)kMfQuadTris - 1) / (uint32_t)kMfQuadTris);
    using Cfg = SoloCfg;
    // Two waves per SIMD run the steady stream 1.5x faster (47 against 70 cycles per product).  Until the item loop moved into scalar
    // registers small launches were better off with one wave per SIMD (half as many rays per block); measured since: two waves win or tie
    // everywhere (C2 3.394 against 3.434 ms, a rank of eight 0.686 against 0.713 ms).  "scan_waves" = 0 / 2: two, 1: one.
    const uint32_t est = estimate_rays(ctx, n0, bounce);
    const uint32_t W = ctx->opt_scan_waves == 1 ? 1u : 2u;
    const uint32_t waves = 4u * W;
    const uint32_t est_gran = (est + Cfg::kRaysPerWave - 1u) / Cfg::kRaysPerWave;
    // A launch has (granules x chunks) work items for its waves, claimed dynamically (rt_scan.hpp).  Late bounces (and every bounce of a
    // rank that owns an eighth of the image) have few granules: cut the triangle range finer, down to 4 quads per chunk, until there are
    // three items per wave (each item pays its ray and group set-up again, ~20 % at 8 quads, so only as far as needed -- thresholds of 1, 2,
    // 4, 8 items per wave measured: 2-4 are best for a rank of four or eight, none matters at N = 1; never more chunks than CUs)
    uint32_t chunk_quads = std::min((uint32_t)ctx->opt_mf_chunk_quads, std::max(real_quads, 1u));
    const uint64_t items_per_wave = getenv("RTGL_AMD_ITEMS_PER_WAVE") ? (uint64_t)std::max(1, atoi(getenv("RTGL_AMD_ITEMS_PER_WAVE"))) : 3ull;      // (tuning; C2: bounce 5's launch 98 -> 89 us with three, bounces 6 and 7 +2 us each; a rank of four or eight, C4, C5: the same with two and three)
    while (chunk_quads > 4u && (uint64_t)est_gran * ((real_quads + chunk_quads - 1) / chunk_quads) < items_per_wave * (uint32_t)ctx->n_cus * waves
           && (real_quads + chunk_quads / 2 - 1) / (chunk_quads / 2) <= (uint32_t)ctx->n_cus)
        chunk_quads /= 2u;
    const uint32_t chunks = (real_quads + chunk_quads - 1) / chunk_quads;
    // (auto: the camera-ray bounce of a small mesh keeps its fixed turns -- almost every item is empty there and a claimed tail only adds
    // round trips: 74 us against 168 on C2; the binned bounces take the hybrid form: 660 -> 589, 553 -> 499 us)
    const int cull = ctx->opt_cull == 2 || (ctx->opt_cull >= 1 && bounce == 0) || (ctx->opt_cull == 3 && binned);
    // (unculled launches have items of equal cost: fixed turns are balanced there and a claimed tail only adds round trips)
    const int dist = (ctx->opt_scan_dynamic == 0 && (bounce == 0 || !cull) && solo_dynamic(ctx) == 3) ? 0 : solo_dynamic(ctx), dynamic = dist == 1;
    // one block per CU (forced by the LDS request); fewer when there is not an item per wave.  Static: the same number of blocks on
    // every chunk.
    uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)est_gran * chunks + waves - 1) / waves, (uint64_t)ctx->n_cus));
    if (dist == 0 || dist == 3) blocks = std::max(1u, std::min((est_gran + waves - 1u) / waves, std::max(1u, (uint32_t)ctx->n_cus / chunks))) * chunks;
    const size_t lds = std::max<size_t>(((size_t)chunk_quads * kMfQuadTiles + 4) * 1024, 96 * 1024);   // + the four rows read two trips ahead behind the last tile; > half of the CU's LDS with the static queue: one block per CU
#ifdef RT_SOLO_STAMPS
    if (!ctx->d_dbg_log) { HIPCHK(ctx, hipMalloc((void **)&ctx->d_dbg_log, (size_t)(2 + (2u << 22)) * 4)); HIPCHK(ctx, hipMemsetAsync(ctx->d_dbg_log, 0, 2048 * 8 + 16 * 64 * 2 * 16 * 8, ctx->stream)); }
#endif
    MfView mf{ctx->d_mf_groups, ctx->n_mf_groups, gq, n_quads, ctx->d_mf_A, ctx->d_dbg_log, ctx->d_mf_cull, ctx->d_edges_s, ctx->d_planes_s, ctx->d_mf_order};
    if (!ctx->solo_attr_set) {
        // allow the whole LDS of a CU (160 KB) minus the kernel's static share as dynamic shared memory.  The attribute belongs to the
        // (function, device) pair, so it is raised once per context -- a context is bound to one device -- not once per process.
        for (const void *fn : {reinterpret_cast<const void *>(&scan_solo_kernel<false, 1, 0>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 1, 0>),
                               reinterpret_cast<const void *>(&scan_solo_kernel<false, 2, 0>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 2, 0>),
                               reinterpret_cast<const void *>(&scan_solo_kernel<false, 1, 1>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 1, 1>),
                               reinterpret_cast<const void *>(&scan_solo_kernel<false, 2, 1>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 2, 1>),
                               reinterpret_cast<const void *>(&scan_solo_kernel<false, 1, 2>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 1, 2>),
                               reinterpret_cast<const void *>(&scan_solo_kernel<false, 2, 2>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 2, 2>),
                               reinterpret_cast<const void *>(&scan_solo_kernel<false, 1, 3>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 1, 3>),
                               reinterpret_cast<const void *>(&scan_solo_kernel<false, 2, 3>), reinterpret_cast<const void *>(&scan_solo_kernel<true, 2, 3>)}) {
            hipFuncAttributes fattr;
            HIPCHK(ctx, hipFuncGetAttributes(&fattr, fn));
            HIPCHK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - fattr.sharedSizeBytes)));
        }
        ctx->solo_attr_set = true;
    }
    // packet culling pays where the 128 rays of a granule are coherent: the camera rays, and every queue that was binned (option "cull":
    // 0 never, 1 bounce 0, 2 every bounce as the queues come, 3 (default) bounce 0 and the binned bounces)

    if (cull) {
        // camera-ray bounce of a single frame: the bits of an earlier frame of the same camera, image and scene, if there are any
        float ro_add = 0.0f, sigma_add = 0.0f;
        bool have_bits = false;
        if (bounce == 0 && cam && camera_keep_widening(*cam, &ro_add, &sigma_add) && !getenv("RTGL_AMD_NO_CAMERA_KEEP")) {
            const size_t need = ((size_t)n0 / 128 + 16) * ctx->wb.keep_words;
            if (ctx->keep0_capacity < need) {
                if (ctx->d_keep0) { HIPCHK(ctx, hipFree(ctx->d_keep0)); ctx->d_keep0 = nullptr; }
                HIPCHK(ctx, hipMalloc((void **)&ctx->d_keep0, need * sizeof(uint32_t)));
                ctx->keep0_capacity = need; ctx->keep0_valid = false;
            }
            have_bits = ctx->keep0_valid && ctx->keep0_n0 == n0 && ctx->keep0_words == ctx->wb.keep_words && ctx->keep0_scene == ctx->scene_version && same_camera(ctx->keep0_params, *cam);
            ctx->wb.keep = ctx->d_keep0;
            if (!have_bits) { ctx->keep0_valid = true; ctx->keep0_n0 = n0; ctx->keep0_words = ctx->wb.keep_words; ctx->keep0_scene = ctx->scene_version; ctx->keep0_params = *cam; }
        } else { ctx->wb.keep = ctx->d_keep; ro_add = sigma_add = 0.0f; }
        if (!have_bits)
        hipLaunchKernelGGL(packet_cull_kernel, dim3(std::max(1u, std::min((est_gran + 3u) / 4u, 8192u))), dim3(256), 0, ctx->stream, ctx->wb, ctx->d_mf_cull, real_quads * (uint32_t)kMfQuadTiles, bounce, ro_add, sigma_add);
        if (dist == 2) {
            // planned: cost prefix sums per chunk [chunks x stride u32][chunks totals u32][chunks + 1 starts u64]
            const uint32_t stride = n0 / Cfg::kRaysPerWave + 1u;
            const size_t off_tot = (((size_t)chunks * stride * sizeof(uint32_t)) + 255) & ~(size_t)255, off_base = (off_tot + (size_t)chunks * sizeof(uint32_t) + 255) & ~(size_t)255;
            const size_t need = off_base + ((size_t)chunks + 1) * sizeof(unsigned long long);
            if (ctx->plan_capacity < need) {
                if (ctx->d_plan) { HIPCHK(ctx, hipFree(ctx->d_plan)); ctx->d_plan = nullptr; }
                HIPCHK(ctx, hipMalloc(&ctx->d_plan, need));
                ctx->plan_capacity = need;
            }
            ctx->wb.plan_prefix = reinterpret_cast<uint32_t *>(ctx->d_plan); ctx->wb.plan_stride = stride;
            ctx->wb.plan_total = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ctx->d_plan) + off_tot);
            ctx->wb.plan_base = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(ctx->d_plan) + off_base);
            if (ctx->opt_counters) hipLaunchKernelGGL(scan_plan_kernel<true>, dim3(chunks), dim3(256), 0, ctx->stream, ctx->wb, bounce, chunk_quads, real_quads, ctx->d_counters);
            else hipLaunchKernelGGL(scan_plan_kernel<false>, dim3(chunks), dim3(256), 0, ctx->stream, ctx->wb, bounce, chunk_quads, real_quads, ctx->d_counters);
            hipLaunchKernelGGL(scan_plan_base_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->wb, chunks);
        }
        if (dynamic) {
        // work items of the culled launch: [one count per chunk][chunks x (granules of the whole image) entries]
        const uint32_t stride = n0 / Cfg::kRaysPerWave + 1u;
        const size_t head = ((size_t)ctx->wb.sched_stride * sizeof(uint32_t) + 255) & ~(size_t)255, need = head + (size_t)chunks * stride * sizeof(uint32_t);
        if (ctx->items_capacity < need) {
            if (ctx->d_items) { HIPCHK(ctx, hipFree(ctx->d_items)); ctx->d_items = nullptr; }
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_items, need));
            ctx->items_capacity = need;
        }
        ctx->wb.item_counts = reinterpret_cast<uint32_t *>(ctx->d_items);
        ctx->wb.items = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ctx->d_items) + head);
        ctx->wb.items_stride = stride;
        HIPCHK(ctx, hipMemsetAsync(ctx->wb.item_counts, 0, (size_t)chunks * sizeof(uint32_t), ctx->stream));
        const dim3 igrid(std::max(1u, std::min((est_gran + 255u) / 256u, 1024u)), chunks);
        if (ctx->opt_counters) hipLaunchKernelGGL(cull_items_kernel<true>, igrid, dim3(256), 0, ctx->stream, ctx->wb, bounce, chunk_quads, real_quads, ctx->d_counters);
        else hipLaunchKernelGGL(cull_items_kernel<false>, igrid, dim3(256), 0, ctx->stream, ctx->wb, bounce, chunk_quads, real_quads, ctx->d_counters);
        }
    }
    // (testing the survivors of small launches in place instead of launching the narrow phase was measured: never faster --
    // rank of eight 0.73 -> 0.76-0.85 ms)
#define RTGL_LAUNCH_SCAN(C, WW, D) hipLaunchKernelGGL((scan_solo_kernel<C, WW, D>), dim3(blocks), dim3(256 * WW), lds, ctx->stream, sc, ctx->wb, mf, bounce, chunk_quads, chunks, ctx->d_counters, ctx->opt_debug_skip_exact, cull)
#define RTGL_LAUNCH_SCAN_W(C, D) do { if (W == 2) RTGL_LAUNCH_SCAN(C, 2, D); else RTGL_LAUNCH_SCAN(C, 1, D); } while (0)
    if (dist == 1) { if (ctx->opt_counters) RTGL_LAUNCH_SCAN_W(true, 1); else RTGL_LAUNCH_SCAN_W(false, 1); }
    else if (dist == 2) { if (ctx->opt_counters) RTGL_LAUNCH_SCAN_W(true, 2); else RTGL_LAUNCH_SCAN_W(false, 2); }
    else if (dist == 3) { if (ctx->opt_counters) RTGL_LAUNCH_SCAN_W(true, 3); else RTGL_LAUNCH_SCAN_W(false, 3); }
    else { if (ctx->opt_counters) RTGL_LAUNCH_SCAN_W(true, 0); else RTGL_LAUNCH_SCAN_W(false, 0); }
#undef RTGL_LAUNCH_SCAN_W
#undef RTGL_LAUNCH_SCAN
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(narrow_phase_kernel, dim3(blocks * waves, kNarrowSplit), dim3(256), 0, ctx->stream, sc, ctx->wb, mf, bounce, blocks * waves);
    return RTGL_OK;
}

// frames.size() > 1: a batch -- every frame's camera rays are generated into its own stretch of queue 0 (n0_frame slots), everything
// behind that sees ONE frame of n0 = B x n0_frame rays, and resolve_batch_kernel applies the frames' results to the image in order
static int launch_wavefront(rtgl_context *ctx, const SceneView &sc, const std::vector<FrameParams> &frames, const ImageView &im, uint32_t n0_frame, uint4 *rng_out)
{
    const FrameParams &P = frames[0];
    const uint32_t B = (uint32_t)frames.size(), n0 = n0_frame * B;
    const dim3 gen_grid((n0_frame + 255) / 256);
    // pick up the ray counts of the most recent finished frame (never blocks)
    if (ctx->counts_pending && hipEventQuery(ctx->counts_ev) == hipSuccess) {
        ctx->counts_pending = false;
        if (ctx->counts_n0 == n0) ctx->est_counts.assign(ctx->h_counts, ctx->h_counts + ctx->counts_len);
        else ctx->est_counts.clear();
        // the fullest candidate region of that frame: grow before the NEXT frame is enqueued (ensure_wave_buffers), never shrink
        const uint32_t peak = ctx->h_counts[ctx->counts_capacity];
        if (!ctx->cand_fixed && peak > ctx->cand_region_pairs) ctx->cand_region_target = (uint32_t)std::min<uint64_t>((uint64_t)peak + peak / 4, 0xFFFFFFF0u);
    }
    for (uint32_t s = 0; s < P.samples; ++s) {
        const uint32_t n_counts = ctx->counts_capacity + 1u <= 256u ? ctx->counts_capacity + 1u : 0u;      // cleared by generate_rays_kernel's first block
        if (!n_counts) HIPCHK(ctx, hipMemsetAsync(ctx->d_counts, 0, counts_bytes(ctx->counts_capacity), ctx->stream));
        // the scan launches' work counters (rt_scan.hpp): cleared by the first threads of generate_rays_kernel where there are enough of them
        uint32_t n_sched = 0u;
        if (ctx->opt_kernel == RTGL_KERNEL_WAVEFRONT_MFMA_SOLO && ctx->d_sched && (solo_dynamic(ctx) == 1 || solo_dynamic(ctx) == 3)) {
            const size_t words = (size_t)(P.max_bounce + 2) * ctx->wb.sched_stride;
            if (words <= (size_t)gen_grid.x * 256u && !getenv("RTGL_AMD_SCHED_FILL")) n_sched = (uint32_t)words;      // (the variable: measurement of the fill launch this replaces)
            else HIPCHK(ctx, hipMemsetAsync(ctx->d_sched, 0, words * sizeof(uint32_t), ctx->stream));
        }
        for (uint32_t f = 0; f < B; ++f)
            hipLaunchKernelGGL(generate_rays_kernel, gen_grid, dim3(256), 0, ctx->stream, frames[f], im, ctx->wb, s, n0_frame,
                               ctx->opt_counters ? ctx->d_counters : (Counters *)nullptr, f == 0 ? n_counts : 0u, f * n0_frame, B > 1 ? f << 28 : 0u, f == 0 ? n0 : 0u, f == 0 ? n_sched : 0u);
        bool binned = false;                                 // the queue of the bounce about to be launched was binned
        for (uint32_t b = 0; b < P.max_bounce; ++b) {
            const int key = ctx->opt_wf_mode * 10 + ctx->opt_wf_rays;
            if (ctx->opt_kernel == RTGL_KERNEL_WAVEFRONT_SPLIT || ctx->opt_kernel == RTGL_KERNEL_WAVEFRONT_MFMA_SOLO) {
                if (sc.n_tri_visits > 0 && ctx->opt_kernel == RTGL_KERNEL_WAVEFRONT_MFMA_SOLO) {
                    kev_mark(ctx);
                    { const int rc = launch_intersect_solo(ctx, sc, n0, b, binned, (B == 1 && P.samples == 1u) ? &P : nullptr); if (rc) return rc; }
                    kev_mark(ctx);
                } else if (sc.n_tri_visits > 0) {
                    kev_mark(ctx);
                    switch (key) {
                    case 1: launch_intersect<1, kScalar>(ctx, sc, n0, b); break;
                    case 2: launch_intersect<2, kScalar>(ctx, sc, n0, b); break;
                    case 4: launch_intersect<4, kScalar>(ctx, sc, n0, b); break;
                    case 8: launch_intersect<8, kScalar>(ctx, sc, n0, b); break;
                    case 11: launch_intersect<1, kLds>(ctx, sc, n0, b); break;
                    case 12: launch_intersect<2, kLds>(ctx, sc, n0, b); break;
                    case 14: launch_intersect<4, kLds>(ctx, sc, n0, b); break;
                    case 18: launch_intersect<8, kLds>(ctx, sc, n0, b); break;
                    default: return fail(ctx, RTGL_ERR_STATE, "unsupported wf_mode / wf_rays combination");
                    }
                    kev_mark(ctx);
                }
                const dim3 shade_grid((estimate_rays(ctx, n0, b) + 255) / 256);
                // ray binning: the queue of the next bounce in (direction bin, origin cell) order, where it is long enough to pay for
                // the three small launches and the extra pass over its rays
                const uint32_t est_next = estimate_rays(ctx, n0, b + 1u);
                const bool bin_next = ctx->opt_kernel == RTGL_KERNEL_WAVEFRONT_MFMA_SOLO && ctx->opt_cull == 3 && sc.n_tri_visits > 0 && b + 1u < P.max_bounce
                                      && est_next >= (uint32_t)ctx->opt_sort_min_rays;
                if (bin_next) {
                    // two sets of bin counters take turns: sort_scatter_kernel zeroes the one the next binned bounce will count in
                    const size_t bins = (size_t)1 << ctx->wb.sort_bits, set_words = bins + bins / kSortSeg;
                    if (!ctx->sort_sets_clean || ctx->sort_set_bits != ctx->wb.sort_bits) {
                        HIPCHK(ctx, hipMemsetAsync(ctx->d_sort_hist, 0, 2 * set_words * sizeof(uint32_t), ctx->stream));
                        ctx->sort_set = 0; ctx->sort_set_bits = ctx->wb.sort_bits;
                    }
                    ctx->sort_sets_clean = false;              // (true again once this bounce's launches are in the stream)
                    if (getenv("RTGL_AMD_HIST_FILL")) HIPCHK(ctx, hipMemsetAsync(ctx->d_sort_hist + (size_t)ctx->sort_set * set_words, 0, bins * sizeof(uint32_t), ctx->stream));      // (measurement: the fill launch per bounce that the turns replace)
                    ctx->wb.sort_hist = ctx->d_sort_hist + (size_t)ctx->sort_set * set_words;
                    ctx->wb.sort_hist_other = ctx->d_sort_hist + (size_t)(ctx->sort_set ^ 1) * set_words;
                    if (ctx->opt_counters) hipLaunchKernelGGL((shade_kernel<true, true>), shade_grid, dim3(256), 0, ctx->stream, sc, P, im, ctx->wb, b, rng_out, ctx->d_counters);
                    else hipLaunchKernelGGL((shade_kernel<false, true>), shade_grid, dim3(256), 0, ctx->stream, sc, P, im, ctx->wb, b, rng_out, ctx->d_counters);
                    hipLaunchKernelGGL(sort_sums_kernel, dim3((unsigned)(bins / kSortSeg)), dim3(256), 0, ctx->stream, ctx->wb);
                    hipLaunchKernelGGL(sort_prefix_kernel, dim3((unsigned)(bins / kSortSeg)), dim3(256), 0, ctx->stream, ctx->wb);
                    hipLaunchKernelGGL(sort_scatter_kernel, dim3(std::max(1u, std::min((est_next + 255u) / 256u, 16384u))), dim3(256), 0, ctx->stream, ctx->wb, b + 1u);
                    HIPCHK(ctx, hipGetLastError());
                    ctx->sort_set ^= 1; ctx->sort_sets_clean = true;
                } else if (ctx->opt_counters)
                    hipLaunchKernelGGL((shade_kernel<true, false>), shade_grid, dim3(256), 0, ctx->stream, sc, P, im, ctx->wb, b, rng_out, ctx->d_counters);
                else
                    hipLaunchKernelGGL((shade_kernel<false, false>), shade_grid, dim3(256), 0, ctx->stream, sc, P, im, ctx->wb, b, rng_out, ctx->d_counters);
                binned = bin_next;
                continue;
            }
            kev_mark(ctx);
            switch (key) {
            case 1: launch_bounce<1, kScalar>(ctx, sc, P, im, n0, b, rng_out); break;
            case 2: launch_bounce<2, kScalar>(ctx, sc, P, im, n0, b, rng_out); break;
            case 4: launch_bounce<4, kScalar>(ctx, sc, P, im, n0, b, rng_out); break;
            case 11: launch_bounce<1, kLds>(ctx, sc, P, im, n0, b, rng_out); break;
            case 12: launch_bounce<2, kLds>(ctx, sc, P, im, n0, b, rng_out); break;
            case 14: launch_bounce<4, kLds>(ctx, sc, P, im, n0, b, rng_out); break;
            default: return fail(ctx, RTGL_ERR_STATE, "unsupported wf_mode / wf_rays combination");
            }
            kev_mark(ctx);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    if (P.samples > 1u) {
        hipLaunchKernelGGL(resolve_kernel, gen_grid, dim3(256), 0, ctx->stream, P, im, ctx->wb);
        HIPCHK(ctx, hipGetLastError());
    }
    if (B > 1) {
        BatchInfo bi{};
        bi.n = B;
        for (uint32_t f = 0; f < B; ++f) { bi.frames[f] = frames[f].frames; bi.reset[f] = frames[f].reset_flag; }
        hipLaunchKernelGGL(resolve_batch_kernel, gen_grid, dim3(256), 0, ctx->stream, im, ctx->wb, bi);
        HIPCHK(ctx, hipGetLastError());
    }
    if (!ctx->counts_pending) {      // feed the next frames' grid sizes; skipped while an earlier copy is in flight
        ctx->counts_len = P.max_bounce + 1; ctx->counts_n0 = n0;
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts, counts_bytes(ctx->counts_capacity), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->counts_ev, ctx->stream));
        ctx->counts_pending = true;
    }
    return RTGL_OK;
}

extern "C" int rtgl_set_frame_params(rtgl_context *ctx, const rtgl_frame_params *p)
{
    ENTER_NOFLUSH(ctx);
    FANOUT(ctx, rtgl_set_frame_params(part, p));
    if (!p) return fail(ctx, RTGL_ERR_INVALID, "params is NULL");
    static_assert(sizeof(FrameParams) == sizeof(rtgl_frame_params), "FrameParams mirrors rtgl_frame_params");
    memcpy(&ctx->params, p, sizeof(FrameParams));
    ctx->have_params = true;
    return RTGL_OK;
}

static int render_batch(rtgl_context *ctx, const std::vector<FrameParams> &batch);

// what the frames of a batch must share: everything the kernels behind ray generation read from the uniforms
static bool batch_compatible(const FrameParams &a, const FrameParams &b)
{
    return a.samples == b.samples && a.max_bounce == b.max_bounce && a.use_envmap == b.use_envmap && memcmp(a.background, b.background, sizeof a.background) == 0;
}

static int flush_pending_noexcept(rtgl_context *ctx) { return flush_pending(ctx); }
static int flush_pending(rtgl_context *ctx)
{
    if (ctx->pending.empty()) return RTGL_OK;
    std::vector<FrameParams> batch;
    batch.swap(ctx->pending);
    return render_batch(ctx, batch);
}

extern "C" int rtgl_render_frame(rtgl_context *ctx)
{
    ENTER_NOFLUSH(ctx);
    if (!ctx->workers.empty()) {                         // every part's frame is submitted by its own thread; this one waits for all of them
        ctx->gathered = false;
        for (auto &w : ctx->workers) { { std::lock_guard<std::mutex> lk(w->m); w->state = PartWorker::kJob; } w->cv.notify_all(); }
        int first = RTGL_OK; const rtgl_context *bad = nullptr;
        for (auto &w : ctx->workers) {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&w] { return w->state == PartWorker::kDone; });
            w->state = PartWorker::kIdle;
            if (w->rc && !first) { first = w->rc; bad = w->part; }
        }
        return first ? fail(ctx, first, std::string("device ") + std::to_string(bad->device) + ": " + bad->error) : RTGL_OK;
    }
    FANOUT(ctx, rtgl_render_frame(part));
    if (!ctx->have_params) return fail(ctx, RTGL_ERR_STATE, "rtgl_set_frame_params has not been called");
    if (ctx->params.samples == 0) return fail(ctx, RTGL_ERR_INVALID, "u_samples == 0 divides by zero in the reference; refused");
    // frame batching: hold the frame back until the batch is full.  Only what the batched pipeline covers: one sample per frame, the
    // per-bounce pipeline (a scene with triangles), no per-frame read-outs (counters, RNG states)
    const bool batchable = ctx->opt_frame_batch > 1 && ctx->params.samples == 1 && ctx->params.max_bounce > 0 && !ctx->opt_counters && !ctx->opt_rng_state
                           && !ctx->opt_kernel_timing && (ctx->n_tri_visits > 0 || ctx->tris_dirty || ctx->kernel_explicit) && ctx->opt_kernel != RTGL_KERNEL_MEGA;
    if (!ctx->pending.empty() && (!batchable || !batch_compatible(ctx->pending.front(), ctx->params))) { const int rc = flush_pending(ctx); if (rc) return rc; }
    if (batchable) {
        ctx->pending.push_back(ctx->params);
        return (int)ctx->pending.size() >= ctx->opt_frame_batch ? flush_pending(ctx) : RTGL_OK;
    }
    return render_batch(ctx, std::vector<FrameParams>(1, ctx->params));
}

// one frame, or the frames of a batch in one set of launches
static int render_batch(rtgl_context *ctx, const std::vector<FrameParams> &batch)
{
    const uint32_t B = (uint32_t)batch.size();
    if (ctx->visits_dirty) { int rc = rebuild_sphere_visits(ctx); if (rc) return rc; }
    if (ctx->tris_dirty) { int rc = rebuild_triangles(ctx); if (rc) return rc; }
    if (ctx->opt_rng_state && !ctx->d_rng)
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_rng, (size_t)std::max(ctx->local_rows, 1) * ctx->width * sizeof(uint4)));

    SceneView sc{};
    sc.spheres = ctx->d_spheres; sc.n_spheres = ctx->n_spheres;
    sc.sphere_visits = ctx->d_sphere_visits; sc.n_sphere_visits = ctx->n_sphere_visits;
    sc.materials = ctx->d_materials; sc.n_materials = ctx->n_materials;
    sc.tri_edges = ctx->d_edges; sc.tri_planes = ctx->d_planes; sc.n_tri_visits = ctx->n_tri_visits;
    sc.env = ctx->d_env; sc.env_w = ctx->env_w; sc.env_h = ctx->env_h; sc.env_c = ctx->env_c; sc.env_faces = ctx->env_faces;
    std::vector<FrameParams> frames(batch);
    for (FrameParams &f : frames) if (!ctx->d_env) f.use_envmap = 0;   // src/renderer.cpp:104-110: no cube map => u_use_envmap = false
    const FrameParams &P = frames[0];
    ImageView im{};
    im.pixels = ctx->d_image; im.width = ctx->width; im.height = ctx->height;
    im.disp_w = ctx->width / 8 * 8; im.disp_h = ctx->height / 8 * 8;
    im.local_rows = ctx->local_rows; im.rank = ctx->rank; im.world = ctx->world; im.strip_rows = ctx->strip_rows;

    if (ctx->opt_counters) HIPCHK(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(Counters), ctx->stream));
    // rows of the dispatch footprint held locally: a prefix of the local rows (strips are 8-row aligned)
    int local_disp_rows = 0;
    for (int lr = 0; lr < ctx->local_rows; ++lr) if (rtgl_local_row_to_global(ctx, lr) < im.disp_h) local_disp_rows = lr + 1;
    const uint32_t n0_frame = (uint32_t)im.disp_w * (uint32_t)local_disp_rows;
    if ((uint64_t)n0_frame * B > 0xFFFFFFF0ull || (size_t)std::max(ctx->local_rows, 1) * ctx->width > (size_t)kBatchPixelMask) return fail(ctx, RTGL_ERR_INVALID, "frame_batch: the batch does not fit 32-bit ray slots");
    const uint32_t n0 = n0_frame * B;            // rays entering bounce 0: all frames of the batch
    uint4 *rng_out = ctx->opt_rng_state ? ctx->d_rng : nullptr;
    // a scene without triangles has no scan to split off: one megakernel launch per frame beats the per-bounce pipeline
    // (C1, 256x256 spheres: 1460 vs 1025 Mpaths/s) unless the caller asked for a specific variant
    const int kernel = (ctx->n_tri_visits == 0 && !ctx->kernel_explicit) ? (int)RTGL_KERNEL_MEGA : ctx->opt_kernel;
    const bool use_wavefront = kernel != RTGL_KERNEL_MEGA && P.max_bounce > 0;
    ctx->kernel_in_use = use_wavefront ? kernel : (int)RTGL_KERNEL_MEGA;
    if (B > 1 && !(use_wavefront && n0 > 0)) {          // (nothing to batch: a scene that lost its triangles meanwhile, an empty tile) one frame at a time
        for (const FrameParams &f : batch) { const int rc = render_batch(ctx, std::vector<FrameParams>(1, f)); if (rc) return rc; }
        return RTGL_OK;
    }
    if (use_wavefront && n0 > 0) { int rc = ensure_wave_buffers(ctx, n0, P.max_bounce, P.samples > 1); if (rc) return rc; }
    ctx->wb.batch_rad = nullptr; ctx->wb.batch_px = 0;
    if (B > 1) {
        const size_t local_px = (size_t)std::max(ctx->local_rows, 1) * ctx->width, need = local_px * B;
        if (ctx->batch_capacity < need) {
            if (ctx->d_batch_rad) { HIPCHK(ctx, hipFree(ctx->d_batch_rad)); ctx->d_batch_rad = nullptr; }
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_batch_rad, need * sizeof(float4)));
            ctx->batch_capacity = need;
        }
        ctx->wb.batch_rad = ctx->d_batch_rad; ctx->wb.batch_px = (uint32_t)local_px;
    }
    ctx->last_batch_frames = B;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    // option kernel_timing = N: every N-th frame since the last reset carries the event pairs (each pair costs ~3 us of gap)
    ctx->timing_this_frame = ctx->opt_kernel_timing > 0 && (ctx->timing_frame_counter++ % (uint32_t)ctx->opt_kernel_timing) == 0u;
    if (ctx->timing_this_frame) {
        if (ctx->kev_frame_start.size() >= 4096) { ctx->kev_used = 0; ctx->kev_frame_start.clear(); }   // bounded history
        ctx->kev_frame_start.push_back(ctx->kev_used);
        kev_mark(ctx);                                   // frame begin
    }
    if (n0 > 0 && !use_wavefront) {
        dim3 grid((im.disp_w + 31) / 32, (local_disp_rows + 7) / 8);
        kev_mark(ctx);
        if (ctx->opt_counters)
            hipLaunchKernelGGL(pathtrace_mega_kernel<true>, grid, dim3(256), 0, ctx->stream, sc, P, im, rng_out, ctx->d_counters);
        else
            hipLaunchKernelGGL(pathtrace_mega_kernel<false>, grid, dim3(256), 0, ctx->stream, sc, P, im, rng_out, ctx->d_counters);
        kev_mark(ctx);
        HIPCHK(ctx, hipGetLastError());
    } else if (n0 > 0) {
        int rc = launch_wavefront(ctx, sc, frames, im, n0_frame, rng_out);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    kev_mark(ctx);                                       // frame end
    ctx->timed = true;
    return RTGL_OK;
}

extern "C" int rtgl_synchronize(rtgl_context *ctx)
{
    ENTER(ctx);
    for (rtgl_context *part : ctx->parts) { const int rc = rtgl_synchronize(part); if (rc) return fail(ctx, rc, part->error); }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RTGL_OK;
}

extern "C" int rtgl_last_frame_ms(rtgl_context *ctx, float *ms)
{
    ENTER(ctx);
    if (!ctx->parts.empty()) {
        if (!ms) return fail(ctx, RTGL_ERR_INVALID, "ms is NULL");
        *ms = 0.0f;
        for (rtgl_context *part : ctx->parts) { float m = 0.0f; const int rc = rtgl_last_frame_ms(part, &m); if (rc) return fail(ctx, rc, part->error); *ms = std::max(*ms, m); }
        return RTGL_OK;
    }
    if (!ms || !ctx->timed) return fail(ctx, RTGL_ERR_STATE, "no frame has been rendered");
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    *ms /= (float)std::max(ctx->last_batch_frames, 1u);          // a batch of frames was timed as one
    return RTGL_OK;
}

// sums the event pairs of frames [first, last) of the recorded history
static int sum_timing(rtgl_context *ctx, size_t first, size_t last, rtgl_frame_timing *out)
{
    memset(out, 0, sizeof *out);
    for (size_t f = first; f < last; ++f) {
        const uint32_t b = ctx->kev_frame_start[f], e = (f + 1 < ctx->kev_frame_start.size()) ? ctx->kev_frame_start[f + 1] : ctx->kev_used;
        if (e < b + 2) continue;
        float ms = 0.0f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->kev[b], ctx->kev[e - 1]));
        out->frame_ms += ms;
        for (uint32_t i = b + 1; i + 1 < e - 1; i += 2) {
            HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->kev[i], ctx->kev[i + 1]));
            out->intersect_ms += ms; out->intersect_launches++;
        }
    }
    return RTGL_OK;
}

extern "C" int rtgl_last_frame_timing(rtgl_context *ctx, rtgl_frame_timing *out)
{
    ENTER(ctx);
    if (!ctx->parts.empty()) { const int rc = rtgl_last_frame_timing(ctx->parts[0], out); return rc ? fail(ctx, rc, ctx->parts[0]->error) : RTGL_OK; }   // device 0's share
    if (!out || !ctx->timed) return fail(ctx, RTGL_ERR_STATE, "no frame has been rendered");
    if (!ctx->opt_kernel_timing || ctx->kev_frame_start.empty()) return fail(ctx, RTGL_ERR_STATE, "option kernel_timing was not enabled before rendering");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return sum_timing(ctx, ctx->kev_frame_start.size() - 1, ctx->kev_frame_start.size(), out);
}

extern "C" int rtgl_accumulated_timing(rtgl_context *ctx, rtgl_frame_timing *out, uint32_t *frames_out)
{
    ENTER(ctx);
    if (!ctx->parts.empty()) { const int rc = rtgl_accumulated_timing(ctx->parts[0], out, frames_out); return rc ? fail(ctx, rc, ctx->parts[0]->error) : RTGL_OK; }
    if (!out) return fail(ctx, RTGL_ERR_INVALID, "out is NULL");
    if (!ctx->opt_kernel_timing) return fail(ctx, RTGL_ERR_STATE, "option kernel_timing is not enabled");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (frames_out) *frames_out = (uint32_t)ctx->kev_frame_start.size();
    return sum_timing(ctx, 0, ctx->kev_frame_start.size(), out);
}

extern "C" int rtgl_timing_reset(rtgl_context *ctx)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_timing_reset(part));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->kev_used = 0; ctx->kev_frame_start.clear(); ctx->timing_frame_counter = 0;
    return RTGL_OK;
}

extern "C" int rtgl_read_image_f32(rtgl_context *ctx, float *rgba)
{
    ENTER(ctx);
    if (!rgba) return fail(ctx, RTGL_ERR_INVALID, "rgba is NULL");
    if (!ctx->parts.empty()) { const int rc = multi_gather(ctx); if (rc) return rc; }
    HIPCHK(ctx, hipMemcpyAsync(rgba, ctx->d_image, (size_t)ctx->local_rows * ctx->width * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RTGL_OK;
}

extern "C" int rtgl_write_image_f32(rtgl_context *ctx, const float *rgba)
{
    ENTER(ctx);
    if (!rgba) return fail(ctx, RTGL_ERR_INVALID, "rgba is NULL");
    if (!ctx->parts.empty()) {                          // scatter the rows to their owners
        ctx->gathered = false;
        std::vector<float> local;
        for (rtgl_context *part : ctx->parts) {
            local.resize((size_t)std::max(part->local_rows, 1) * ctx->width * 4);
            for (int lr = 0; lr < part->local_rows; ++lr)
                memcpy(local.data() + (size_t)lr * ctx->width * 4, rgba + (size_t)rtgl_local_row_to_global(part, lr) * ctx->width * 4, (size_t)ctx->width * 16);
            const int rc = rtgl_write_image_f32(part, local.data());
            if (rc) return fail(ctx, rc, part->error);
        }
        return RTGL_OK;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_image, rgba, (size_t)ctx->local_rows * ctx->width * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RTGL_OK;
}

extern "C" int rtgl_clear_image(rtgl_context *ctx)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_clear_image(part));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_image, 0, (size_t)ctx->local_rows * ctx->width * 16, ctx->stream));
    return RTGL_OK;
}

extern "C" int rtgl_read_image_u8(rtgl_context *ctx, uint8_t *rgba, int flip)
{
    ENTER(ctx);
    if (!rgba) return fail(ctx, RTGL_ERR_INVALID, "rgba is NULL");
    if (!ctx->parts.empty()) { const int rc = multi_gather(ctx); if (rc) return rc; }
    size_t n = (size_t)ctx->local_rows * ctx->width;
    if (n == 0) return RTGL_OK;
    if (!ctx->d_u8) HIPCHK(ctx, hipMalloc((void **)&ctx->d_u8, n * 4));
    hipLaunchKernelGGL(image_to_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->d_image, ctx->d_u8, ctx->width, ctx->local_rows, flip);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(rgba, ctx->d_u8, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RTGL_OK;
}

extern "C" int rtgl_local_rows(const rtgl_context *ctx) { return ctx ? ctx->local_rows : RTGL_ERR_INVALID; }

extern "C" int rtgl_local_row_to_global(const rtgl_context *ctx, int lr)
{
    if (!ctx || lr < 0 || lr >= ctx->local_rows) return RTGL_ERR_INVALID;
    if (ctx->world == 1) return lr;
    int ls = lr / ctx->strip_rows, within = lr - ls * ctx->strip_rows;
    return (ls * ctx->world + ctx->rank) * ctx->strip_rows + within;
}

extern "C" void *rtgl_device_image(rtgl_context *ctx)
{
    if (!ctx) return nullptr;
    if (!ctx->pending.empty()) {                         // frames a batching context still holds back: a failed submission is an error, not a stale image
        if (hipSetDevice(ctx->device) != hipSuccess) { ctx->error = "rtgl_device_image: hipSetDevice failed"; return nullptr; }
        if (flush_pending(ctx) != RTGL_OK) return nullptr;      // (ctx->error holds the reason)
    }
    return (void *)ctx->d_image;
}

extern "C" int rtgl_bind_device_image(rtgl_context *ctx, void *dptr)
{
    ENTER(ctx);
    if (!ctx->parts.empty()) return fail(ctx, RTGL_ERR_STATE, "a multi-device context renders into its own per-device tile buffers");
    ctx->d_image = dptr ? (float4 *)dptr : ctx->d_image_own;
    return RTGL_OK;
}

extern "C" int rtgl_set_stream(rtgl_context *ctx, void *hip_stream)
{
    ENTER(ctx);
    if (!ctx->parts.empty()) return fail(ctx, RTGL_ERR_STATE, "a multi-device context owns one stream per device");
    const hipStream_t want = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (want != ctx->own_stream && other_stream_in_use(ctx, ctx->device, want) && !concurrent_pipelines_allowed())
        return fail(ctx, RTGL_ERR_STATE, "rtgl_set_stream: another context of this process renders on a different stream of this device; two path-tracing pipelines "
                                         "running concurrently on one device have produced wrong frames (DESIGN.md 5.2).  Bind the SAME stream to all of them, or set "
                                         "RTGL_AMD_ALLOW_CONCURRENT_PIPELINES=1 to take this over");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = want;
    register_ctx_stream(ctx, ctx->device, ctx->stream);
    return RTGL_OK;
}

extern "C" int rtgl_get_counters(rtgl_context *ctx, rtgl_counters *out)
{
    ENTER(ctx);
    if (!out) return fail(ctx, RTGL_ERR_INVALID, "out is NULL");
    if (!ctx->parts.empty()) {
        memset(out, 0, sizeof *out);
        for (rtgl_context *part : ctx->parts) {
            rtgl_counters c; const int rc = rtgl_get_counters(part, &c);
            if (rc) return fail(ctx, rc, part->error);
            out->paths += c.paths; out->segments += c.segments; out->triangle_tests += c.triangle_tests; out->candidates += c.candidates; out->env_lookups += c.env_lookups; out->culled_tests += c.culled_tests;
        }
        return RTGL_OK;
    }
    Counters c;
    HIPCHK(ctx, hipMemcpyAsync(&c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof *out);
    out->paths = c.paths; out->segments = c.segments; out->triangle_tests = c.tri_tests;
    out->candidates = c.candidates; out->env_lookups = c.env_lookups; out->culled_tests = c.culled_tests;
    return RTGL_OK;
}

extern "C" int rtgl_read_rng_state(rtgl_context *ctx, uint32_t *xyzw)
{
    ENTER(ctx);
    if (!xyzw) return fail(ctx, RTGL_ERR_INVALID, "xyzw is NULL");
    if (!ctx->parts.empty()) {                          // rows from their owners, global row order
        std::vector<uint32_t> local;
        for (rtgl_context *part : ctx->parts) {
            local.resize((size_t)std::max(part->local_rows, 1) * ctx->width * 4);
            const int rc = rtgl_read_rng_state(part, local.data());
            if (rc) return fail(ctx, rc, part->error);
            for (int lr = 0; lr < part->local_rows; ++lr)
                memcpy(xyzw + (size_t)rtgl_local_row_to_global(part, lr) * ctx->width * 4, local.data() + (size_t)lr * ctx->width * 4, (size_t)ctx->width * 16);
        }
        return RTGL_OK;
    }
    if (!ctx->opt_rng_state || !ctx->d_rng) return fail(ctx, RTGL_ERR_STATE, "option rng_state was not enabled before rendering");
    HIPCHK(ctx, hipMemcpyAsync(xyzw, ctx->d_rng, (size_t)ctx->local_rows * ctx->width * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RTGL_OK;
}

extern "C" int rtgl_set_option(rtgl_context *ctx, const char *key, int value)
{
    ENTER(ctx);
    FANOUT(ctx, rtgl_set_option(part, key, value));
    if (!key) return fail(ctx, RTGL_ERR_INVALID, "key is NULL");
    if (!strcmp(key, "kernel")) {
        if (value == RTGL_KERNEL_REMOVED_3)
            return fail(ctx, RTGL_ERR_INVALID, "kernel variant 3 (three waves per SIMD) was removed: it was not deterministic (DESIGN.md section 5); use 4");
        if (value < RTGL_KERNEL_MEGA || value > RTGL_KERNEL_WAVEFRONT_MFMA_SOLO)
            return fail(ctx, RTGL_ERR_INVALID, "unknown kernel variant");
        ctx->opt_kernel = value; ctx->kernel_explicit = true;
    } else if (!strcmp(key, "wf_rays")) {
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(ctx, RTGL_ERR_INVALID, "wf_rays must be 1, 2, 4 or 8");
        ctx->opt_wf_rays = value;
    } else if (!strcmp(key, "wf_chunk")) {
        if (value < kBoundGroup || value % kBoundGroup || (uint32_t)value > kMaxChunk) return fail(ctx, RTGL_ERR_INVALID, "wf_chunk must be a multiple of 64 in [64, 4096]");
        ctx->opt_wf_chunk = value;
    } else if (!strcmp(key, "debug_skip_exact")) {      // timing diagnostics only: the image is wrong
        if (value < 0 || value > 3) return fail(ctx, RTGL_ERR_INVALID, "debug_skip_exact must be 0, 1, 2 or 3");
        ctx->opt_debug_skip_exact = value;               // 1: survivors are dropped instead of tested; 2: broad phase rejects everything; 3: every segment through the list loop (image right)
    } else if (!strcmp(key, "mf_chunk_quads")) {
        if (value < 1 || (uint32_t)value > kMfMaxChunkQuads) return fail(ctx, RTGL_ERR_INVALID, "mf_chunk_quads must be in [1, 32]");
        ctx->opt_mf_chunk_quads = value;
    } else if (!strcmp(key, "scan_waves")) {
        if (value < 0 || value > 2) return fail(ctx, RTGL_ERR_INVALID, "scan_waves (waves per SIMD of the kernel-4 scan) must be 0 (default: two), 1 or 2");
        ctx->opt_scan_waves = value;
    } else if (!strcmp(key, "scan_dynamic")) {
        if (value < 0 || value > 4) return fail(ctx, RTGL_ERR_INVALID, "scan_dynamic must be 0 (chosen by the mesh), 1 (static turns), 2 (dynamic claims), 3 (planned: equal-cost intervals) or 4 (hybrid: turns + a claimed tail)");
        ctx->opt_scan_dynamic = value;
    } else if (!strcmp(key, "frame_batch")) {
        if (value < 1 || value > (int)kBatchMax) return fail(ctx, RTGL_ERR_INVALID, "frame_batch (consecutive frames traced in one set of launches) must be 1..16");
        ctx->opt_frame_batch = value;
    } else if (!strcmp(key, "cull")) {
        if (value < 0 || value > 3) return fail(ctx, RTGL_ERR_INVALID, "cull must be 0 (off), 1 (camera rays), 2 (every bounce, queues as they come) or 3 (camera rays + binned bounces)");
        ctx->opt_cull = value;
    } else if (!strcmp(key, "sort_min_rays")) {
        if (value < 0) return fail(ctx, RTGL_ERR_INVALID, "sort_min_rays (a bounce's queue is binned when at least this many rays are expected) must be >= 0");
        ctx->opt_sort_min_rays = value;
    } else if (!strcmp(key, "mf_group_quads")) {
        if (value < 1 || value > (int)kMfMaxGroupQuads || (value & (value - 1))) return fail(ctx, RTGL_ERR_INVALID, "mf_group_quads must be a power of two in [1, 64]");
        if (value != (int)ctx->mf_group_quads) ctx->tris_dirty = true;                         // local origins and A tiles are per group
        ctx->opt_mf_group_quads = value; ctx->group_explicit = true;
    } else if (!strcmp(key, "wf_packed")) {
        ctx->opt_wf_packed = value != 0;
    } else if (!strcmp(key, "wf_early")) {
        if (value < 0) return fail(ctx, RTGL_ERR_INVALID, "wf_early is the number of leading bounces with the wave-level edge short circuit (>= 0)");
        ctx->opt_wf_early = value;
    } else if (!strcmp(key, "wf_mode")) {
        if (value != kScalar && value != kLds) return fail(ctx, RTGL_ERR_INVALID, "wf_mode must be 0 (scalar) or 1 (lds)");
        ctx->opt_wf_mode = value;
    } else if (!strcmp(key, "rng_state")) ctx->opt_rng_state = value != 0;
    else if (!strcmp(key, "counters")) ctx->opt_counters = value != 0;
    else if (!strcmp(key, "kernel_timing")) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (value < 0) return fail(ctx, RTGL_ERR_INVALID, "kernel_timing must be 0 (off) or the sampling period in frames");
        ctx->opt_kernel_timing = value; ctx->kev_used = 0; ctx->kev_frame_start.clear(); ctx->timing_frame_counter = 0; ctx->timing_this_frame = false;
    } else return fail(ctx, RTGL_ERR_INVALID, std::string("unknown option ") + key);
    return RTGL_OK;
}

extern "C" int rtgl_get_option(rtgl_context *ctx, const char *key, int *value)
{
    ENTER(ctx);
    if (!key || !value) return fail(ctx, RTGL_ERR_INVALID, "NULL argument");
    if (!ctx->parts.empty()) { const int rc = rtgl_get_option(ctx->parts[0], key, value); return rc ? fail(ctx, rc, ctx->parts[0]->error) : RTGL_OK; }
    if (!strcmp(key, "kernel")) *value = ctx->opt_kernel;
    else if (!strcmp(key, "kernel_in_use")) *value = ctx->kernel_in_use;
    else if (!strcmp(key, "wf_rays")) *value = ctx->opt_wf_rays;
    else if (!strcmp(key, "wf_mode")) *value = ctx->opt_wf_mode;
    else if (!strcmp(key, "wf_chunk")) *value = ctx->opt_wf_chunk;
    else if (!strcmp(key, "wf_early")) *value = ctx->opt_wf_early;
    else if (!strcmp(key, "wf_packed")) *value = ctx->opt_wf_packed;
    else if (!strcmp(key, "mf_sets")) *value = kSoloSets;
    else if (!strcmp(key, "mf_chunk_quads")) *value = ctx->opt_mf_chunk_quads;
    else if (!strcmp(key, "mf_group_quads")) *value = (int)ctx->mf_group_quads;
    else if (!strcmp(key, "cull")) *value = ctx->opt_cull;
    else if (!strcmp(key, "sort_min_rays")) *value = ctx->opt_sort_min_rays;
    else if (!strcmp(key, "scan_waves")) *value = ctx->opt_scan_waves;
    else if (!strcmp(key, "scan_dynamic")) *value = ctx->opt_scan_dynamic;
    else if (!strcmp(key, "frame_batch")) *value = ctx->opt_frame_batch;
    else if (!strcmp(key, "rng_state")) *value = ctx->opt_rng_state;
    else if (!strcmp(key, "counters")) *value = ctx->opt_counters;
    else if (!strcmp(key, "kernel_timing")) *value = ctx->opt_kernel_timing;
    else if (!strcmp(key, "cand_region_pairs")) *value = (int)ctx->cand_region_pairs;      // kernel 4: current capacity of one wave's candidate region
    else if (!strcmp(key, "device_mbytes")) {                                              // device memory held by this context (MiB, rounded up)
        size_t b = (size_t)std::max(ctx->local_rows, 1) * ctx->width * 16;
        b += (size_t)ctx->wave_capacity * (68 + 8) * 2 + (ctx->wave_multi ? (size_t)std::max(ctx->local_rows, 1) * ctx->width * 64 : 0);
        b += ((size_t)ctx->cand_regions * ctx->cand_region_pairs) * 8 + (size_t)ctx->cand_regions * 4;
        b += (size_t)ctx->n_tri_visits * (sizeof(TriEdges) + sizeof(TriPlane) + 4 + 112) + (size_t)ctx->n_vec4 * 16 + (size_t)ctx->env_faces * ctx->env_w * ctx->env_h * ctx->env_c;
        if (ctx->d_rng) b += (size_t)std::max(ctx->local_rows, 1) * ctx->width * 16;
        b += ctx->batch_capacity * 16;
        b += ctx->stage_capacity * 76 + (ctx->sort_bits_alloc ? ((size_t)8 << ctx->sort_bits_alloc) : 0);
        *value = (int)((b + (1u << 20) - 1) >> 20);
    }
    else return fail(ctx, RTGL_ERR_INVALID, std::string("unknown option ") + key);
    return RTGL_OK;
}
