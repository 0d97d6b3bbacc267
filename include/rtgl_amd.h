/*
 * rtgl_amd.h -- C ABI of the MI355X-native progressive path tracer (drop-in boundary).
 *
 * The reference (gue-ni/raytracer.glsl) has no plugin / FFI layer: its hot path sits behind the
 * C++ class `Renderer : public Window` (reference src/renderer.h:127-148) and the OpenGL driver.
 * This header is the flat C boundary a maintainer binds instead of the GL calls; the C++ facade
 * in include/rtgl/renderer.h (same class names and method signatures as the reference) is a thin
 * layer over exactly these entry points.  Each entry point cites the reference interface it
 * replaces.  Plain pointers and sizes only; every call returns 0 on success or a negative
 * rtgl_status, with text available from rtgl_last_error().
 *
 * Threading (reference: single-threaded, everything on the thread that owns the GL context,
 * src/window.cpp:13): one host thread per context; calls on one context must not overlap.
 * Ownership (reference: setters copy synchronously via glBufferData, src/gfx/gl.h:93-97): every
 * upload copies; the caller may free its buffer as soon as the call returns.
 */
#ifndef RTGL_AMD_H
#define RTGL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtgl_context rtgl_context;

typedef enum rtgl_status {
    RTGL_OK = 0,
    RTGL_ERR_INVALID = -1,   /* bad argument */
    RTGL_ERR_DEVICE = -2,    /* HIP runtime error (text in rtgl_last_error) */
    RTGL_ERR_NO_DEVICE = -3, /* no usable gfx950 device: the library has no CPU fallback */
    RTGL_ERR_STATE = -4      /* call not valid in the current state */
} rtgl_status;

/* The 17 uniforms of reference shaders/raytracer.glsl:62-81, set per frame by
 * src/renderer.cpp:96-123 (glUniform* by name).  camera_fov is in radians (renderer.cpp:116).
 * `time` is uploaded by the reference but never read by the shader; kept for symmetry. */
typedef struct rtgl_frame_params {
    int32_t  frames;       /* u_frames  = Window::m_frames after the pre-render increment (window.cpp:42) */
    uint32_t samples;      /* u_samples (renderer.h:168: always 1 in the reference) */
    uint32_t max_bounce;   /* u_max_bounce */
    float    time;         /* u_time (dead) */
    float    background[3];
    int32_t  reset_flag;   /* u_reset_flag */
    int32_t  use_envmap;   /* u_use_envmap (false when no cube map is set, renderer.cpp:104-110) */
    int32_t  use_dof;      /* u_use_dof */
    int32_t  random;       /* u_random = rand() once per frame (renderer.cpp:102) */
    float    camera_position[3];
    float    camera_fov;
    float    camera_aperture;
    float    camera_focal_length;
    float    camera_forward[3];
    float    camera_up[3];
    float    camera_right[3];
} rtgl_frame_params;

/* Device-side work counters of the last rendered frame (no reference counterpart; feeds the
 * Msegments/s and Gtests/s figures of SURVEY.md 8(d1)). */
typedef struct rtgl_counters {
    uint64_t paths;          /* pixel-samples started */
    uint64_t segments;       /* bounce iterations executed */
    uint64_t triangle_tests; /* ray x triangle edge-function evaluations */
    uint64_t candidates;     /* tests that reached the exact (reference-order) re-evaluation */
    uint64_t env_lookups;
    uint64_t culled_tests;   /* of triangle_tests: pairs skipped by the packet cull (certified rejections, never evaluated) */
    uint64_t reserved[2];
} rtgl_counters;

/* Kernel variants (rtgl_set_option "kernel"; the environment variable RTGL_AMD_KERNEL=0, 1, 2 or 4 changes the default of new
 * contexts).  All produce bit-identical images.  Unless a variant was requested explicitly, a scene without triangles is rendered
 * with RTGL_KERNEL_MEGA (there is no scan to split off; one launch per frame). */
enum {
    RTGL_KERNEL_MEGA = 0,            /* one launch per frame, one lane per pixel, whole path in registers */
    RTGL_KERNEL_WAVEFRONT = 1,       /* one fused launch per bounce over the compacted ray queue */
    RTGL_KERNEL_WAVEFRONT_SPLIT = 2, /* per bounce: intersect (ray blocks x triangle chunks, fp32 VALU filter) + shade */
    RTGL_KERNEL_REMOVED_3 = 3,       /* (round 1's three-waves-per-SIMD matrix-core scan: not deterministic, removed; refused) */
    RTGL_KERNEL_WAVEFRONT_MFMA_SOLO = 4 /* default: as 2, with a conservative bf16 matrix-core broad phase in front of the exact test: one or two
                                         * waves per SIMD, persistent blocks, A tiles in LDS, hand-ordered instruction stream, exact tests in a
                                         * separate narrow-phase kernel */
};

/* -- lifetime: replaces Renderer::Renderer(width,height) GL object creation (src/renderer.cpp:21-64).
 * The accumulation image is RGBA32F, width x height, zero-initialised (the reference leaves it
 * undefined).  device = HIP device ordinal. */
int rtgl_create(rtgl_context **out, int width, int height, int device);

/* Same, but this context owns only the row strips s with (s % world) == rank, where strip s covers
 * image rows [s*strip_rows, (s+1)*strip_rows).  Pixel seeds and camera rays use absolute pixel
 * coordinates, so the union over ranks is bit-identical to a single-context render.  strip_rows
 * must be a multiple of 8.  (No reference counterpart: SURVEY.md 8(e).) */
int rtgl_create_tiled(rtgl_context **out, int width, int height, int device, int rank, int world, int strip_rows);

/* Single-process multi-device context: the image is cut into strips of strip_rows rows, strip s belongs to
 * devices[s % n_devices]; every device holds the whole scene and renders its strips (one dispatch per device replaces the single
 * glDispatchCompute of src/renderer.cpp:129-134); the tile buffers are gathered to devices[0] (peer copies over xGMI, one 2-D copy
 * per device) when the image is read -- rtgl_read_image_* do it implicitly, rtgl_gather_tiles explicitly (then rtgl_device_image
 * is the assembled image on devices[0]).  Every other entry point takes the handle like a single-device one: uploads and options
 * go to all devices, counters are summed, rtgl_last_frame_ms is the slowest device.  Bit-identical to a single-device render.
 * The same ordinal may appear more than once.  rtgl_bind_device_image / rtgl_set_stream are refused on such a handle.
 * rtgl_render_frame hands each device's frame to a submit thread of the context's own (created here, joined in rtgl_destroy) and returns
 * when all have submitted; the environment variable RTGL_AMD_MULTI_THREADS=0, read here, keeps submission on the caller's thread. */
int rtgl_create_multi(rtgl_context **out, int width, int height, const int *devices, int n_devices, int strip_rows);
int rtgl_device_count(const rtgl_context *ctx);   /* 1 for a single-device context */
int rtgl_gather_tiles(rtgl_context *ctx);         /* enqueue the gather on devices[0]'s stream (no-op for a single-device context); every device's
                                                    * stream then waits for the copies before its next frame (the tiles are being read) */

void rtgl_destroy(rtgl_context *ctx);
const char *rtgl_last_error(const rtgl_context *ctx); /* ctx may be NULL: error of the last failed create */

/* -- scene upload: replaces the set_* family (src/renderer.cpp:151-216), layouts per the shader's
 * buffer declarations (shaders/raytracer.glsl:11-60).  count = number of elements. */
int rtgl_upload_spheres(rtgl_context *ctx, const void *spheres, uint32_t count);     /* 32 B each; set_spheres :151-155 */
int rtgl_upload_materials(rtgl_context *ctx, const void *materials, uint32_t count); /* 32 B each; set_materials :157-161 */
int rtgl_upload_meshes(rtgl_context *ctx, const void *meshes, uint32_t count);       /* 16 B each; set_meshes :181-185 */
int rtgl_upload_vertices(rtgl_context *ctx, const void *vec4s, uint32_t vec4_count); /* 16 B each, 3 per triangle; set_vertices :174-179 */
int rtgl_upload_nodes(rtgl_context *ctx, const void *nodes, uint32_t count);         /* 48 B each; set_nodes :187-191 */
/* faces: 6 (or fewer) tightly packed 8-bit images in +X,-X,+Y,-Y,+Z,-Z order, channels 3 or 4;
 * replaces CubemapTexture's constructor (src/gfx/gl.cpp:241-260) + set_envmap (renderer.cpp:163-172).
 * nfaces < 6 reproduces the incomplete cube of a failed face load (lookups return black). */
int rtgl_upload_envmap(rtgl_context *ctx, const uint8_t *faces, int nfaces, int width, int height, int channels);

/* -- per frame: replaces the uniform uploads + glDispatchCompute + glMemoryBarrier of
 * Renderer::render (src/renderer.cpp:96-134).  rtgl_render_frame enqueues on the context's stream
 * and returns; rtgl_synchronize waits.  (Option "frame_batch" > 1: it may hold the frame back until the batch is full, see below.) */
int rtgl_set_frame_params(rtgl_context *ctx, const rtgl_frame_params *params);
int rtgl_render_frame(rtgl_context *ctx);
int rtgl_synchronize(rtgl_context *ctx);

/* -- image access: replaces glGetTexImage in save_to_file (src/renderer.cpp:218-223).
 * f32: RGBA32F rows bottom-up exactly as stored (row 0 = pixel y 0).  For a tiled context the
 * buffer holds only the local strips, packed in increasing strip order (rtgl_local_rows rows).
 * u8: clamp to [0,1], scale by 255, round to nearest (what GL_UNSIGNED_BYTE readback does);
 * flip != 0 writes the top row first like stbi_write_png's flipped output (renderer.cpp:240). */
int rtgl_read_image_f32(rtgl_context *ctx, float *rgba);
int rtgl_read_image_u8(rtgl_context *ctx, uint8_t *rgba, int flip);
int rtgl_write_image_f32(rtgl_context *ctx, const float *rgba); /* preload / resume the accumulation image */
int rtgl_clear_image(rtgl_context *ctx);
int rtgl_local_rows(const rtgl_context *ctx);      /* rows held by this context (== height when not tiled) */
int rtgl_local_row_to_global(const rtgl_context *ctx, int local_row);

/* -- plumbing for callers that own device memory / streams (PyTorch, RCCL) */
void *rtgl_device_image(rtgl_context *ctx);                 /* device pointer of the local RGBA32F buffer (NULL: see rtgl_last_error) */
int rtgl_bind_device_image(rtgl_context *ctx, void *dptr);  /* render into caller-owned device memory (local_rows*width*16 B) */
/* hipStream_t; NULL restores the context's own stream.  All contexts of a process on one device submit to ONE stream of the library's by
 * default, so that their pipelines never run concurrently (several path-tracing pipelines at once on one MI355X have produced wrong
 * frames: DESIGN.md 5.2).  rtgl_set_stream therefore FAILS with RTGL_ERR_STATE when another live context of the process renders on a
 * different stream of the same device; bind the same stream to all of them, or set RTGL_AMD_ALLOW_CONCURRENT_PIPELINES=1 and take the
 * ordering over.  The library cannot see other PROCESSES on the device: do not run two path-tracing processes on one GPU at the same time.
 * With "frame_batch" > 1, synchronising the bound stream is NOT enough to know that a frame has been submitted: only rtgl_synchronize and
 * the read-out calls submit the frames a batching context holds back. */
int rtgl_set_stream(rtgl_context *ctx, void *hip_stream);

/* -- diagnostics */
int rtgl_get_counters(rtgl_context *ctx, rtgl_counters *out);  /* synchronises */
int rtgl_read_rng_state(rtgl_context *ctx, uint32_t *xyzw);    /* per local pixel final PCG4D state of the last frame; needs option "rng_state"=1 */
/* keys: "kernel" (enum above), "wf_rays" (rays per lane 1/2/4/8), "wf_mode" (0 scalar-fed, 1 LDS tiles),
 * "wf_chunk" (triangles per work item of the split intersect kernel, multiple of 64), "wf_early" (leading bounces
 * that use the wave-level edge short circuit), "wf_packed" (v_pk_fma_f32 ray pairs on/off), "mf_chunk_quads" (kernel 4: 40-triangle quads
 * per work item = per block's LDS-resident chunk, 1..32), "scan_waves" (waves per SIMD of the kernel-4 scan: 0 default (= 2), 1, 2), "scan_dynamic" (work distribution of the kernel-4 scan: 0 chosen by the mesh (default), 1 static, 2 dynamic), "cull" (packet culling: a granule of 128 rays skips the tiles
 * of 10 triangles for which every one of its rays is certified to be rejected by the reference's own test: 0 off, 1 on the camera-ray
 * bounce, 2 on every bounce with the queues as they come, 3 (default) on the camera-ray bounce and on every bounce whose queue was BINNED
 * -- moved into (direction cell, origin cell) order between the bounces, which is what makes its granules coherent), "sort_min_rays"
 * (cull 3: a bounce's queue is binned when at least this many rays are expected, default 131072), "mf_group_quads" (quads
 * sharing one local origin: a power of two up to 64; changing it rebuilds the broad-phase data at the next frame),
 * "rng_state", "counters", "kernel_timing" (0 off; N > 0: every N-th frame since the last rtgl_timing_reset carries HIP
 * event pairs around its dominant-kernel launches), "frame_batch" (1 (default) .. 16, also RTGL_AMD_FRAME_BATCH: with B > 1 rtgl_render_frame
 * only records the frame until B frames are waiting, then traces them in ONE set of launches and applies their results to the image in
 * frame order -- bit-identical to frame-by-frame, B times the rays per launch (what a rank of a multi-GPU run lacks).  Every other entry
 * point submits the waiting frames first, so the image a caller reads is always complete; frames that differ in samples, bounce limit,
 * environment switch or background close a batch early; with "counters", "rng_state" or "kernel_timing" on, with more than one sample
 * per frame and for scenes without triangles frames are rendered one by one; rtgl_destroy submits frames that are still waiting;
 * rtgl_device_image returns NULL when that submission fails) */
int rtgl_set_option(rtgl_context *ctx, const char *key, int value);
int rtgl_get_option(rtgl_context *ctx, const char *key, int *value);   /* also "kernel_in_use": the variant the last frame ran */
/* elapsed GPU milliseconds of the last rtgl_render_frame (HIP events on the context's stream) */
int rtgl_last_frame_ms(rtgl_context *ctx, float *ms);
/* Per-kernel GPU time of the last frame, from HIP events recorded around every launch of the dominant
 * kernel on the context's stream.  Needs option "kernel_timing"=1 before rendering. */
typedef struct rtgl_frame_timing {
    float    frame_ms;            /* whole frame, first launch to last */
    float    intersect_ms;        /* sum over the ray x triangle kernel launches (the path-trace megakernel when kernel = 0) */
    uint32_t intersect_launches;
    uint32_t reserved;
} rtgl_frame_timing;
int rtgl_last_frame_timing(rtgl_context *ctx, rtgl_frame_timing *out);
/* Same figures summed over every frame rendered since the last rtgl_timing_reset (or since "kernel_timing"
 * was enabled): lets a caller keep frames queued back to back and read the HIP-event times once at the
 * end.  frames_out (may be NULL) receives the number of frames covered.  Synchronises. */
int rtgl_accumulated_timing(rtgl_context *ctx, rtgl_frame_timing *out, uint32_t *frames_out);
int rtgl_timing_reset(rtgl_context *ctx);

#ifdef __cplusplus
}
#endif
#endif /* RTGL_AMD_H */
