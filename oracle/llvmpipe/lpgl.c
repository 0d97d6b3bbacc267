/*
 * lpgl.c -- headless OpenGL 4.3 compute runner on Mesa llvmpipe (TEST INFRASTRUCTURE ONLY).
 *
 * Purpose: execute the *reference's own* compute shader (the file is passed in by path /
 * source string at run time and is never stored in this repository) on the CPU, so that
 * golden vectors for tests/golden/ can be generated in the build container and so that the
 * reference can be re-timed on host cores.  Nothing in the product path links or loads this.
 *
 * How: there is no X server, EGL or OSMesa in the image, so the swrast DRI driver is loaded
 * directly (dlopen of swrast_dri.so + libglapi) and driven through the DRI_Core / DRI_SWRast
 * loader interface declared in <GL/internal/dri_interface.h>.  After the context is current
 * everything is plain GL 4.3: SSBOs, an RGBA32F image, a cube map, uniforms, glDispatchCompute.
 *
 * What it replaces on the reference side (for the oracle only): the GL half of
 * src/renderer.cpp:21-64 (object creation), :89-134 (bind/uniform/dispatch) and
 * src/gfx/gl.cpp:241-260 (cube map upload), restated as a flat C API usable through ctypes.
 *
 * Build: see oracle/Makefile (gcc -shared -fPIC lpgl.c -ldl -> oracle/_ref/liblpgl.so).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GL_GLEXT_PROTOTYPES 0
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

#ifndef LPGL_DRI_PATH
#define LPGL_DRI_PATH "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"
#endif

/* ---- DRI plumbing -------------------------------------------------------------------- */

static void ld_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void ld_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void ld_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void ld_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension g_loader_ext = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ld_get_drawable_info,
    .putImage = ld_put_image,
    .getImage = ld_get_image,
    .putImage2 = ld_put_image2,
    .getImage2 = ld_get_image2,
};
static const __DRIextension *g_loader_exts[] = { &g_loader_ext.base, NULL };

static void *g_glapi, *g_dri;
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static void *(*g_getproc)(const char *);
static char g_err[512];

#define GLF(ret, name, ...) typedef ret (APIENTRY *pf_##name)(__VA_ARGS__); static pf_##name p_##name
GLF(const GLubyte *, glGetString, GLenum);
GLF(GLenum, glGetError, void);
GLF(GLuint, glCreateShader, GLenum);
GLF(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLF(void, glCompileShader, GLuint);
GLF(void, glGetShaderiv, GLuint, GLenum, GLint *);
GLF(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(GLuint, glCreateProgram, void);
GLF(void, glAttachShader, GLuint, GLuint);
GLF(void, glLinkProgram, GLuint);
GLF(void, glGetProgramiv, GLuint, GLenum, GLint *);
GLF(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(void, glUseProgram, GLuint);
GLF(void, glDeleteProgram, GLuint);
GLF(void, glDeleteShader, GLuint);
GLF(GLint, glGetUniformLocation, GLuint, const GLchar *);
GLF(void, glUniform1i, GLint, GLint);
GLF(void, glUniform1ui, GLint, GLuint);
GLF(void, glUniform1f, GLint, GLfloat);
GLF(void, glUniform3f, GLint, GLfloat, GLfloat, GLfloat);
GLF(void, glGenBuffers, GLsizei, GLuint *);
GLF(void, glDeleteBuffers, GLsizei, const GLuint *);
GLF(void, glBindBuffer, GLenum, GLuint);
GLF(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum);
GLF(void, glBindBufferBase, GLenum, GLuint, GLuint);
GLF(void, glGetBufferSubData, GLenum, GLintptr, GLsizeiptr, void *);
GLF(void, glGenTextures, GLsizei, GLuint *);
GLF(void, glDeleteTextures, GLsizei, const GLuint *);
GLF(void, glBindTexture, GLenum, GLuint);
GLF(void, glActiveTexture, GLenum);
GLF(void, glTexParameteri, GLenum, GLenum, GLint);
GLF(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
GLF(void, glGetTexImage, GLenum, GLint, GLenum, GLenum, void *);
GLF(void, glBindImageTexture, GLuint, GLuint, GLint, GLboolean, GLint, GLenum, GLenum);
GLF(void, glDispatchCompute, GLuint, GLuint, GLuint);
GLF(void, glMemoryBarrier, GLbitfield);
GLF(void, glFinish, void);
GLF(void, glPixelStorei, GLenum, GLint);

#define LOAD(name) do { p_##name = (pf_##name)g_getproc(#name); \
    if (!p_##name) { snprintf(g_err, sizeof g_err, "missing GL entry point %s", #name); return -5; } } while (0)

const char *lpgl_last_error(void) { return g_err; }

int lpgl_init(void)
{
    if (g_ctx) return 0;
    g_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!g_glapi) { snprintf(g_err, sizeof g_err, "dlopen libglapi: %s", dlerror()); return -1; }
    const char *path = getenv("LPGL_DRI_PATH");
    g_dri = dlopen(path ? path : LPGL_DRI_PATH, RTLD_NOW | RTLD_GLOBAL);
    if (!g_dri) { snprintf(g_err, sizeof g_err, "dlopen swrast_dri: %s", dlerror()); return -1; }
    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(g_dri, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_exts) { snprintf(g_err, sizeof g_err, "no %s_swrast", __DRI_DRIVER_GET_EXTENSIONS); return -2; }
    const __DRIextension **exts = get_exts();
    for (int i = 0; exts && exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) {
        snprintf(g_err, sizeof g_err, "DRI_Core/DRI_SWRast(v4) not offered"); return -2;
    }
    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) { snprintf(g_err, sizeof g_err, "createNewScreen2 failed"); return -3; }
    const uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3 };
    unsigned cerr = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &cerr, NULL);
    if (!g_ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs failed (%u)", cerr); return -3; }
    g_draw = g_swrast->createNewDrawable(g_screen, configs[0], NULL);
    if (!g_draw || !g_core->bindContext(g_ctx, g_draw, g_draw)) {
        snprintf(g_err, sizeof g_err, "bindContext failed"); return -4;
    }
    g_getproc = (void *(*)(const char *))dlsym(g_glapi, "_glapi_get_proc_address");
    if (!g_getproc) { snprintf(g_err, sizeof g_err, "no _glapi_get_proc_address"); return -5; }
    LOAD(glGetString); LOAD(glGetError); LOAD(glCreateShader); LOAD(glShaderSource); LOAD(glCompileShader);
    LOAD(glGetShaderiv); LOAD(glGetShaderInfoLog); LOAD(glCreateProgram); LOAD(glAttachShader);
    LOAD(glLinkProgram); LOAD(glGetProgramiv); LOAD(glGetProgramInfoLog); LOAD(glUseProgram);
    LOAD(glDeleteProgram); LOAD(glDeleteShader); LOAD(glGetUniformLocation); LOAD(glUniform1i);
    LOAD(glUniform1ui); LOAD(glUniform1f); LOAD(glUniform3f); LOAD(glGenBuffers); LOAD(glDeleteBuffers);
    LOAD(glBindBuffer); LOAD(glBufferData); LOAD(glBindBufferBase); LOAD(glGetBufferSubData);
    LOAD(glGenTextures); LOAD(glDeleteTextures); LOAD(glBindTexture); LOAD(glActiveTexture);
    LOAD(glTexParameteri); LOAD(glTexImage2D); LOAD(glGetTexImage); LOAD(glBindImageTexture);
    LOAD(glDispatchCompute); LOAD(glMemoryBarrier); LOAD(glFinish); LOAD(glPixelStorei);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    return 0;
}

const char *lpgl_info(int which)
{
    if (!g_ctx) return "";
    return (const char *)p_glGetString(which == 0 ? GL_VERSION : which == 1 ? GL_RENDERER : GL_SHADING_LANGUAGE_VERSION);
}

int lpgl_get_error(void) { return (int)p_glGetError(); }

/* Compile+link one compute shader from a source string.  Returns program id (>0) or <0;
 * the info log (warnings included) is copied to log[]. */
int lpgl_compute_program(const char *src, char *log, int loglen)
{
    if (log && loglen > 0) log[0] = 0;
    GLuint sh = p_glCreateShader(GL_COMPUTE_SHADER);
    p_glShaderSource(sh, 1, &src, NULL);
    p_glCompileShader(sh);
    GLint ok = 0, n = 0;
    p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (log && loglen > 1) { p_glGetShaderInfoLog(sh, loglen - 1, &n, log); log[n] = 0; }
    if (!ok) { p_glDeleteShader(sh); snprintf(g_err, sizeof g_err, "compute shader compile failed"); return -1; }
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, sh);
    p_glLinkProgram(prog);
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        if (log && loglen > 1) { p_glGetProgramInfoLog(prog, loglen - 1, &n, log); log[n] = 0; }
        p_glDeleteProgram(prog); p_glDeleteShader(sh);
        snprintf(g_err, sizeof g_err, "compute program link failed"); return -2;
    }
    p_glDeleteShader(sh);
    return (int)prog;
}

void lpgl_delete_program(int prog) { p_glDeleteProgram((GLuint)prog); }

/* SSBO: create (id==0) or re-fill, glBufferData verbatim (src/gfx/gl.h:93-97), bind to `binding`
 * (src/renderer.cpp:89-93).  bytes==0 leaves the binding without storage like the reference does
 * for buffers that were never set. */
int lpgl_ssbo(int id, int binding, const void *data, long bytes)
{
    GLuint b = (GLuint)id;
    if (!b) p_glGenBuffers(1, &b);
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, b);
    if (bytes > 0) p_glBufferData(GL_SHADER_STORAGE_BUFFER, bytes, data, GL_STATIC_DRAW);
    p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)binding, b);
    return (int)b;
}

void lpgl_ssbo_read(int id, void *out, long bytes)
{
    p_glMemoryBarrier(GL_ALL_BARRIER_BITS);
    p_glFinish();
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, (GLuint)id);
    p_glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, bytes, out);
}

void lpgl_delete_buffer(int id) { GLuint b = (GLuint)id; p_glDeleteBuffers(1, &b); }

/* RGBA32F accumulation image (src/renderer.cpp:58-63); init may be NULL (contents then
 * undefined by GL, zero on llvmpipe) */
int lpgl_image_rgba32f(int w, int h, const float *init)
{
    GLuint t;
    p_glGenTextures(1, &t);
    p_glActiveTexture(GL_TEXTURE0 + 1);
    p_glBindTexture(GL_TEXTURE_2D, t);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, init);
    return (int)t;
}

void lpgl_bind_image(int tex)
{
    p_glBindImageTexture(0, (GLuint)tex, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
}

void lpgl_read_image(int tex, float *out)
{
    p_glMemoryBarrier(GL_ALL_BARRIER_BITS);
    p_glFinish();
    p_glActiveTexture(GL_TEXTURE0 + 1);
    p_glBindTexture(GL_TEXTURE_2D, (GLuint)tex);
    p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, out);
}

/* Same readback as the reference's save path (src/renderer.cpp:218-223): 8-bit, driver clamps */
void lpgl_read_image_u8(int tex, unsigned char *out)
{
    p_glMemoryBarrier(GL_ALL_BARRIER_BITS);
    p_glFinish();
    p_glActiveTexture(GL_TEXTURE0 + 1);
    p_glBindTexture(GL_TEXTURE_2D, (GLuint)tex);
    p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, out);
}

void lpgl_delete_texture(int tex) { GLuint t = (GLuint)tex; p_glDeleteTextures(1, &t); }

/* Cube map on texture unit 0 (where the reference's compute sampler ends up reading, SURVEY A.9
 * item 11): faces in +X,-X,+Y,-Y,+Z,-Z order, 8-bit, `channels` 3 or 4, `nfaces` may be < 6 to
 * reproduce the incomplete-cube case (src/gfx/gl.cpp:246-258).  Parameters per
 * src/renderer.cpp:163-172. */
int lpgl_cubemap(const unsigned char *faces, int nfaces, int w, int h, int channels)
{
    GLuint t;
    p_glGenTextures(1, &t);
    p_glActiveTexture(GL_TEXTURE0);
    p_glBindTexture(GL_TEXTURE_CUBE_MAP, t);
    GLenum fmt = channels == 4 ? GL_RGBA : GL_RGB;
    for (int i = 0; i < nfaces && i < 6; i++)
        p_glTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + i, 0, (GLint)fmt, w, h, 0, fmt, GL_UNSIGNED_BYTE,
                       faces + (size_t)i * w * h * channels);
    p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    p_glTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_R, GL_CLAMP_TO_EDGE);
    return (int)t;
}

void lpgl_use(int prog) { p_glUseProgram((GLuint)prog); }
int lpgl_uniform1i(int prog, const char *name, int v)
{ GLint l = p_glGetUniformLocation((GLuint)prog, name); if (l < 0) return -1; p_glUniform1i(l, v); return 0; }
int lpgl_uniform1ui(int prog, const char *name, unsigned v)
{ GLint l = p_glGetUniformLocation((GLuint)prog, name); if (l < 0) return -1; p_glUniform1ui(l, v); return 0; }
int lpgl_uniform1f(int prog, const char *name, float v)
{ GLint l = p_glGetUniformLocation((GLuint)prog, name); if (l < 0) return -1; p_glUniform1f(l, v); return 0; }
int lpgl_uniform3f(int prog, const char *name, float x, float y, float z)
{ GLint l = p_glGetUniformLocation((GLuint)prog, name); if (l < 0) return -1; p_glUniform3f(l, x, y, z); return 0; }

/* glDispatchCompute + image barrier (src/renderer.cpp:131-134), then glFinish so callers can time it */
void lpgl_dispatch(int gx, int gy, int gz)
{
    p_glDispatchCompute((GLuint)gx, (GLuint)gy, (GLuint)gz);
    p_glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT | GL_SHADER_STORAGE_BARRIER_BIT);
    p_glFinish();
}
