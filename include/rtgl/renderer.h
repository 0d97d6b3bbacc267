// renderer.h -- C++ facade with the reference's Renderer API (src/renderer.h:22-148) over the C ABI in
// include/rtgl_amd.h.  A program written against the reference's `Renderer` (see src/main.cpp there:
// construct, set_* / load_obj / transform, run()) compiles against this header unchanged and links
// librtgl_amd.so instead of OpenGL.  Header-only; any C++17 host compiler.
//
// Kept from the reference: class and method names, argument meaning, POD layouts (static_asserted
// against the shader's buffer strides), copy-on-set ownership, "print and continue" error behaviour
// (src/gfx/gl.cpp:79-94,254-258; src/renderer.cpp:264-267), frame / reset bookkeeping
// (src/renderer.cpp:98,123-127), the camera controls (src/renderer.cpp:311-420).
// Deliberate differences, each one where the reference is in GL-undefined or platform-dependent
// territory: Mesh is 16 bytes on every compiler (item 13); the accumulation image starts
// zeroed (item 9); set_kdtree(vec4) is accepted but only uploads the vertices (item 15: the shader would
// read triangle nodes as sphere ranges); the ImGui panel does not exist (public setters replace it).
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../rtgl_amd.h"
#include "glm_shim.h"
#include "kdtree.h"
#include "png_io.h"
#include "window.h"

struct Triangle
{
    glm::vec4 v[3];
    AABB bounds() const { return {glm::min(v[0], glm::min(v[1], v[2])), glm::max(v[0], glm::max(v[1], v[2]))}; }
};
static_assert(sizeof(Triangle) == 48, "3 x vec4 per triangle (shaders/raytracer.glsl:54-56)");

inline std::vector<Triangle> to_triangles(const std::vector<glm::vec4> &vertices)
{
    std::vector<Triangle> out(vertices.size() / 3);     // a trailing partial triangle is dropped
    for (size_t i = 0; i < out.size(); ++i)
        for (int k = 0; k < 3; ++k) out[i].v[k] = vertices[i * 3 + k];
    return out;
}

struct alignas(16) Sphere {
    glm::vec3 center;
    float radius;
    int material = 0;
    Sphere(const glm::vec3 &center_, float radius_, int mat = 0) : center(center_), radius(radius_), material(mat) {}
    AABB bounds() const { return {glm::vec4(center - radius, 0.0f), glm::vec4(center + radius, 0.0f)}; }
};
static_assert(sizeof(Sphere) == 32, "Sphere stride (std140, shaders/raytracer.glsl:11-15)");

// stream form of the reference (src/renderer.h:64-68)
inline std::ostream &operator<<(std::ostream &os, const Sphere &obj)
{
    os << "Sphere { c = " << obj.center << ", r = " << obj.radius << " }";
    return os;
}

enum MaterialType : unsigned int { DIFFUSE = 0, SPECULAR = 1, TRANSMISSIVE = 2 };

struct alignas(16) Material {
    glm::vec4 albedo;        // rgb + smoothness in w
    glm::vec3 emission;
    MaterialType type;
    Material(const glm::vec3 &albedo_, const glm::vec3 &emission_ = glm::vec3(0.0f), float smoothness = 0.0f,
             const MaterialType &type_ = DIFFUSE)
        : albedo(albedo_, smoothness), emission(emission_), type(type_) {}
};
static_assert(sizeof(Material) == 32, "Material stride (std140, shaders/raytracer.glsl:17-21)");

struct alignas(16) Mesh {
    unsigned int start;   // first triangle
    unsigned int size;    // triangle count
    int material;         // unused by the shader
    Mesh(unsigned int start_, unsigned int size_, int mat = 0) : start(start_), size(size_), material(mat) {}
};
static_assert(sizeof(Mesh) == 16, "Mesh stride (std140, shaders/raytracer.glsl:23-27)");

namespace gfx {
// [0,255] -> [0,1] (src/gfx/util.h:11-21)
template <typename T> constexpr glm::vec3 rgb(T r, T g, T b) { return glm::vec3(static_cast<float>(r), static_cast<float>(g), static_cast<float>(b)) / 255.0f; }
constexpr glm::vec3 rgb(uint32_t hex) { return rgb((hex & 0xff0000U) >> 16, (hex & 0x00ff00U) >> 8, (hex & 0x0000ffU) >> 0); }
namespace gl {
// Six 8-bit face images, +X,-X,+Y,-Y,+Z,-Z (src/gfx/gl.h:209-212, src/gfx/gl.cpp:241-260).  A face that
// fails to load prints "failed to load image" and stops, leaving the cube incomplete (lookups return
// black), exactly as the reference does.
struct CubemapTexture {
    int width = 0, height = 0, channels = 0, faces = 0;
    std::vector<uint8_t> data;
    CubemapTexture(const std::array<std::string, 6> &paths, bool flip_vertically = false)
    {
        for (int i = 0; i < 6; ++i) {
            rtgl::Image8 img;
            if (!rtgl::read_png(paths[i], img, flip_vertically) || (faces > 0 && (img.width != width || img.height != height || img.channels != channels))) {
                std::cerr << "failed to load image " << paths[i] << std::endl;
                break;
            }
            width = img.width; height = img.height; channels = img.channels;
            data.insert(data.end(), img.pixels.begin(), img.pixels.end());
            faces++;
        }
    }
    // from memory (no reference counterpart): faces tightly packed in the same order
    CubemapTexture(const uint8_t *pixels, int nfaces, int w, int h, int c)
        : width(w), height(h), channels(c), faces(nfaces), data(pixels, pixels + (size_t)nfaces * w * h * c) {}
};
}  // namespace gl
}  // namespace gfx
using namespace gfx::gl;

inline glm::vec3 vector_from_spherical(float pitch, float yaw)
{
    return {std::cos(yaw) * std::sin(pitch), std::cos(pitch), std::sin(yaw) * std::sin(pitch)};
}

struct Camera {
    glm::vec3 position;
    float fov = 45.0f;
    float focal_length = 10.0f;
    float aperture = 0.001f;
    float pitch = (float)(M_PI / 2);
    float yaw = (float)(M_PI / 2);
    glm::vec3 forward = {0.0f, 0.0f, 1.0f};
    glm::vec3 up = {0.0f, 1.0f, 0.0f};
    glm::vec3 right = {-1.0f, 0.0f, 0.0f};
    Camera(const glm::vec3 &position_, float fov_) : position(position_), fov(fov_) {}
};

class Renderer : public Window
{
public:
    // src/renderer.cpp:21-64; `device` and the tiling arguments are extensions with defaults
    Renderer(int width, int height, int device = 0, int rank = 0, int world = 1, int strip_rows = 16)
        : Window(width, height, "Pathtracer"), m_camera(glm::vec3(0.0f, 0.0f, -35.0f), 33.0f)
    {
        std::vector<int> devices;                                               // RTGL_AMD_DEVICES=0,1,2,3: tile the frame across these GPUs
        if (const char *env = std::getenv("RTGL_AMD_DEVICES"))
            for (const char *c = env; *c;) { char *end = nullptr; const long v = std::strtol(c, &end, 10); if (end == c) break; devices.push_back((int)v); c = (*end == ',') ? end + 1 : end; }
        const int rc = (world == 1 && devices.size() > 1) ? rtgl_create_multi(&m_ctx, width, height, devices.data(), (int)devices.size(), 8)
                                                          : rtgl_create_tiled(&m_ctx, width, height, device, rank, world, strip_rows);
        if (rc != RTGL_OK) {
            std::cerr << "rtgl: " << rtgl_last_error(nullptr) << std::endl;     // the reference prints and carries on
            m_ctx = nullptr;
        }
    }
    // one process, several devices: the frame is tiled across `devices` (strips of strip_rows rows) and gathered to devices[0]
    // when it is read or saved; RTGL_AMD_DEVICES=0,1,2,... does the same for an unmodified caller of Renderer(width, height)
    Renderer(int width, int height, const std::vector<int> &devices, int strip_rows = 8)
        : Window(width, height, "Pathtracer"), m_camera(glm::vec3(0.0f, 0.0f, -35.0f), 33.0f)
    {
        if (rtgl_create_multi(&m_ctx, width, height, devices.data(), (int)devices.size(), strip_rows) != RTGL_OK) {
            std::cerr << "rtgl: " << rtgl_last_error(nullptr) << std::endl;
            m_ctx = nullptr;
        }
    }
    ~Renderer() override { rtgl_destroy(m_ctx); }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    void render(float dt) override                                          // src/renderer.cpp:66-149
    {
        (void)dt;
        if (!m_ctx) return;
        rtgl_frame_params p{};
        p.time = m_time;
        p.frames = m_frames;
        p.samples = m_samples;
        p.max_bounce = static_cast<unsigned int>(m_bounces);
        std::memcpy(p.background, &m_background.x, sizeof p.background);
        p.random = rand();
        p.use_envmap = m_envmap ? (m_use_envmap ? 1 : 0) : 0;
        p.use_dof = m_use_dof ? 1 : 0;
        std::memcpy(p.camera_position, &m_camera.position.x, 12);
        p.camera_fov = glm::radians(m_camera.fov);
        p.camera_aperture = m_camera.aperture;
        p.camera_focal_length = m_camera.focal_length;
        std::memcpy(p.camera_forward, &m_camera.forward.x, 12);
        std::memcpy(p.camera_right, &m_camera.right.x, 12);
        std::memcpy(p.camera_up, &m_camera.up.x, 12);
        p.reset_flag = m_reset ? 1 : 0;
        if (m_reset) {                       // the stale frame count was already captured above
            m_reset = false;
            m_time = 0; m_frames = 0;
        }
        m_last_params = p;
        check(rtgl_set_frame_params(m_ctx, &p));
        check(rtgl_render_frame(m_ctx));
    }

    void event(const SDL_Event &event) override                             // src/renderer.cpp:311-392
    {
        switch (event.type) {
        case SDL_MOUSEBUTTONDOWN: if (event.button.button == SDL_BUTTON_LEFT) m_mousedown = true; break;
        case SDL_MOUSEBUTTONUP: if (event.button.button == SDL_BUTTON_LEFT) m_mousedown = false; break;
        case SDL_MOUSEMOTION:
            if (m_mousedown) {
                const float sensitivity = 0.01f;
                m_camera.yaw += static_cast<float>(event.motion.xrel) * sensitivity;
                m_camera.pitch += static_cast<float>(event.motion.yrel) * sensitivity;
                m_camera.forward = vector_from_spherical(m_camera.pitch, m_camera.yaw);
                m_camera.right = glm::normalize(glm::cross(m_camera.forward, glm::vec3(0.0f, 1.0f, 0.0f)));
                m_camera.up = glm::normalize(glm::cross(m_camera.right, m_camera.forward));
                reset_buffer();
            }
            break;
        case SDL_KEYDOWN:
            if (event.key.repeat != 0) return;
            switch (event.key.keysym.sym) {
            case SDLK_SPACE: save_to_file(); break;
            case SDLK_r: reset_buffer(); break;
            case SDLK_j: m_bounces++; reset_buffer(); break;
            case SDLK_k: if (m_bounces > 1) { m_bounces--; reset_buffer(); } break;
            default: break;
            }
            break;
        default: break;
        }
    }

    void keyboard_state(const Uint8 *state) override                        // src/renderer.cpp:394-420
    {
        const float speed = 10.0f * m_clock.delta;
        auto move = [&](int sc, const glm::vec3 &dir, float sign) {
            if (state[sc]) { m_camera.position += dir * (speed * sign); reset_buffer(); }
        };
        move(SDL_SCANCODE_W, m_camera.forward, +1); move(SDL_SCANCODE_S, m_camera.forward, -1);
        move(SDL_SCANCODE_A, m_camera.right, -1);   move(SDL_SCANCODE_D, m_camera.right, +1);
        move(SDL_SCANCODE_E, m_camera.up, +1);      move(SDL_SCANCODE_Q, m_camera.up, -1);
    }

    // ---- scene setters: synchronous copies, like glBufferData (src/renderer.cpp:151-216)
    void set_spheres(const std::vector<Sphere> &spheres) { if (m_ctx) check(rtgl_upload_spheres(m_ctx, spheres.data(), (uint32_t)spheres.size())); }
    void set_materials(const std::vector<Material> &materials) { if (m_ctx) check(rtgl_upload_materials(m_ctx, materials.data(), (uint32_t)materials.size())); }
    void set_envmap(std::unique_ptr<CubemapTexture> envmap)
    {
        m_envmap = std::move(envmap);
        if (m_ctx && m_envmap)
            check(rtgl_upload_envmap(m_ctx, m_envmap->data.data(), m_envmap->faces, std::max(m_envmap->width, 1), std::max(m_envmap->height, 1),
                                     m_envmap->channels == 3 ? 3 : 4));
    }
    void set_vertices(const std::vector<glm::vec4> &vertices)
    {
        auto triangles = to_triangles(vertices);
        if (m_ctx) check(rtgl_upload_vertices(m_ctx, triangles.data(), (uint32_t)triangles.size() * 3));
    }
    void set_meshes(const std::vector<Mesh> &meshes) { if (m_ctx) check(rtgl_upload_meshes(m_ctx, meshes.data(), (uint32_t)meshes.size())); }
    void set_nodes(const std::vector<KdNode> &nodes) { if (m_ctx) check(rtgl_upload_nodes(m_ctx, nodes.data(), (uint32_t)nodes.size())); }
    void set_kdtree(const std::vector<Sphere> &objects)                     // src/renderer.cpp:193-203
    {
        KdTree<Sphere, 1, 2> tree(objects);
        set_spheres(tree.primitives());
        set_nodes(tree.nodes());
        m_use_bvh = true;
    }
    void set_kdtree(const std::vector<glm::vec4> &objects)                  // src/renderer.cpp:205-216, see header note
    {
        std::cerr << "rtgl: set_kdtree(vertices): triangle nodes are not consumed by the path tracer; uploading vertices only" << std::endl;
        set_vertices(objects);
    }

    // triangle soup with w = 1.0 (src/renderer.cpp:255-304); polygons are fanned; failure -> empty vector
    static std::vector<glm::vec4> load_obj(const std::string &path)
    {
        std::ifstream in(path);
        if (!in) { std::cerr << "obj: cannot open " << path << std::endl; return {}; }
        std::vector<glm::vec3> pos;
        std::vector<glm::vec4> out;
        std::string line;
        while (std::getline(in, line)) {
            std::istringstream ls(line);
            std::string tag;
            ls >> tag;
            if (tag == "v") { glm::vec3 p; ls >> p.x >> p.y >> p.z; pos.push_back(p); }
            else if (tag == "f") {
                std::vector<long> idx;
                std::string tok;
                while (ls >> tok) {
                    long i = std::strtol(tok.c_str(), nullptr, 10);          // "v", "v/vt", "v//vn", "v/vt/vn"
                    if (i < 0) i = (long)pos.size() + i + 1;
                    idx.push_back(i);
                }
                for (size_t k = 1; k + 1 < idx.size(); ++k)
                    for (long i : {idx[0], idx[k], idx[k + 1]}) {
                        if (i < 1 || (size_t)i > pos.size()) { std::cerr << "obj: bad index in " << path << std::endl; return {}; }
                        out.emplace_back(pos[(size_t)i - 1], 1.0f);
                    }
            }
        }
        std::printf("# of vertices  = %d\n# of triangles = %d\n", (int)pos.size(), (int)out.size() / 3);
        return out;
    }

    // T * R * S (src/renderer.cpp:247-253)
    static glm::mat4 transform(const glm::vec3 &translate, const glm::vec3 &scale, const glm::quat &rotate = glm::quat(glm::vec3(0.0f)))
    {
        return glm::translate(glm::mat4(1.0f), translate) * glm::mat4(rotate) * glm::scale(glm::mat4(1.0f), scale);
    }

    // ---- what the ImGui panel exposes in the reference (src/renderer.cpp:68-83)
    void reset_buffer() { m_reset = true; }
    void set_bounces(int b) { m_bounces = b; }
    void set_use_envmap(bool v) { m_use_envmap = v; }
    void set_use_dof(bool v) { m_use_dof = v; }
    void set_background(const glm::vec3 &c) { m_background = c; }
    Camera &camera() { return m_camera; }
    rtgl_context *context() { return m_ctx; }
    const rtgl_frame_params &last_frame_params() const { return m_last_params; }   // what the last render() uploaded

    // 8-bit, clamped, vertically flipped PNG named render_<W>x<H>_<unixtime>_<frames>.png (src/renderer.cpp:218-245)
    void save_to_file() const
    {
        if (!m_ctx) return;
        const int rows = rtgl_local_rows(m_ctx);
        std::vector<uint8_t> px((size_t)m_width * rows * 4);
        if (rtgl_read_image_u8(m_ctx, px.data(), 1) != RTGL_OK) { std::cerr << rtgl_last_error(m_ctx) << std::endl; return; }
        const std::time_t ts = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
        const std::string filename = "render_" + std::to_string(m_width) + "x" + std::to_string(m_height) + "_" + std::to_string(ts) + "_" + std::to_string(m_frames) + ".png";
        if (rtgl::write_png(filename, px.data(), m_width, rows, 4)) std::cout << "Wrote render to " << filename << std::endl;
        else std::cerr << "Could not write render to " << filename << std::endl;
    }
    // raw accumulation image, RGBA32F, row 0 = bottom (extension)
    std::vector<float> read_image() const
    {
        std::vector<float> img((size_t)m_width * (m_ctx ? rtgl_local_rows(m_ctx) : 0) * 4);
        if (m_ctx) check(rtgl_read_image_f32(m_ctx, img.data()));
        return img;
    }


    // ---- lossless outputs and a resumable state (the reference itself only writes the 8-bit PNG above).  Like the
    // reference's file operations they print and return false on failure.
    // RGBA32F as it sits in the accumulation image: width * height * 16 bytes, no header, row 0 = bottom
    bool save_raw(const std::string &path) const
    {
        const std::vector<float> img = read_image();
        return write_file(path, "", img.data(), img.size() * sizeof(float));
    }
    // Portable float map: "PF\n<w> <h>\n-1.0\n" (colour, little-endian), RGB float32, bottom row first -- the image's own row order
    bool save_pfm(const std::string &path) const
    {
        const std::vector<float> img = read_image();
        std::vector<float> rgb(img.size() / 4 * 3);
        for (size_t i = 0; i < img.size() / 4; ++i) { rgb[3 * i] = img[4 * i]; rgb[3 * i + 1] = img[4 * i + 1]; rgb[3 * i + 2] = img[4 * i + 2]; }
        const std::string head = "PF\n" + std::to_string(m_width) + " " + std::to_string(img.size() / 4 / (size_t)std::max(m_width, 1)) + "\n-1.0\n";
        return write_file(path, head, rgb.data(), rgb.size() * sizeof(float));
    }
    // The progressive state: the accumulation image and the frame counters the running mean depends on (u_frames, src/renderer.cpp:98).
    // load_state() into a Renderer of the same size continues exactly where save_state() stopped: the next frame is mixed in with
    // weight 1 / (frames + 1) as if the process had never ended.  (u_random continues from the caller's rand() stream, which is not
    // part of the state; scene and camera are the caller's to restore.)
    bool save_state(const std::string &path) const
    {
        const std::vector<float> img = read_image();
        StateHeader h{};
        std::memcpy(h.magic, "RTGLST01", 8);
        h.width = m_width; h.height = (int32_t)(img.size() / 4 / (size_t)std::max(m_width, 1)); h.frames = m_frames; h.time = m_time;
        return write_file(path, std::string(reinterpret_cast<const char *>(&h), sizeof h), img.data(), img.size() * sizeof(float));
    }
    bool load_state(const std::string &path)
    {
        std::ifstream f(path, std::ios::binary);
        StateHeader h{};
        if (!f || !f.read(reinterpret_cast<char *>(&h), sizeof h) || std::memcmp(h.magic, "RTGLST01", 8) != 0) { std::cerr << "Could not read state from " << path << std::endl; return false; }
        if (!m_ctx || h.width != m_width || h.height != rtgl_local_rows(m_ctx)) { std::cerr << "State in " << path << " is " << h.width << "x" << h.height << ", not this renderer's size" << std::endl; return false; }
        std::vector<float> img((size_t)h.width * h.height * 4);
        if (!f.read(reinterpret_cast<char *>(img.data()), (std::streamsize)(img.size() * sizeof(float)))) { std::cerr << "State in " << path << " is truncated" << std::endl; return false; }
        if (rtgl_write_image_f32(m_ctx, img.data()) != RTGL_OK) { std::cerr << "rtgl: " << rtgl_last_error(m_ctx) << std::endl; return false; }
        m_frames = h.frames; m_time = h.time; m_reset = false;
        return true;
    }

private:
    struct StateHeader { char magic[8]; int32_t width, height, frames; float time; };
    static bool write_file(const std::string &path, const std::string &head, const void *data, size_t bytes)
    {
        std::ofstream f(path, std::ios::binary);
        if (f) { f.write(head.data(), (std::streamsize)head.size()); f.write(static_cast<const char *>(data), (std::streamsize)bytes); }
        if (!f) { std::cerr << "Could not write " << path << std::endl; return false; }
        return true;
    }
    void check(int rc) const { if (rc != RTGL_OK) std::cerr << "rtgl: " << rtgl_last_error(m_ctx) << std::endl; }

    rtgl_context *m_ctx = nullptr;
    rtgl_frame_params m_last_params{};
    std::unique_ptr<CubemapTexture> m_envmap = nullptr;
    int m_bounces = 5;
    unsigned int m_samples = 1;
    Camera m_camera;
    bool m_reset = false;
    bool m_mousedown = false;
    bool m_use_envmap = true;
    bool m_use_dof = true;
    bool m_use_bvh = false;
    glm::vec3 m_background = glm::vec3(0.52f, 0.80f, 0.92f);
};
