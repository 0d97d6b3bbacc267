// facade_demo.cpp -- a caller written against the reference's Renderer API (the shape of the reference's
// src/main.cpp: srand(0), construct Renderer, set_* / load_obj / transform, run()) compiled against
// include/rtgl/renderer.h.  Used by tests/test_facade.py: it dumps the vertex buffer it uploaded and the
// final accumulation image so the test can replay the same inputs through the oracle.
//   facade_demo <dir> <width> <height> <frames> <reset_at_frame|0>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rtgl/renderer.h"

static bool dump(const std::string &path, const void *data, size_t bytes)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(data, 1, bytes, f) == bytes;
    std::fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: facade_demo dir width height frames reset_at\n"); return 2; }
    const std::string dir = argv[1];
    const int width = std::atoi(argv[2]), height = std::atoi(argv[3]), frames = std::atoi(argv[4]), reset_at = std::atoi(argv[5]);

    srand(0);
    Renderer renderer(width, height);
    const bool have_device = renderer.context() != nullptr;      // without a device: still dump the marshalled inputs

    const float r = 10000, room = 16.0f, sr = 4.0f;
    const std::vector<Sphere> spheres = {
        Sphere({0.0f, -(room + r), 0.0f}, r, 0),
        Sphere({3.0f, room + 10.0f, 0.0f}, 3.0f, 1),
        Sphere({-18.0f, -room + sr, 0.0f}, sr, 4),
        Sphere({-6.0f, -room + sr, 0.0f}, sr, 5),
        Sphere({+18.0f, -room + sr, 0.0f}, sr, 7),
    };
    const std::vector<Material> materials = {
        Material(gfx::rgb(0xAAAAAA)),
        Material(gfx::rgb(0xFFFFFF), gfx::rgb(0xFFFEFA) * 30.0f),
        Material(gfx::rgb(0xBC0000)),
        Material(gfx::rgb(0x00BC00)),
        Material(gfx::rgb(0xAAAAAA), gfx::rgb(0x0), 1.0f, MaterialType::SPECULAR),
        Material(gfx::rgb(0xFFFFFF), gfx::rgb(0x0), 0.0f, MaterialType::TRANSMISSIVE),
        Material(gfx::rgb(0xFF5733), gfx::rgb(0x0), 0.0f, MaterialType::TRANSMISSIVE),
        Material(gfx::rgb(0xAAAAAA), gfx::rgb(0x0), 0.5f, MaterialType::SPECULAR),
    };
    renderer.set_materials(materials);
    dump(dir + "/materials.raw", materials.data(), materials.size() * sizeof(Material));

    // spheres through the reference's own route for visible spheres: a kd-tree over them
    renderer.set_kdtree(spheres);
    {
        KdTree<Sphere, 1, 2> tree(spheres);
        auto nodes = tree.nodes();
        auto prims = tree.primitives();
        dump(dir + "/nodes.raw", nodes.data(), nodes.size() * sizeof(KdNode));
        dump(dir + "/spheres.raw", prims.data(), prims.size() * sizeof(Sphere));
    }

    std::vector<glm::vec4> obj = Renderer::load_obj(dir + "/mesh.obj");
    glm::mat4 matrix = Renderer::transform(glm::vec3(6.0f, -room + sr, -2.0f), glm::vec3(sr), glm::quat(glm::vec3(0.0f, 0.6f, 0.0f)));
    for (glm::vec4 &vertex : obj) {
        vertex = matrix * vertex;
        vertex.w = 6;
    }
    renderer.set_vertices(obj);
    renderer.set_meshes({Mesh(0, (unsigned int)obj.size() / 3, 6)});
    dump(dir + "/vertices.raw", obj.data(), obj.size() * sizeof(glm::vec4));

    const std::array<std::string, 6> faces = {dir + "/right.png", dir + "/left.png", dir + "/top.png", dir + "/bottom.png", dir + "/front.png", dir + "/back.png"};
    {
        CubemapTexture probe(faces);
        dump(dir + "/envmap.raw", probe.data.data(), probe.data.size());
    }
    renderer.set_envmap(std::make_unique<CubemapTexture>(faces));
    if (!have_device) return 3;

    renderer.set_bounces(6);
    // resume: RTGL_DEMO_LOAD_STATE=<file> continues the accumulation a previous run saved
    if (const char *st = std::getenv("RTGL_DEMO_LOAD_STATE")) if (!renderer.load_state(st)) return 5;
    std::vector<rtgl_frame_params> used;
    for (int f = 1; f <= frames; ++f) {
        if (f == reset_at) {                      // what pressing 'r' does in the reference (src/renderer.cpp:365-367)
            SDL_Event e{};
            e.type = SDL_KEYDOWN;
            e.key.repeat = 0;
            e.key.keysym.sym = SDLK_r;
            renderer.push_event(e);
        }
        renderer.set_frame_budget(1);
        renderer.run();                           // one loop iteration: m_frames++, events, render
        used.push_back(renderer.last_frame_params());
    }
    dump(dir + "/params.raw", used.data(), used.size() * sizeof(rtgl_frame_params));
    std::vector<float> img = renderer.read_image();
    if (!dump(dir + "/image.raw", img.data(), img.size() * sizeof(float))) return 4;
    if (argc > 6) renderer.save_to_file();
    if (const char *st = std::getenv("RTGL_DEMO_SAVE_STATE"))
        if (!renderer.save_state(st) || !renderer.save_pfm(dir + "/image.pfm") || !renderer.save_raw(dir + "/image_rgba32f.raw")) return 6;
    return 0;
}
