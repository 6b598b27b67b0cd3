// The reference program's pipeline (src/main.cpp:124-281) on this repository's C++ host layer
// (include/rt_scene.hpp) and librt_amd.so: build the built-in scene, preprocess, build the tree,
// render on the GPU, post-process on the GPU, write output.ppm, report the same four timings.
//
//   render_builtin [width height spp seed [output.ppm [mesh.obj|mesh.ply]]]
//
// With a mesh file the 27 cubes are replaced by that mesh, scaled into the same place.
#include "rt_scene.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace {

// Two triangles per quad of four consecutive vertices, wound like the reference's quads (src/main.cpp:25-37).
void appendQuads(rt::Mesh& m, unsigned quadCount) {
    for (unsigned q = 0; q < quadCount; ++q) {
        std::uint32_t const b = 4 * q;
        m.tris.push_back({b, b + 2, b + 1});
        m.tris.push_back({b + 1, b + 2, b + 3});
    }
}

rt::Mesh unitPlane() {                                     // y = 0, normal +y (src/main.cpp:40-56)
    rt::Mesh m;
    for (int i = 0; i < 4; ++i) {
        m.vertexPositions.push_back({(i & 1) ? 0.5f : -0.5f, 0.0f, (i & 2) ? 0.5f : -0.5f});
        m.vertexNormals.push_back({0.0f, 1.0f, 0.0f});
    }
    appendQuads(m, 1);
    return m;
}

rt::Mesh unitCube() {                                      // face and corner order of src/main.cpp:59-121
    struct Face { rt::Vec3 n, right, up; };                // corner = centre + (-+)right/2 + (+-)up/2
    Face const faces[6] = {
        {{0, 0, 1}, {1, 0, 0}, {0, 1, 0}},   {{0, 0, -1}, {-1, 0, 0}, {0, 1, 0}}, {{0, 1, 0}, {1, 0, 0}, {0, 0, -1}},
        {{0, -1, 0}, {1, 0, 0}, {0, 0, 1}},  {{-1, 0, 0}, {0, 0, 1}, {0, 1, 0}},  {{1, 0, 0}, {0, 0, -1}, {0, 1, 0}}};
    rt::Mesh m;
    for (auto const& f : faces)
        for (int i = 0; i < 4; ++i) {
            float const sr = (i & 1) ? 0.5f : -0.5f, su = (i & 2) ? -0.5f : 0.5f;
            m.vertexPositions.push_back({0.5f * f.n.x + sr * f.right.x + su * f.up.x, 0.5f * f.n.y + sr * f.right.y + su * f.up.y,
                                         0.5f * f.n.z + sr * f.right.z + su * f.up.z});
            m.vertexNormals.push_back(f.n);
        }
    appendQuads(m, 6);
    return m;
}

double seconds(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
}

}  // namespace

int main(int argcAll, char** argvAll) {
    // "--bvh" anywhere on the command line: render through the alternative accelerator (RT_FLAG_ACCEL_BVH);
    // "--gpu-build": build the BSP tree with rt_bsp_build_device (the same tree);
    // "--devices=0,1,...": share the frame's rows out among these GPUs (rt::RenderOptions::devices, rt_render_multi)
    std::vector<char*> args;
    bool hierarchy = false, gpuBuild = false;
    std::vector<int> devices;
    for (int i = 0; i < argcAll; ++i) {
        std::string const a = argvAll[i];
        if (i > 0 && a == "--bvh") hierarchy = true;
        else if (i > 0 && a == "--gpu-build") gpuBuild = true;
        else if (i > 0 && a.rfind("--devices=", 0) == 0) {
            for (std::size_t at = 10; at < a.size();) {
                devices.push_back(std::atoi(a.c_str() + at));
                std::size_t const comma = a.find(',', at);
                at = comma == std::string::npos ? a.size() : comma + 1;
            }
        }
        else args.push_back(argvAll[i]);
    }
    int const argc = static_cast<int>(args.size());
    char** const argv = args.data();
    unsigned const width = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1920;
    unsigned const height = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 1080;
    rt::RenderOptions options;
    options.boundingVolumeHierarchy = hierarchy;
    options.devices = devices;
    if (!devices.empty()) options.device = devices.front();
    if (argc > 3) options.pixelSampleRate = std::strtoul(argv[3], nullptr, 10);
    if (argc > 4) options.seed = std::strtoull(argv[4], nullptr, 10);
    std::string const output = argc > 5 ? argv[5] : "output.ppm";
    try {
        // Scene of src/main.cpp:134-187.
        float const halfPi = 1.57079632679489661923132169163975144f;
        float const fov = 45.0f * 0.01745329251994329576923690768489f;            // glm::radians(45.0f)
        rt::Scene scene;
        scene.camera = {{9.0f, 8.0f, 16.0f}, rt::quatFromEuler({0.3f, -2.6f, 0.0f}), fov};
        scene.meshes = rt::Meshes{unitPlane(), argc > 6 ? rt::loadMesh(argv[6]) : unitCube()};
        scene.materials = {{{0.25f, 0.25f, 0.25f}, 0.9f, 0.0f}, {{1.0f, 1.0f, 1.0f}, 0.04f, 1.0f}};
        rt::Quat const identity{1.0f, 0.0f, 0.0f, 0.0f};
        scene.models.add({{2.0f, 0.0f, 2.0f}, identity, {16.0f, 1.0f, 16.0f}}, 0, 0);                                  // floor
        scene.models.add({{0.0f, 5.0f, -6.0f}, rt::quatFromEuler({halfPi, 0.0f, 0.0f}), {20.0f, 1.0f, 10.0f}}, 0, 1);    // mirrors
        scene.models.add({{-6.0f, 5.0f, 0.0f}, rt::quatFromEuler({0.0f, 0.0f, -halfPi}), {10.0f, 1.0f, 20.0f}}, 0, 1);
        unsigned const divisor = 3;
        float const scale = 4.0f, span = scale - scale / divisor, cubeScale = 0.75f * scale / divisor;
        for (unsigned x = 0; x < divisor; ++x)
            for (unsigned y = 0; y < divisor; ++y)
                for (unsigned z = 0; z < divisor; ++z) {
                    float const fx = x / static_cast<float>(divisor - 1), fy = y / static_cast<float>(divisor - 1),
                                fz = z / static_cast<float>(divisor - 1);
                    rt::Vec3 const colour{rt_srgb_to_linear(fx), rt_srgb_to_linear(fy), rt_srgb_to_linear(fz)};
                    scene.materials.push_back({colour, 0.5f, 0.5f, colour});
                    rt::Vec3 const position{(fx - 0.5f) * span + 0.0f, (fy - 0.5f) * span + 2.5f, (fz - 0.5f) * span + 0.0f};
                    scene.models.add({position, identity, {cubeScale, cubeScale, cubeScale}}, 1,
                                     static_cast<std::uint32_t>(scene.materials.size() - 1));
                }

        auto const t0 = std::chrono::steady_clock::now();
        scene.preprocess();
        rt::BSPTree const bspTree{scene, 0, 48, gpuBuild ? 0 : -1};
        auto const t1 = std::chrono::steady_clock::now();
        std::vector<rt::Vec3> renderBuffer(static_cast<std::size_t>(width) * height);
        rt::RenderData const renderData{width, height, scene.camera.position, scene.pixelToRayTransform(width, height), {bspTree, scene}};
        rt_stats stats{};
        rt::render(renderData, renderBuffer.data(), options, &stats);
        auto const t2 = std::chrono::steady_clock::now();
        auto const image = rt::postprocess(renderBuffer.data(), width, height, options.device);
        auto const t3 = std::chrono::steady_clock::now();
        rt::writePPM(output, image, width, height);

        std::printf("Preprocess done in %.3f s (%zu models, %u leaves, depth %u)\n", seconds(t0, t1), scene.models.meshes.size(),
                    bspTree.leafCount(), bspTree.depth());
        std::printf("Render done in %.3f s (%.1f Mrays/s, %llu path segments)\n", seconds(t1, t2),
                    stats.segments / seconds(t1, t2) / 1e6, static_cast<unsigned long long>(stats.segments));
        std::printf("Postprocess done in %.3f s\n", seconds(t2, t3));
        std::printf("Pipeline done in %.3f s\n", seconds(t0, t3));
    }
    catch (std::exception const& e) {
        std::fprintf(stderr, "render_builtin: %s\n", e.what());
        return 1;
    }
    return 0;
}
