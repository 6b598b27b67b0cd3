// TEST INFRASTRUCTURE ONLY -- the drop-in proof.
//
// This translation unit is the reference program itself (src/main.cpp with src/scene.cpp, compiled
// where they lie under /root/reference/src by oracle/build_ref.sh) with exactly one change: the body
// of render() (src/render.hpp:215-234) is the binding to librt_amd.so shown below, the same text a
// maintainer would put into src/render.hpp.  Scene construction, preprocessing, the BSPTree
// constructor, post-processing and the PPM writer all remain the reference's own code.  The result,
// oracle/_ref/ref_main_gpu, runs on the GPU box and writes output.ppm to the current directory.
//
// Two environment variables exist only so that a test can compare its output with the Python path:
// RT_DROPIN_SPP overrides PIXEL_SAMPLE_RATE (src/render.hpp:47) and RT_DROPIN_SEED replaces the
// std::random_device seed.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <sstream>
#include <iomanip>
#include <immintrin.h>

#include "rt_amd.h"

// BSPTree keeps _root/_inodes/_leaves private and has no accessor (src/bsp.hpp:147-172); a maintainer
// would add three accessors instead.
#define private public
#include "bsp.hpp"
#undef private

// The reference's own render() stays in the program under another name.
#define render render_on_cpu
#include "render.hpp"
#undef render

// ---- the binding: replaces src/render.hpp:215-234 -------------------------------------------------
inline void render(RenderData const& data, Span<glm::vec3> image) {
    assert(image.size() == data.imageWidth * data.imageHeight);
    auto const& rt = data.rayTraceData;
    auto const& tree = rt.bspTree;
    auto const models = static_cast<std::uint32_t>(rt.vertexRanges.size());
    static_assert(sizeof(tree._inodes[0]) == sizeof(rt_inode), "INode and rt_inode must have the same layout");

    // Leaves: triangle indices and the preprocessed triangles held in the leaf's 8-wide blocks
    // (Leaf::tris, src/bsp.hpp:161-168; PreprocessedTriBlock, src/geometry.hpp:53-58).
    std::vector<std::uint32_t> leafFirst{0};
    std::vector<rt_tri_ref> refs;
    std::vector<float> leafPretris;
    auto lane = [](FVec8 const& v, unsigned i) { float f[8]; std::memcpy(f, &v, 32); return f[i]; };
    for (auto const& leaf : tree._leaves) {
        for (unsigned i = 0; i < leaf.triCount; ++i) {
            refs.push_back({leaf.triIndices[i].mesh, leaf.triIndices[i].tri});
            auto const& b = leaf.tris[i / 8];
            unsigned const l = i % 8;
            for (FVec3_8 const* v : {&b.normal, &b.v1, &b.v1ToV3, &b.v1ToV2}) {      // rt_scene_desc order
                leafPretris.push_back(lane(v->x, l));
                leafPretris.push_back(lane(v->y, l));
                leafPretris.push_back(lane(v->z, l));
            }
        }
        leafFirst.push_back(static_cast<std::uint32_t>(refs.size()));
    }
    // Per-model offsets with the model->mesh and model->material permutations applied.
    std::vector<std::uint32_t> vertexBegin(models + 1), pretriBegin(models + 1), triBegin(models), baseTris;
    std::vector<float> materials(12 * static_cast<std::size_t>(models));
    std::uint32_t triTotal = 0;
    for (std::uint32_t m = 0; m < models; ++m) {
        vertexBegin[m] = static_cast<std::uint32_t>(rt.vertexRanges[m].begin);
        pretriBegin[m] = triTotal;
        triBegin[m] = static_cast<std::uint32_t>(rt.triRanges[m].begin);
        triTotal += static_cast<std::uint32_t>(rt.triRanges[m].size);
        auto const& mat = rt.materials[m];
        FastFVec3 emission = mat.emission;                       // toGLMVec3 is not const-qualified
        auto const em = emission.toGLMVec3();
        float const v[12] = {mat.ndfAlphaSq, mat.geometryAlphaSq, mat.f0.x, mat.f0.y, mat.f0.z,
                             mat.adjustedColour.x, mat.adjustedColour.y, mat.adjustedColour.z, em.x, em.y, em.z, 0.0f};
        std::copy(v, v + 12, materials.begin() + 12 * static_cast<std::size_t>(m));
    }
    vertexBegin[models] = static_cast<std::uint32_t>(rt.vertexNormals.size());
    pretriBegin[models] = triTotal;
    for (auto const& t : rt.tris) { baseTris.push_back(t.v1); baseTris.push_back(t.v2); baseTris.push_back(t.v3); }

    rt_scene_desc d{};
    std::memcpy(d.root_box, &tree._root.box, 24);
    d.root_index = tree._root.index;
    d.inodes = reinterpret_cast<rt_inode const*>(tree._inodes.data());
    d.inode_count = static_cast<std::uint32_t>(tree._inodes.size());
    d.leaf_first_ref = leafFirst.data();
    d.leaf_refs = refs.data();
    d.leaf_count = static_cast<std::uint32_t>(leafFirst.size() - 1);
    d.leaf_pretris = leafPretris.data();
    d.model_pretri_begin = pretriBegin.data();
    d.pretri_count = triTotal;
    d.vertex_normals = reinterpret_cast<float const*>(rt.vertexNormals.data());
    d.model_vertex_begin = vertexBegin.data();
    d.vertex_count = vertexBegin[models];
    d.base_tris = baseTris.data();
    d.model_tri_begin = triBegin.data();
    d.base_tri_count = static_cast<std::uint32_t>(baseTris.size() / 3);
    d.model_materials = materials.data();
    d.model_count = models;

    rt_scene* gpu = nullptr;
    if (rt_scene_create(&d, /*device*/ 0, &gpu) != RT_OK) throw std::runtime_error(rt_last_error());

    rt_render_params p{};
    p.width = p.w = data.imageWidth;
    p.height = p.h = data.imageHeight;
    p.spp = PIXEL_SAMPLE_RATE;
    p.bounce_limit = RAY_BOUNCE_LIMIT;
    p.t_min = RAY_INTERSECTION_T_MIN;
    p.seed = std::random_device{}();
    if (char const* s = std::getenv("RT_DROPIN_SPP")) p.spp = static_cast<std::uint32_t>(std::strtoul(s, nullptr, 10));
    if (char const* s = std::getenv("RT_DROPIN_SEED")) p.seed = std::strtoull(s, nullptr, 10);
    std::memcpy(p.camera_position, &data.cameraPosition, 12);
    std::memcpy(p.pixel_to_ray, &data.pixelToRayTransform, 36);        // glm::mat3 is column-major
    int const rc = rt_render(gpu, &p, reinterpret_cast<float*>(image.data()), nullptr);
    rt_scene_destroy(gpu);
    if (rc != RT_OK) throw std::runtime_error(rt_last_error());
}
// ---- end of the binding -----------------------------------------------------------------------------

#include "scene.cpp"
#include "main.cpp"
