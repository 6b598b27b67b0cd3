// rt_scene.hpp -- C++17 host side above the C ABI of librt_amd.so, with the reference's vocabulary:
// Camera, Material, MeshTransform, Meshes, Models, Scene, BSPTree, RayTraceData, RenderData, render().
// Header-only; every computation is a call into the library (include/rt_host.h for the steps the
// reference runs on the host before render(), include/rt_amd.h for the GPU path), so a scene prepared
// here is bit-identical to one prepared by the reference's own code.
//
// Differences from the reference that a caller sees:
//   * index types are 32-bit (src/index_types.hpp:8-11 uses 16-bit);
//   * the constants of src/render.hpp:47-49 are run-time RenderOptions;
//   * failures throw std::runtime_error (the reference only asserts);
//   * no GLM dependency: Vec3/Quat are plain structs with GLM's member order.
#ifndef RT_SCENE_HPP
#define RT_SCENE_HPP

#include "rt_amd.h"
#include "rt_host.h"

#include <array>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rt {

struct Vec3 { float x, y, z; };
struct Quat { float w, x, y, z; };                         // glm::quat{w, x, y, z}

inline Quat quatFromEuler(Vec3 euler) {                    // glm::quat(glm::vec3 eulerAngles)
    Quat q;
    rt_quat_from_euler(&euler.x, &q.w);
    return q;
}

struct Camera {                                            // src/camera.hpp:10-31
    Vec3 position;
    Quat orientation;
    float fov;
};

struct Material {                                          // src/material.hpp:12-20
    Vec3 colour;
    float roughness, metalness;
    Vec3 emission{0.0f, 0.0f, 0.0f};
};

struct MeshTransform {                                     // src/mesh.hpp:57-60
    Vec3 position;
    Quat orientation;
    Vec3 scale;
};

struct Mesh {                                              // one entry of Meshes' initializer list, src/scene.cpp:8
    std::vector<Vec3> vertexPositions, vertexNormals;
    std::vector<std::array<std::uint32_t, 3>> tris;        // indices relative to this mesh's vertices
};

// Reads a Wavefront OBJ or PLY file (rt_mesh_load).
inline Mesh loadMesh(std::string const& path) {
    rt_mesh_file* f = nullptr;
    if (rt_mesh_load(path.c_str(), &f) != 0) throw std::runtime_error(rt_mesh_last_error());
    std::uint32_t nv = 0, nt = 0;
    rt_mesh_counts(f, &nv, &nt);
    float const *p = nullptr, *n = nullptr;
    std::uint32_t const* t = nullptr;
    rt_mesh_view(f, &p, &n, &t);
    Mesh m;
    m.vertexPositions.resize(nv); m.vertexNormals.resize(nv); m.tris.resize(nt);
    for (std::uint32_t i = 0; i < nv; ++i) {
        m.vertexPositions[i] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
        m.vertexNormals[i] = {n[3 * i], n[3 * i + 1], n[3 * i + 2]};
    }
    for (std::uint32_t i = 0; i < nt; ++i) m.tris[i] = {t[3 * i], t[3 * i + 1], t[3 * i + 2]};
    rt_mesh_destroy(f);
    return m;
}

struct Range { std::uint32_t begin, size; };               // VertexRange / TriRange, src/mesh.hpp:18-47

struct Meshes {                                            // src/scene.hpp:17-26, src/scene.cpp:8-30
    std::vector<Vec3> vertexPositions, vertexNormals;
    std::vector<std::array<std::uint32_t, 3>> tris;
    std::vector<Range> vertexRanges, triRanges;

    Meshes() = default;
    Meshes(std::initializer_list<Mesh> meshes) { for (auto const& m : meshes) add(m); }
    void add(Mesh const& m) {
        if (m.vertexPositions.size() != m.vertexNormals.size()) throw std::runtime_error("Meshes: positions and normals differ in count");
        for (auto const& t : m.tris)
            for (auto i : t) if (i >= m.vertexPositions.size()) throw std::runtime_error("Meshes: triangle index outside the mesh");
        vertexRanges.push_back({static_cast<std::uint32_t>(vertexPositions.size()), static_cast<std::uint32_t>(m.vertexPositions.size())});
        triRanges.push_back({static_cast<std::uint32_t>(tris.size()), static_cast<std::uint32_t>(m.tris.size())});
        vertexPositions.insert(vertexPositions.end(), m.vertexPositions.begin(), m.vertexPositions.end());
        vertexNormals.insert(vertexNormals.end(), m.vertexNormals.begin(), m.vertexNormals.end());
        tris.insert(tris.end(), m.tris.begin(), m.tris.end());
    }
};

struct Models {                                            // src/scene.hpp:33-37
    std::vector<MeshTransform> meshTransforms;
    std::vector<std::uint32_t> meshes, materials;
    void add(MeshTransform const& t, std::uint32_t mesh, std::uint32_t material) {
        meshTransforms.push_back(t); meshes.push_back(mesh); materials.push_back(material);
    }
};

struct InstantiatedMeshes {                                // src/mesh.hpp:79-83
    std::vector<Vec3> vertexPositions, vertexNormals;
    std::vector<std::uint32_t> vertexBegin;                // per model, plus one trailing entry
};

struct PreprocessedTris {                                  // src/mesh.hpp:141-144; 12 floats per triangle
    std::vector<float> tris;
    std::vector<std::uint32_t> triBegin;                   // per model, plus one trailing entry
};

struct Scene {                                             // src/scene.hpp:29-41
    Camera camera;
    Meshes meshes;
    std::vector<Material> materials;
    Models models;
    std::vector<std::array<float, 12>> preprocessedMaterials;
    InstantiatedMeshes instantiatedMeshes;
    PreprocessedTris preprocessedTris;
    std::array<float, 6> boundingBox{};

    // The sequence of src/main.cpp:191-208.  inflateMode 0 is the reference's `*= 1.001f`.
    void preprocess(int inflateMode = 0) {
        std::uint32_t const modelCount = static_cast<std::uint32_t>(models.meshes.size());
        for (std::uint32_t m = 0; m < modelCount; ++m)
            if (models.meshes[m] >= meshes.vertexRanges.size() || models.materials[m] >= materials.size())
                throw std::runtime_error("Scene: model refers to a missing mesh or material");
        preprocessedMaterials.resize(materials.size());
        for (std::size_t i = 0; i < materials.size(); ++i) {
            auto const& a = materials[i];
            float const in[8] = {a.colour.x, a.colour.y, a.colour.z, a.roughness, a.metalness, a.emission.x, a.emission.y, a.emission.z};
            rt_preprocess_material(in, preprocessedMaterials[i].data());
        }
        std::size_t vertices = 0, triangles = 0;
        for (auto mesh : models.meshes) { vertices += meshes.vertexRanges[mesh].size; triangles += meshes.triRanges[mesh].size; }
        instantiatedMeshes.vertexPositions.resize(vertices);
        instantiatedMeshes.vertexNormals.resize(vertices);
        instantiatedMeshes.vertexBegin.assign(modelCount + 1, 0);
        static_assert(sizeof(MeshTransform) == 40 && sizeof(Range) == 8 && sizeof(Vec3) == 12, "layouts shared with the C ABI");
        rt_instantiate_meshes(&meshes.vertexPositions.data()->x, &meshes.vertexNormals.data()->x,
                              &meshes.vertexRanges.data()->begin, &models.meshTransforms.data()->position.x,
                              models.meshes.data(), modelCount, &instantiatedMeshes.vertexPositions.data()->x,
                              &instantiatedMeshes.vertexNormals.data()->x, instantiatedMeshes.vertexBegin.data());
        preprocessedTris.tris.resize(12 * triangles);
        preprocessedTris.triBegin.assign(modelCount + 1, 0);
        rt_preprocess_tris(&instantiatedMeshes.vertexPositions.data()->x, instantiatedMeshes.vertexBegin.data(),
                           meshes.tris.data()->data(), &meshes.triRanges.data()->begin, models.meshes.data(), modelCount,
                           preprocessedTris.tris.data(), preprocessedTris.triBegin.data());
        rt_scene_bounding_box(&instantiatedMeshes.vertexPositions.data()->x, vertices, inflateMode, boundingBox.data());
    }

    // pixelToRayTransform (src/camera.hpp:33-46), column-major {right, down, o}.
    std::array<float, 9> pixelToRayTransform(unsigned imageWidth, unsigned imageHeight) const {
        Vec3 f, d, r;
        rt_camera_basis(&camera.orientation.w, &f.x, &d.x, &r.x);
        std::array<float, 9> m{};
        rt_pixel_to_ray_transform(&f.x, &d.x, &r.x, camera.fov, imageWidth, imageHeight, m.data());
        return m;
    }
};

class BSPTree {                                            // src/bsp.hpp:32-48: built from a preprocessed Scene
public:
    // buildDevice < 0: built on the host with `threads` workers; otherwise the same tree with the per-level work on that GPU
    explicit BSPTree(Scene const& scene, int threads = 0, std::uint32_t maxDepth = 48, int buildDevice = -1) {
        float const* positions = &scene.instantiatedMeshes.vertexPositions.data()->x;
        std::uint32_t const models = static_cast<std::uint32_t>(scene.models.meshes.size());
        int const rc = buildDevice < 0
            ? rt_bsp_build(positions, scene.instantiatedMeshes.vertexBegin.data(), scene.meshes.tris.data()->data(),
                           &scene.meshes.triRanges.data()->begin, scene.models.meshes.data(), models, scene.boundingBox.data(), 32,
                           maxDepth, threads, &_bsp)
            : rt_bsp_build_device(positions, scene.instantiatedMeshes.vertexBegin.data(), scene.meshes.tris.data()->data(),
                                  &scene.meshes.triRanges.data()->begin, scene.models.meshes.data(), models,
                                  scene.boundingBox.data(), 32, maxDepth, buildDevice, &_bsp);
        if (rc != RT_OK) throw std::runtime_error(std::string("BSPTree: ") + rt_last_error());
    }
    ~BSPTree() { rt_bsp_destroy(_bsp); }
    BSPTree(BSPTree const&) = delete;
    BSPTree& operator=(BSPTree const&) = delete;
    rt_bsp const* handle() const { return _bsp; }
    std::uint32_t depth() const { std::uint32_t c[5]; rt_bsp_counts(_bsp, c); return c[3]; }
    std::uint32_t leafCount() const { std::uint32_t c[5]; rt_bsp_counts(_bsp, c); return c[1]; }
private:
    rt_bsp* _bsp = nullptr;
};

struct RayTraceData {                                      // src/render.hpp:28-35
    BSPTree const& bspTree;
    Scene const& scene;                                    // normals, ranges, triangles, materials of the models
};

struct RenderData {                                        // src/render.hpp:38-44
    unsigned imageWidth, imageHeight;
    Vec3 cameraPosition;
    std::array<float, 9> pixelToRayTransform;
    RayTraceData rayTraceData;
};

struct RenderOptions {                                     // src/render.hpp:47-49 made run-time
    std::uint32_t pixelSampleRate = 2048;
    std::uint32_t rayBounceLimit = 8;
    float rayIntersectionTMin = 1e-3f;
    std::uint64_t seed = std::random_device{}();
    int device = 0;
    std::vector<int> devices;              // several GPUs: the frame's rows are shared out among them (rt_render_multi); empty: `device` alone
    bool russianRoulette = false;          // RT_FLAG_RUSSIAN_ROULETTE
    bool perLobeSampling = false;          // RT_FLAG_LOBE_SAMPLING
    bool boundingVolumeHierarchy = false;  // RT_FLAG_ACCEL_BVH: the alternative accelerator instead of the BSP tree
    bool sortRays = false;                 // RT_FLAG_SORT_RAYS: live paths re-ordered by origin cell and octant per bounce
};

// A scene resident on one GPU: RayTraceData uploaded once, rendered many times.
class DeviceScene {
public:
    DeviceScene(RayTraceData const& rt, int device = 0, bool withHierarchy = false) {
        Scene const& s = rt.scene;
        std::uint32_t const models = static_cast<std::uint32_t>(s.models.meshes.size());
        std::vector<std::uint32_t> triBegin(models);
        std::vector<float> materials(12 * static_cast<std::size_t>(models));
        for (std::uint32_t m = 0; m < models; ++m) {
            triBegin[m] = s.meshes.triRanges[s.models.meshes[m]].begin;
            auto const& pm = s.preprocessedMaterials[s.models.materials[m]];
            std::copy(pm.begin(), pm.end(), materials.begin() + 12 * static_cast<std::size_t>(m));
        }
        rt_scene_desc d{};
        rt_bsp_view(rt.bspTree.handle(), d.root_box, &d.root_index, &d.inodes, &d.leaf_first_ref, &d.leaf_refs);
        std::uint32_t counts[5];
        rt_bsp_counts(rt.bspTree.handle(), counts);
        d.inode_count = counts[0];
        d.leaf_count = counts[1];
        d.pretris = s.preprocessedTris.tris.data();
        d.model_pretri_begin = s.preprocessedTris.triBegin.data();
        d.pretri_count = s.preprocessedTris.triBegin.back();
        d.vertex_normals = &s.instantiatedMeshes.vertexNormals.data()->x;
        d.model_vertex_begin = s.instantiatedMeshes.vertexBegin.data();
        d.vertex_count = s.instantiatedMeshes.vertexBegin.back();
        d.base_tris = s.meshes.tris.data()->data();
        d.model_tri_begin = triBegin.data();
        d.base_tri_count = static_cast<std::uint32_t>(s.meshes.tris.size());
        d.model_materials = materials.data();
        d.model_count = models;
        if (rt_scene_create(&d, device, &_scene) != RT_OK) throw std::runtime_error(rt_last_error());
        // the alternative accelerator is cheap to build (a fraction of the BSP build) and costs a third of the BSP's memory
        if (withHierarchy && rt_scene_build_bvh4(_scene, &d, 4, 0) != RT_OK) {
            std::string const message = rt_last_error();
            rt_scene_destroy(_scene);
            throw std::runtime_error(message);
        }
    }
    ~DeviceScene() { rt_scene_destroy(_scene); }
    DeviceScene(DeviceScene const&) = delete;
    DeviceScene& operator=(DeviceScene const&) = delete;
    rt_scene* handle() const { return _scene; }
private:
    rt_scene* _scene = nullptr;
};

inline rt_render_params renderParams(RenderData const& data, RenderOptions const& o) {
    rt_render_params p{};
    p.width = p.w = data.imageWidth;
    p.height = p.h = data.imageHeight;
    p.spp = o.pixelSampleRate;
    p.bounce_limit = o.rayBounceLimit;
    p.t_min = o.rayIntersectionTMin;
    p.seed = o.seed;
    p.camera_position[0] = data.cameraPosition.x; p.camera_position[1] = data.cameraPosition.y; p.camera_position[2] = data.cameraPosition.z;
    for (int i = 0; i < 9; ++i) p.pixel_to_ray[i] = data.pixelToRayTransform[i];
    p.flags = (o.russianRoulette ? RT_FLAG_RUSSIAN_ROULETTE : 0u) | (o.perLobeSampling ? RT_FLAG_LOBE_SAMPLING : 0u)
              | (o.boundingVolumeHierarchy ? RT_FLAG_ACCEL_BVH : 0u) | (o.sortRays ? RT_FLAG_SORT_RAYS : 0u);
    return p;
}

// render(), src/render.hpp:215-234: image holds imageWidth*imageHeight pixels of linear HDR radiance.
inline void render(DeviceScene& scene, RenderData const& data, Vec3* image, RenderOptions const& options = {}, rt_stats* stats = nullptr) {
    rt_render_params const p = renderParams(data, options);
    if (rt_render(scene.handle(), &p, &image->x, stats) != RT_OK) throw std::runtime_error(rt_last_error());
}

inline void render(RenderData const& data, Vec3* image, RenderOptions const& options = {}, rt_stats* stats = nullptr) {
    if (options.devices.size() > 1) {
        // one copy of the scene per GPU, one frame shared out among them
        std::vector<std::unique_ptr<DeviceScene>> scenes;
        std::vector<rt_scene*> handles;
        for (int device : options.devices) {
            scenes.push_back(std::make_unique<DeviceScene>(data.rayTraceData, device, options.boundingVolumeHierarchy));
            handles.push_back(scenes.back()->handle());
        }
        rt_render_params const p = renderParams(data, options);
        if (rt_render_multi(handles.data(), static_cast<std::uint32_t>(handles.size()), &p, &image->x, stats) != RT_OK)
            throw std::runtime_error(rt_last_error());
        return;
    }
    DeviceScene scene{data.rayTraceData, options.devices.empty() ? options.device : options.devices.front(), options.boundingVolumeHierarchy};
    render(scene, data, image, options, stats);
}

// src/main.cpp:233-240 in one call; the result is the payload of the reference's output.ppm.
inline std::vector<std::uint8_t> postprocess(Vec3 const* hdr, unsigned imageWidth, unsigned imageHeight, int device = 0) {
    std::vector<std::uint8_t> rgb8(static_cast<std::size_t>(imageWidth) * imageHeight * 3);
    if (rt_postprocess(device, &hdr->x, imageWidth, imageHeight, rgb8.data()) != RT_OK) throw std::runtime_error(rt_last_error());
    return rgb8;
}

inline void writePPM(std::string const& path, std::vector<std::uint8_t> const& rgb8, unsigned imageWidth, unsigned imageHeight) {
    if (rt_write_ppm(path.c_str(), rgb8.data(), imageWidth, imageHeight) != 0) throw std::runtime_error("cannot write " + path);
}

}  // namespace rt
#endif
