// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// C-ABI harness around the *real* reference sources, compiled where they lie under
// /root/reference/src by oracle/build_ref.sh into oracle/_ref/libref.so.  The library only exists in
// the development container (the reference never travels to the GPU box); it is used to
//   (1) validate the plain-C restatement in oracle/rt_oracle.c, and
//   (2) generate the committed golden vectors under tests/golden/ (oracle/make_golden.py).
//
// Everything computational below is a call into the reference's own code:
//   preprocessMaterial            src/material.hpp:32
//   pixelToRayTransform           src/camera.hpp:33
//   instantiateMeshes             src/mesh.hpp:90
//   preprocessTris                src/mesh.hpp:149
//   computeBoundingBox            src/geometry.hpp:85
//   BSPTree ctor / intersector    src/bsp.hpp:35, :50
//   lineIntersectsBox             src/geometry.hpp:135
//   triIntersectsBox              src/geometry.hpp:217
//   rayTrace                      src/render.hpp:53
// The harness adds only argument marshalling, the scene *data* of src/main.cpp:134-187, and a
// per-pixel sample loop with a runtime sample count and explicit FastRNG seeds (the reference's
// own loop, src/render.hpp:219-232, has a compile-time count and seeds from std::random_device).

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <execution>
#include <fstream>
#include <initializer_list>
#include <iostream>
#include <iterator>
#include <limits>
#include <memory>
#include <optional>
#include <random>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>
#include <chrono>
#include <string>
#include <sstream>
#include <iomanip>
#include <immintrin.h>

// The tree arrays of BSPTree are private and have no accessor (src/bsp.hpp:147-172).
#define private public
#include "bsp.hpp"
#undef private
#include "scene.cpp"
// plane(), cube() and quadMeshTris() live in main.cpp next to main(); rename main() away.
#define main reference_main_unused
#include "main.cpp"
#undef main

namespace {

struct RefScene {
    Scene scene;
    glm::mat3 pixelToRay;
    BoundingBox rootBox;
    std::unique_ptr<BSPTree> bsp;
    unsigned width, height;

    RefScene(Scene s) : scene{std::move(s)} {}

    void finish(unsigned w, unsigned h, int inflateMode) {
        width = w;
        height = h;
        // Same sequence as src/main.cpp:191-216.
        scene.preprocessedMaterials.resize(scene.materials.size());
        std::transform(scene.materials.cbegin(), scene.materials.cend(), scene.preprocessedMaterials.begin(),
                       preprocessMaterial);
        pixelToRay = ::pixelToRayTransform(scene.camera.forward(), scene.camera.down(), scene.camera.right(),
                                           scene.camera.fov, w, h);
        scene.instantiatedMeshes = instantiateMeshes(
            readOnlySpan(scene.meshes.vertexPositions), readOnlySpan(scene.meshes.vertexNormals),
            readOnlySpan(scene.meshes.vertexRanges), readOnlySpan(scene.models.meshTransforms),
            readOnlySpan(scene.models.meshes));
        scene.preprocessedTris = preprocessTris(
            readOnlySpan(scene.instantiatedMeshes.vertexPositions), readOnlySpan(scene.instantiatedMeshes.vertexRanges),
            readOnlySpan(scene.meshes.tris),
            PermutedSpan{readOnlySpan(scene.meshes.triRanges), readOnlySpan(scene.models.meshes)});
        rootBox = computeBoundingBox(readOnlySpan(scene.instantiatedMeshes.vertexPositions));
        if (inflateMode == 0) {
            // src/main.cpp:207-208 (multiplicative).
            rootBox.min *= 1.001f;
            rootBox.max *= 1.001f;
        }
        else {
            // Additive inflation for scenes whose boxes do not straddle the origin.
            auto const ext = rootBox.max - rootBox.min;
            auto const pad = 0.001f * std::max(ext.x, std::max(ext.y, ext.z));
            rootBox.min -= glm::vec3{pad};
            rootBox.max += glm::vec3{pad};
        }
        bsp = std::make_unique<BSPTree>(
            readOnlySpan(scene.instantiatedMeshes.vertexPositions), readOnlySpan(scene.instantiatedMeshes.vertexRanges),
            readOnlySpan(scene.meshes.tris),
            PermutedSpan{readOnlySpan(scene.meshes.triRanges), readOnlySpan(scene.models.meshes)},
            readOnlySpan(scene.preprocessedTris.tris), readOnlySpan(scene.preprocessedTris.triRanges), rootBox);
    }

    RayTraceData rayTraceData() const {
        return {
            *bsp,
            readOnlySpan(scene.instantiatedMeshes.vertexNormals), readOnlySpan(scene.instantiatedMeshes.vertexRanges),
            readOnlySpan(scene.meshes.tris),
            PermutedSpan{readOnlySpan(scene.meshes.triRanges), readOnlySpan(scene.models.meshes)},
            PermutedSpan{readOnlySpan(scene.preprocessedMaterials), readOnlySpan(scene.models.materials)}};
    }
};

}  // namespace

extern "C" {

// Scene of src/main.cpp:134-187 with a runtime DIVISOR (3 in the reference) and image size.
void* ref_scene_builtin(unsigned divisor, unsigned width, unsigned height) {
    Scene scene{
        {{9.0f, 8.0f, 16.0f}, glm::vec3{0.3f, -2.6f, 0.0f}, glm::radians(45.0f)},
        {plane(), cube()},
        {{{0.25f, 0.25f, 0.25f}, 0.9f, 0.0f, {0.0f, 0.0f, 0.0f}}, {{1.0f, 1.0f, 1.0f}, 0.04f, 1.0f, {0.0f, 0.0f, 0.0f}}},
        {{{{2.0f, 0.0f, 2.0f}, {1.0f, 0.0f, 0.0f, 0.0f}, {16.0f, 1.0f, 16.0f}},
          {{0.0f, 5.0f, -6.0f}, glm::vec3{glm::half_pi<float>(), 0.0f, 0.0f}, {20.0f, 1.0f, 10.0f}},
          {{-6.0f, 5.0f, 0.0f}, glm::vec3{0.0f, 0.0f, -glm::half_pi<float>()}, {10.0f, 1.0f, 20.0f}}},
         {0, 0, 0},
         {0, 1, 1}},
        {},
        {}};
    glm::vec3 const POSITION{0.0f, 2.5f, 0.0f};
    float const SCALE = 4.0f;
    float const SUBSCALE = 0.75f;
    for (unsigned x = 0; x < divisor; ++x) {
        auto const xFract = x / static_cast<float>(divisor - 1);
        for (unsigned y = 0; y < divisor; ++y) {
            auto const yFract = y / static_cast<float>(divisor - 1);
            for (unsigned z = 0; z < divisor; ++z) {
                auto const zFract = z / static_cast<float>(divisor - 1);
                auto const colour = srgbToLinear(glm::vec3{xFract, yFract, zFract});
                scene.materials.push_back({colour, 0.5f, 0.5f, colour});
                auto const position = glm::vec3{(xFract - 0.5f) * (SCALE - SCALE / divisor),
                                                (yFract - 0.5f) * (SCALE - SCALE / divisor),
                                                (zFract - 0.5f) * (SCALE - SCALE / divisor)} + POSITION;
                glm::quat const orientation{1.0f, 0.0f, 0.0f, 0.0f};
                glm::vec3 const scale{SUBSCALE * SCALE / divisor};
                scene.models.meshTransforms.push_back({position, orientation, scale});
                scene.models.meshes.push_back(1);
                scene.models.materials.push_back(intCast<MaterialIndex>(scene.materials.size() - 1));
            }
        }
    }
    auto* rs = new RefScene{std::move(scene)};
    rs->finish(width, height, 0);
    return rs;
}

// General scene from flat arrays.  Index widths follow src/index_types.hpp:8-11 (16 bit).
//   meshVertexRanges / meshTriRanges: {begin,size} pairs per base mesh
//   materials: 8 floats each {colour[3], roughness, metalness, emission[3]}   (src/material.hpp:14-19)
//   transforms: 10 floats each {position[3], quat w,x,y,z, scale[3]}          (src/mesh.hpp:57-60)
//   camera: 8 floats {position[3], quat w,x,y,z, fov}                         (src/camera.hpp:13-16)
void* ref_scene_create(float const* positions, float const* normals, unsigned vertexCount,
                       std::uint16_t const* tris, unsigned triCount,
                       std::uint32_t const* meshVertexRanges, std::uint32_t const* meshTriRanges, unsigned meshCount,
                       float const* materials, unsigned materialCount,
                       float const* transforms, std::uint16_t const* modelMeshes,
                       std::uint16_t const* modelMaterials, unsigned modelCount,
                       float const* camera, unsigned width, unsigned height, int inflateMode) {
    Scene scene{{{camera[0], camera[1], camera[2]}, glm::quat{camera[3], camera[4], camera[5], camera[6]}, camera[7]},
                {}, {}, {}, {}, {}};
    for (unsigned i = 0; i < vertexCount; ++i) {
        scene.meshes.vertexPositions.push_back({positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]});
        scene.meshes.vertexNormals.push_back({normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]});
    }
    for (unsigned i = 0; i < triCount; ++i) {
        scene.meshes.tris.push_back({tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]});
    }
    for (unsigned i = 0; i < meshCount; ++i) {
        scene.meshes.vertexRanges.push_back({meshVertexRanges[2 * i], meshVertexRanges[2 * i + 1]});
        scene.meshes.triRanges.push_back({meshTriRanges[2 * i], meshTriRanges[2 * i + 1]});
    }
    for (unsigned i = 0; i < materialCount; ++i) {
        float const* m = materials + 8 * i;
        scene.materials.push_back({{m[0], m[1], m[2]}, m[3], m[4], {m[5], m[6], m[7]}});
    }
    for (unsigned i = 0; i < modelCount; ++i) {
        float const* t = transforms + 10 * i;
        scene.models.meshTransforms.push_back({{t[0], t[1], t[2]}, glm::quat{t[3], t[4], t[5], t[6]}, {t[7], t[8], t[9]}});
        scene.models.meshes.push_back(modelMeshes[i]);
        scene.models.materials.push_back(modelMaterials[i]);
    }
    auto* rs = new RefScene{std::move(scene)};
    rs->finish(width, height, inflateMode);
    return rs;
}

void ref_scene_destroy(void* h) {
    delete static_cast<RefScene*>(h);
}

// counts: [0] models [1] instantiated vertices [2] preprocessed tris [3] inodes [4] leaves
//         [5] base vertices [6] base tris [7] base meshes [8] materials
void ref_scene_counts(void* h, std::uint32_t* counts) {
    auto const& rs = *static_cast<RefScene*>(h);
    counts[0] = static_cast<std::uint32_t>(rs.scene.models.meshes.size());
    counts[1] = static_cast<std::uint32_t>(rs.scene.instantiatedMeshes.vertexPositions.size());
    counts[2] = static_cast<std::uint32_t>(rs.scene.preprocessedTris.tris.size());
    counts[3] = static_cast<std::uint32_t>(rs.bsp->_inodes.size());
    counts[4] = static_cast<std::uint32_t>(rs.bsp->_leaves.size());
    counts[5] = static_cast<std::uint32_t>(rs.scene.meshes.vertexPositions.size());
    counts[6] = static_cast<std::uint32_t>(rs.scene.meshes.tris.size());
    counts[7] = static_cast<std::uint32_t>(rs.scene.meshes.vertexRanges.size());
    counts[8] = static_cast<std::uint32_t>(rs.scene.materials.size());
}

// Base (un-instantiated) scene description, so a test can feed the very same inputs elsewhere.
void ref_get_base(void* h, float* positions, float* normals, std::uint32_t* tris, std::uint32_t* meshVertexRanges,
                  std::uint32_t* meshTriRanges, float* materials, float* transforms, std::uint32_t* modelMeshes,
                  std::uint32_t* modelMaterials, float* camera) {
    auto const& s = static_cast<RefScene*>(h)->scene;
    for (std::size_t i = 0; i < s.meshes.vertexPositions.size(); ++i) {
        std::memcpy(positions + 3 * i, &s.meshes.vertexPositions[i], 12);
        std::memcpy(normals + 3 * i, &s.meshes.vertexNormals[i], 12);
    }
    for (std::size_t i = 0; i < s.meshes.tris.size(); ++i) {
        tris[3 * i] = s.meshes.tris[i].v1;
        tris[3 * i + 1] = s.meshes.tris[i].v2;
        tris[3 * i + 2] = s.meshes.tris[i].v3;
    }
    for (std::size_t i = 0; i < s.meshes.vertexRanges.size(); ++i) {
        meshVertexRanges[2 * i] = s.meshes.vertexRanges[i].begin;
        meshVertexRanges[2 * i + 1] = s.meshes.vertexRanges[i].size;
        meshTriRanges[2 * i] = s.meshes.triRanges[i].begin;
        meshTriRanges[2 * i + 1] = s.meshes.triRanges[i].size;
    }
    for (std::size_t i = 0; i < s.materials.size(); ++i) {
        auto const& m = s.materials[i];
        float const v[8] = {m.colour.x, m.colour.y, m.colour.z, m.roughness, m.metalness,
                            m.emission.x, m.emission.y, m.emission.z};
        std::memcpy(materials + 8 * i, v, sizeof v);
    }
    for (std::size_t i = 0; i < s.models.meshes.size(); ++i) {
        auto const& t = s.models.meshTransforms[i];
        float const v[10] = {t.position.x, t.position.y, t.position.z, t.orientation.w, t.orientation.x,
                             t.orientation.y, t.orientation.z, t.scale.x, t.scale.y, t.scale.z};
        std::memcpy(transforms + 10 * i, v, sizeof v);
        modelMeshes[i] = s.models.meshes[i];
        modelMaterials[i] = s.models.materials[i];
    }
    auto const& c = s.camera;
    float const cv[8] = {c.position.x, c.position.y, c.position.z, c.orientation.w, c.orientation.x,
                         c.orientation.y, c.orientation.z, c.fov};
    std::memcpy(camera, cv, sizeof cv);
}

// Preprocessed scene: instantiated positions/normals (3 floats each), per-model vertex ranges {begin,size},
// preprocessed tris (12 floats each in struct order normal,v1,v1ToV3,v1ToV2), per-model tri ranges,
// per-model preprocessed material (12 floats: ndfAlphaSq, geometryAlphaSq, f0[3], adjustedColour[3],
// emission[3], 0), camera position (3), pixelToRay (9, column-major), root box (6).
void ref_get_preprocessed(void* h, float* positions, float* normals, std::uint32_t* vertexRanges, float* pretris,
                          std::uint32_t* triRanges, float* modelMaterials, float* cameraPosition, float* pixelToRay,
                          float* rootBox) {
    auto const& rs = *static_cast<RefScene*>(h);
    auto const& s = rs.scene;
    auto const nv = s.instantiatedMeshes.vertexPositions.size();
    std::memcpy(positions, s.instantiatedMeshes.vertexPositions.data(), nv * 12);
    std::memcpy(normals, s.instantiatedMeshes.vertexNormals.data(), nv * 12);
    for (std::size_t i = 0; i < s.models.meshes.size(); ++i) {
        vertexRanges[2 * i] = s.instantiatedMeshes.vertexRanges[i].begin;
        vertexRanges[2 * i + 1] = s.instantiatedMeshes.vertexRanges[i].size;
        triRanges[2 * i] = s.preprocessedTris.triRanges[i].begin;
        triRanges[2 * i + 1] = s.preprocessedTris.triRanges[i].size;
        auto const& m = s.preprocessedMaterials[s.models.materials[i]];
        float const v[12] = {m.ndfAlphaSq, m.geometryAlphaSq, m.f0.x, m.f0.y, m.f0.z,
                             m.adjustedColour.x, m.adjustedColour.y, m.adjustedColour.z,
                             m.emission[0], m.emission[1], m.emission[2], 0.0f};
        std::memcpy(modelMaterials + 12 * i, v, sizeof v);
    }
    static_assert(sizeof(PreprocessedTri) == 48);
    std::memcpy(pretris, s.preprocessedTris.tris.data(), s.preprocessedTris.tris.size() * 48);
    std::memcpy(cameraPosition, &s.camera.position, 12);
    std::memcpy(pixelToRay, &rs.pixelToRay, 36);
    std::memcpy(rootBox, &rs.rootBox, 24);
}

// Tree dump.  root: 6 floats box + index.  inodes: per inode 12 floats {neg box, pos box} and 3 ints
// {neg index, pos index, axis}.  leaves: triCount per leaf, and 32 {mesh,tri} pairs per leaf.
void ref_get_bsp(void* h, float* rootBox, std::int32_t* rootIndex, float* inodeBoxes, std::int32_t* inodeInts,
                 std::uint32_t* leafCounts, std::uint32_t* leafRefs) {
    auto const& bsp = *static_cast<RefScene*>(h)->bsp;
    std::memcpy(rootBox, &bsp._root.box, 24);
    *rootIndex = bsp._root.index;
    for (std::size_t i = 0; i < bsp._inodes.size(); ++i) {
        auto const& n = bsp._inodes[i];
        std::memcpy(inodeBoxes + 12 * i, &n.negativeChild.box, 24);
        std::memcpy(inodeBoxes + 12 * i + 6, &n.positiveChild.box, 24);
        inodeInts[3 * i] = n.negativeChild.index;
        inodeInts[3 * i + 1] = n.positiveChild.index;
        inodeInts[3 * i + 2] = n.divisionAxis;
    }
    for (std::size_t i = 0; i < bsp._leaves.size(); ++i) {
        auto const& l = bsp._leaves[i];
        leafCounts[i] = l.triCount;
        for (unsigned j = 0; j < 32; ++j) {
            leafRefs[(i * 32 + j) * 2] = j < l.triCount ? l.triIndices[j].mesh : 0xFFFFFFFFu;
            leafRefs[(i * 32 + j) * 2 + 1] = j < l.triCount ? l.triIndices[j].tri : 0xFFFFFFFFu;
        }
    }
}

// Nearest intersection for n rays.  out per ray: t,u,v,point[3] (6 floats) and mesh,tri (2 uint32;
// 0xFFFFFFFF on a miss).  src/bsp.hpp:50-145.
void ref_intersect(void* h, float const* origins, float const* directions, std::size_t n, float tMin, int frontOnly,
                   float* outFloats, std::uint32_t* outIds) {
    auto const& bsp = *static_cast<RefScene*>(h)->bsp;
    for (std::size_t i = 0; i < n; ++i) {
        Line const line{{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]},
                        {directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]}};
        auto const r = frontOnly ? bsp.lineTriNearestIntersection<SurfaceConsideration::FRONT_ONLY>(line, tMin)
                                 : bsp.lineTriNearestIntersection<SurfaceConsideration::ALL>(line, tMin);
        float* f = outFloats + 6 * i;
        if (r) {
            f[0] = r->t; f[1] = r->pointCoord2; f[2] = r->pointCoord3;
            f[3] = r->point.x; f[4] = r->point.y; f[5] = r->point.z;
            outIds[2 * i] = r->meshTriIndex.mesh;
            outIds[2 * i + 1] = r->meshTriIndex.tri;
        }
        else {
            for (int k = 0; k < 6; ++k) f[k] = 0.0f;
            outIds[2 * i] = outIds[2 * i + 1] = 0xFFFFFFFFu;
        }
    }
}

// Camera ray through (sampleX, sampleY) exactly as src/render.hpp:227.
void ref_camera_rays(void* h, float const* sampleXY, std::size_t n, float* directions) {
    auto const& rs = *static_cast<RefScene*>(h);
    for (std::size_t i = 0; i < n; ++i) {
        auto const d = glm::normalize(rs.pixelToRay * glm::vec3{sampleXY[2 * i], sampleXY[2 * i + 1], 1.0f});
        std::memcpy(directions + 3 * i, &d, 12);
    }
}

// src/geometry.hpp:135.  boxes: 6 floats each.
void ref_line_intersects_box(float const* origins, float const* directions, float const* boxes, std::size_t n,
                             std::uint8_t* out) {
    for (std::size_t i = 0; i < n; ++i) {
        Line const line{{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]},
                        {directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]}};
        BoundingBox const box{{boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]},
                              {boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]}};
        out[i] = lineIntersectsBox(line, box) ? 1 : 0;
    }
}

// src/geometry.hpp:217.  tris: 9 floats each.
void ref_tri_intersects_box(float const* tris, float const* boxes, std::size_t n, std::uint8_t* out) {
    for (std::size_t i = 0; i < n; ++i) {
        float const* t = tris + 9 * i;
        Tri const tri{{t[0], t[1], t[2]}, {t[3], t[4], t[5]}, {t[6], t[7], t[8]}};
        BoundingBox const box{{boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]},
                              {boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]}};
        out[i] = triIntersectsBox(tri, box) ? 1 : 0;
    }
}

// src/render.hpp:53 for n rays, each with its own FastRNG seed.  out: 3 floats per ray.
void ref_ray_trace(void* h, float const* origins, float const* directions, std::uint64_t const* seeds, std::size_t n,
                   float* out) {
    auto const data = static_cast<RefScene*>(h)->rayTraceData();
    for (std::size_t i = 0; i < n; ++i) {
        Line const line{{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]},
                        {directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]}};
        FastRNG rng{seeds[i]};
        auto c = rayTrace(data, line, rng);
        out[3 * i] = c[0];
        out[3 * i + 1] = c[1];
        out[3 * i + 2] = c[2];
    }
}

// The per-pixel body of src/render.hpp:219-232 with a runtime sample count; pixel p uses
// FastRNG{seed + p} so the image is reproducible and independent of thread scheduling.
void ref_render(void* h, unsigned spp, std::uint64_t seed, float* image) {
    auto const& rs = *static_cast<RefScene*>(h);
    auto const data = rs.rayTraceData();
    std::size_t const pixels = static_cast<std::size_t>(rs.width) * rs.height;
#pragma omp parallel for schedule(dynamic, 64)
    for (std::size_t index = 0; index < pixels; ++index) {
        auto const pixelX = index % rs.width;
        auto const pixelY = index / rs.width;
        FastRNG rng{seed + index};
        FastFVec3 colour{0.0f, 0.0f, 0.0f};
        for (unsigned i = 0; i < spp; ++i) {
            auto const sampleX = pixelX + rng.unitFloatOpen();
            auto const sampleY = pixelY + rng.unitFloatOpen();
            auto const rayDirection = glm::normalize(rs.pixelToRay * glm::vec3{sampleX, sampleY, 1.0f});
            Line const ray{rs.scene.camera.position, rayDirection};
            colour += rayTrace(data, ray, rng);
        }
        colour /= spp;
        auto const c = colour.toGLMVec3();
        std::memcpy(image + 3 * index, &c, 12);
    }
}

// Post-processing exactly as src/main.cpp:233-240 applies it.
void ref_tone_srgb_flags(const float* hdr, std::size_t pixels, float* out) {
    std::vector<glm::vec3> buf(pixels);
    std::memcpy(buf.data(), hdr, pixels * 12);
    std::transform(buf.cbegin(), buf.cend(), buf.begin(), reinhardToneMap);
    std::transform(buf.cbegin(), buf.cend(), buf.begin(), static_cast<glm::vec3(*)(glm::vec3)>(linearToSRGB));
    std::transform(buf.cbegin(), buf.cend(), buf.begin(), nanToRed);
    std::transform(buf.cbegin(), buf.cend(), buf.begin(), infToGreen);
    std::memcpy(out, buf.data(), pixels * 12);
}

void ref_median3(const float* image, unsigned width, unsigned height, float* out) {
    std::size_t const pixels = static_cast<std::size_t>(width) * height;
    std::vector<glm::vec3> in(pixels), res(pixels);
    std::memcpy(in.data(), image, pixels * 12);
    std::copy(in.cbegin(), in.cend(), res.begin());
    medianFilter<1>(readOnlySpan(in), width, Span{res});
    std::memcpy(out, res.data(), pixels * 12);
}

void ref_to_u8(const float* image, std::size_t pixels, std::uint8_t* out) {
    for (std::size_t i = 0; i < pixels; ++i) {
        glm::vec3 p;
        std::memcpy(&p, image + 3 * i, 12);
        glm::u8vec3 const q = floatTo8BitUInt(p);
        out[3 * i] = q.x; out[3 * i + 1] = q.y; out[3 * i + 2] = q.z;
    }
}

}  // extern "C"
