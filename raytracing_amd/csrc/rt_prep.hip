// Scene preparation on the GPU: instantiateMeshes (src/mesh.hpp:90-138) and preprocessTris
// (src/mesh.hpp:149-188 with preprocessTri, src/geometry.hpp:77-82), one lane per instantiated vertex or
// triangle.  Same operation order as the host versions in rt_host.cpp (glm's scalar code paths, no fused
// multiply-add: this file is compiled with -ffp-contract=off; division and square root are HIP's correctly
// rounded defaults), so the arrays are bit-identical to theirs.
#include "rt_host.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

int rt_internal_fail(int code, const char* message);

namespace {

#define PREP_TRY(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return rt_internal_fail(RT_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

// Device buffers of one call, released on every exit path.
struct DeviceBuffers {
    std::vector<void*> all;
    ~DeviceBuffers() { for (void* p : all) (void)hipFree(p); }
    hipError_t make(void** out, size_t bytes, const void* src) {
        hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        all.push_back(*out);
        if (src && bytes) e = hipMemcpy(*out, src, bytes, hipMemcpyHostToDevice);
        return e;
    }
};

// Per model: columns c0, c1, c2, c3 of MeshTransform::matrix() (src/mesh.hpp:62-69 over glm::mat3_cast,
// libraries/glm/glm/gtc/quaternion.inl:41-66) and the columns n0, n1, n2 of the normal matrix, the inverse
// transpose of the 3x3 part (src/mesh.hpp:74-76; libraries/glm/glm/gtc/matrix_inverse.inl:43-63).  21 floats.
__global__ void k_model_matrices(const float* __restrict__ transforms, uint32_t models, float* __restrict__ out) {
    uint32_t const m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= models) return;
    float const* t = transforms + 10 * static_cast<size_t>(m);
    float const w = t[3], x = t[4], y = t[5], z = t[6];
    float const xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z;
    float const wx = w * x, wy = w * y, wz = w * z;
    float const c0x = (1.0f - 2.0f * (yy + zz)) * t[7], c0y = (2.0f * (xy + wz)) * t[7], c0z = (2.0f * (xz - wy)) * t[7];
    float const c1x = (2.0f * (xy - wz)) * t[8], c1y = (1.0f - 2.0f * (xx + zz)) * t[8], c1z = (2.0f * (yz + wx)) * t[8];
    float const c2x = (2.0f * (xz + wy)) * t[9], c2y = (2.0f * (yz - wx)) * t[9], c2z = (1.0f - 2.0f * (xx + yy)) * t[9];
    float const det = c0x * (c1y * c2z - c1z * c2y) - c0y * (c1x * c2z - c1z * c2x) + c0z * (c1x * c2y - c1y * c2x);
    float* o = out + 21 * static_cast<size_t>(m);
    o[0] = c0x; o[1] = c0y; o[2] = c0z; o[3] = c1x; o[4] = c1y; o[5] = c1z; o[6] = c2x; o[7] = c2y; o[8] = c2z;
    o[9] = t[0]; o[10] = t[1]; o[11] = t[2];
    o[12] = +(c1y * c2z - c2y * c1z) / det; o[13] = -(c1x * c2z - c2x * c1z) / det; o[14] = +(c1x * c2y - c2x * c1y) / det;
    o[15] = -(c0y * c2z - c2y * c0z) / det; o[16] = +(c0x * c2z - c2x * c0z) / det; o[17] = -(c0x * c2y - c2x * c0y) / det;
    o[18] = +(c0y * c1z - c1y * c0z) / det; o[19] = -(c0x * c1z - c1x * c0z) / det; o[20] = +(c0x * c1y - c1x * c0y) / det;
}

// Index of the model whose range [begin[m], begin[m+1]) holds item i.
__device__ __forceinline__ uint32_t modelOf(const uint32_t* __restrict__ begin, uint32_t models, uint32_t i) {
    uint32_t lo = 0, hi = models;            // invariant: begin[lo] <= i < begin[hi]
    while (hi - lo > 1) {
        uint32_t const mid = (lo + hi) / 2;
        if (begin[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// mat4x3 * vec4{p, 1} and normalize(mat3 * n), summed left to right as glm does (src/mesh.hpp:109-131).
__global__ void k_instantiate(const float* __restrict__ positions, const float* __restrict__ normals,
                              const uint32_t* __restrict__ meshVertexRanges, const uint32_t* __restrict__ modelMeshes,
                              const uint32_t* __restrict__ modelVertexBegin, const float* __restrict__ matrices,
                              uint32_t models, uint32_t total, float* __restrict__ outPositions, float* __restrict__ outNormals) {
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint32_t const m = modelOf(modelVertexBegin, models, i);
    uint32_t const src = meshVertexRanges[2 * modelMeshes[m]] + (i - modelVertexBegin[m]);
    float const* c = matrices + 21 * static_cast<size_t>(m);
    float const px = positions[3 * static_cast<size_t>(src)], py = positions[3 * static_cast<size_t>(src) + 1], pz = positions[3 * static_cast<size_t>(src) + 2];
    float const nx = normals[3 * static_cast<size_t>(src)], ny = normals[3 * static_cast<size_t>(src) + 1], nz = normals[3 * static_cast<size_t>(src) + 2];
    outPositions[3 * static_cast<size_t>(i)] = c[0] * px + c[3] * py + c[6] * pz + c[9] * 1.0f;
    outPositions[3 * static_cast<size_t>(i) + 1] = c[1] * px + c[4] * py + c[7] * pz + c[10] * 1.0f;
    outPositions[3 * static_cast<size_t>(i) + 2] = c[2] * px + c[5] * py + c[8] * pz + c[11] * 1.0f;
    float const wx = c[12] * nx + c[15] * ny + c[18] * nz;
    float const wy = c[13] * nx + c[16] * ny + c[19] * nz;
    float const wz = c[14] * nx + c[17] * ny + c[20] * nz;
    float const inv = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);
    outNormals[3 * static_cast<size_t>(i)] = wx * inv;
    outNormals[3 * static_cast<size_t>(i) + 1] = wy * inv;
    outNormals[3 * static_cast<size_t>(i) + 2] = wz * inv;
}

// preprocessTri, src/geometry.hpp:77-82: {normal = cross(v2 - v1, v3 - v1), v1, v3 - v1, v2 - v1}.
__global__ void k_preprocess_tris(const float* __restrict__ instPositions, const uint32_t* __restrict__ modelVertexBegin,
                                  const uint32_t* __restrict__ baseTris, const uint32_t* __restrict__ meshTriRanges,
                                  const uint32_t* __restrict__ modelMeshes, const uint32_t* __restrict__ modelPretriBegin,
                                  uint32_t models, uint32_t total, float* __restrict__ out) {
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint32_t const m = modelOf(modelPretriBegin, models, i);
    uint32_t const* idx = baseTris + 3 * (static_cast<size_t>(meshTriRanges[2 * modelMeshes[m]]) + (i - modelPretriBegin[m]));
    float const* verts = instPositions + 3 * static_cast<size_t>(modelVertexBegin[m]);
    float const* a = verts + 3 * static_cast<size_t>(idx[0]);
    float const* b = verts + 3 * static_cast<size_t>(idx[1]);
    float const* c = verts + 3 * static_cast<size_t>(idx[2]);
    float const e12x = b[0] - a[0], e12y = b[1] - a[1], e12z = b[2] - a[2];
    float const e13x = c[0] - a[0], e13y = c[1] - a[1], e13z = c[2] - a[2];
    float* o = out + 12 * static_cast<size_t>(i);
    o[0] = e12y * e13z - e13y * e12z;          // glm::cross(x, y) = x.y*y.z - y.y*x.z, ...
    o[1] = e12z * e13x - e13z * e12x;
    o[2] = e12x * e13y - e13x * e12y;
    o[3] = a[0]; o[4] = a[1]; o[5] = a[2];
    o[6] = e13x; o[7] = e13y; o[8] = e13z;
    o[9] = e12x; o[10] = e12y; o[11] = e12z;
}

int selectDevice(int device, const char* what) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return rt_internal_fail(RT_ERR_NO_DEVICE, (std::string(what) + ": no HIP device available").c_str());
    if (device < 0 || device >= count) return rt_internal_fail(RT_ERR_INVALID, (std::string(what) + ": device index out of range").c_str());
    PREP_TRY(hipSetDevice(device));
    return RT_OK;
}

// Offsets of the models' items (vertices or triangles) in the instantiated arrays, and what the base arrays
// must hold for them.
int modelOffsets(const uint32_t* meshRanges, uint32_t meshCount, const uint32_t* modelMeshes, uint32_t modelCount,
                 std::vector<uint32_t>& begin, uint64_t& baseExtent, const char* what) {
    begin.assign(static_cast<size_t>(modelCount) + 1, 0u);
    baseExtent = 0;
    for (uint32_t k = 0; k < meshCount; ++k)
        baseExtent = std::max<uint64_t>(baseExtent, static_cast<uint64_t>(meshRanges[2 * k]) + meshRanges[2 * k + 1]);
    uint64_t cursor = 0;
    for (uint32_t m = 0; m < modelCount; ++m) {
        if (modelMeshes[m] >= meshCount) return rt_internal_fail(RT_ERR_INVALID, (std::string(what) + ": model refers to a mesh that does not exist").c_str());
        begin[m] = static_cast<uint32_t>(cursor);
        cursor += meshRanges[2 * modelMeshes[m] + 1];
        if (cursor > 0xFFFFFFF0ull) return rt_internal_fail(RT_ERR_LIMIT, (std::string(what) + ": more than 2^32 instantiated items").c_str());
    }
    begin[modelCount] = static_cast<uint32_t>(cursor);
    return RT_OK;
}

template <class Body>
int guardedPrep(const char* what, Body&& body) {
    try { return body(); }
    catch (std::bad_alloc const&) { return rt_internal_fail(RT_ERR_NOMEM, (std::string(what) + ": out of host memory").c_str()); }
    catch (...) { return rt_internal_fail(RT_ERR_INVALID, (std::string(what) + ": unexpected exception").c_str()); }
}

}  // namespace

extern "C" {

int rt_instantiate_meshes_device(int device, const float* positions, const float* normals, const uint32_t* mesh_vertex_ranges,
                                 uint32_t mesh_count, const float* transforms, const uint32_t* model_meshes,
                                 uint32_t model_count, float* out_positions, float* out_normals,
                                 uint32_t* out_model_vertex_begin) {
    return guardedPrep("rt_instantiate_meshes_device", [&]() -> int {
        if (!out_model_vertex_begin || (model_count && (!positions || !normals || !mesh_vertex_ranges || !transforms || !model_meshes)))
            return rt_internal_fail(RT_ERR_INVALID, "rt_instantiate_meshes_device: null argument");
        std::vector<uint32_t> begin;
        uint64_t baseVertices = 0;
        int rc = modelOffsets(mesh_vertex_ranges, mesh_count, model_meshes, model_count, begin, baseVertices, "rt_instantiate_meshes_device");
        if (rc != RT_OK) return rc;
        std::copy(begin.begin(), begin.end(), out_model_vertex_begin);
        uint32_t const total = begin[model_count];
        if (total == 0) return RT_OK;
        if (!out_positions || !out_normals) return rt_internal_fail(RT_ERR_INVALID, "rt_instantiate_meshes_device: null output");
        rc = selectDevice(device, "rt_instantiate_meshes_device");
        if (rc != RT_OK) return rc;
        DeviceBuffers buf;
        void *dPos, *dNrm, *dRanges, *dTransforms, *dMeshes, *dBegin, *dMatrices, *dOutPos, *dOutNrm;
        PREP_TRY(buf.make(&dPos, 12 * baseVertices, positions));
        PREP_TRY(buf.make(&dNrm, 12 * baseVertices, normals));
        PREP_TRY(buf.make(&dRanges, 8 * static_cast<size_t>(mesh_count), mesh_vertex_ranges));
        PREP_TRY(buf.make(&dTransforms, 40 * static_cast<size_t>(model_count), transforms));
        PREP_TRY(buf.make(&dMeshes, 4 * static_cast<size_t>(model_count), model_meshes));
        PREP_TRY(buf.make(&dBegin, 4 * (static_cast<size_t>(model_count) + 1), begin.data()));
        PREP_TRY(buf.make(&dMatrices, 84 * static_cast<size_t>(model_count), nullptr));
        PREP_TRY(buf.make(&dOutPos, 12 * static_cast<size_t>(total), nullptr));
        PREP_TRY(buf.make(&dOutNrm, 12 * static_cast<size_t>(total), nullptr));
        hipLaunchKernelGGL(k_model_matrices, dim3((model_count + 63) / 64), dim3(64), 0, nullptr, static_cast<const float*>(dTransforms),
                           model_count, static_cast<float*>(dMatrices));
        hipLaunchKernelGGL(k_instantiate, dim3((total + 255) / 256), dim3(256), 0, nullptr, static_cast<const float*>(dPos),
                           static_cast<const float*>(dNrm), static_cast<const uint32_t*>(dRanges), static_cast<const uint32_t*>(dMeshes),
                           static_cast<const uint32_t*>(dBegin), static_cast<const float*>(dMatrices), model_count, total,
                           static_cast<float*>(dOutPos), static_cast<float*>(dOutNrm));
        PREP_TRY(hipGetLastError());
        PREP_TRY(hipMemcpy(out_positions, dOutPos, 12 * static_cast<size_t>(total), hipMemcpyDeviceToHost));
        PREP_TRY(hipMemcpy(out_normals, dOutNrm, 12 * static_cast<size_t>(total), hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

int rt_preprocess_tris_device(int device, const float* inst_positions, const uint32_t* model_vertex_begin,
                              const uint32_t* base_tris, const uint32_t* mesh_tri_ranges, uint32_t mesh_count,
                              const uint32_t* model_meshes, uint32_t model_count, float* out_pretris,
                              uint32_t* out_model_pretri_begin) {
    return guardedPrep("rt_preprocess_tris_device", [&]() -> int {
        if (!out_model_pretri_begin || (model_count && (!inst_positions || !model_vertex_begin || !base_tris || !mesh_tri_ranges || !model_meshes)))
            return rt_internal_fail(RT_ERR_INVALID, "rt_preprocess_tris_device: null argument");
        std::vector<uint32_t> begin;
        uint64_t baseTris = 0;
        int rc = modelOffsets(mesh_tri_ranges, mesh_count, model_meshes, model_count, begin, baseTris, "rt_preprocess_tris_device");
        if (rc != RT_OK) return rc;
        std::copy(begin.begin(), begin.end(), out_model_pretri_begin);
        uint32_t const total = begin[model_count];
        if (total == 0) return RT_OK;
        if (!out_pretris) return rt_internal_fail(RT_ERR_INVALID, "rt_preprocess_tris_device: null output");
        rc = selectDevice(device, "rt_preprocess_tris_device");
        if (rc != RT_OK) return rc;
        DeviceBuffers buf;
        void *dPos, *dVBegin, *dTris, *dRanges, *dMeshes, *dBegin, *dOut;
        PREP_TRY(buf.make(&dPos, 12 * static_cast<size_t>(model_vertex_begin[model_count]), inst_positions));
        PREP_TRY(buf.make(&dVBegin, 4 * (static_cast<size_t>(model_count) + 1), model_vertex_begin));
        PREP_TRY(buf.make(&dTris, 12 * baseTris, base_tris));
        PREP_TRY(buf.make(&dRanges, 8 * static_cast<size_t>(mesh_count), mesh_tri_ranges));
        PREP_TRY(buf.make(&dMeshes, 4 * static_cast<size_t>(model_count), model_meshes));
        PREP_TRY(buf.make(&dBegin, 4 * (static_cast<size_t>(model_count) + 1), begin.data()));
        PREP_TRY(buf.make(&dOut, 48 * static_cast<size_t>(total), nullptr));
        hipLaunchKernelGGL(k_preprocess_tris, dim3((total + 255) / 256), dim3(256), 0, nullptr, static_cast<const float*>(dPos),
                           static_cast<const uint32_t*>(dVBegin), static_cast<const uint32_t*>(dTris), static_cast<const uint32_t*>(dRanges),
                           static_cast<const uint32_t*>(dMeshes), static_cast<const uint32_t*>(dBegin), model_count, total,
                           static_cast<float*>(dOut));
        PREP_TRY(hipGetLastError());
        PREP_TRY(hipMemcpy(out_pretris, dOut, 48 * static_cast<size_t>(total), hipMemcpyDeviceToHost));
        return RT_OK;
    });
}

}  // extern "C"
