// Re-ordering of a bounce's ray queue (RT_FLAG_SORT_RAYS): the slots of the live paths are sorted by a key made of the
// cell of the ray's origin (a 32 x 32 x 32 grid over the scene's root box, Morton order) and the octant of its
// direction, so that the lanes of a wave -- which take consecutive queue entries -- start in the same part of the tree
// and head the same way.  Only the ORDER in which paths are traced changes: every path keeps its slot, its samples and
// its arithmetic, so the image is the same bit for bit.  No counterpart in the reference (one path per CPU thread).
#include "rt_amd.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <string>

int rt_internal_fail(int code, const char* message);

namespace {

__device__ __forceinline__ uint32_t spread5(uint32_t v) {        // bits 0-4 of v to bits 0, 3, 6, 9, 12
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}

__global__ void k_sort_keys(const float4* __restrict__ A, const float4* __restrict__ B, const uint32_t* __restrict__ queue,
                            uint32_t n, float lox, float loy, float loz, float sx, float sy, float sz,
                            uint32_t* __restrict__ keys) {
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t const slot = queue[i];
    float4 const a = A[slot], b = B[slot];
    // NaN or out-of-range coordinates land in cell 0 or 31: any key is a valid key
    float const fx = fminf(fmaxf((a.x - lox) * sx, 0.0f), 31.0f), fy = fminf(fmaxf((a.y - loy) * sy, 0.0f), 31.0f),
                fz = fminf(fmaxf((a.z - loz) * sz, 0.0f), 31.0f);
    uint32_t const cell = spread5(static_cast<uint32_t>(fx)) | (spread5(static_cast<uint32_t>(fy)) << 1)
                        | (spread5(static_cast<uint32_t>(fz)) << 2);
    uint32_t const octant = (a.w < 0.0f ? 1u : 0u) | (b.x < 0.0f ? 2u : 0u) | (b.y < 0.0f ? 4u : 0u);
    keys[i] = (cell << 3) | octant;
}

}  // namespace

// Bytes of temporary storage the sort of up to `maxN` entries needs.
int rt_internal_sort_temp_bytes(uint32_t maxN, size_t* bytes) {
    *bytes = 0;
    hipError_t const e = hipcub::DeviceRadixSort::SortPairs(nullptr, *bytes, static_cast<const uint32_t*>(nullptr),
                                                            static_cast<uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr),
                                                            static_cast<uint32_t*>(nullptr), static_cast<int>(maxN), 0, 18);
    if (e != hipSuccess) return rt_internal_fail(RT_ERR_HIP, (std::string("radix sort sizing: ") + hipGetErrorString(e)).c_str());
    return RT_OK;
}

// Sorts queue[0..n) by key, through `alt` (n entries) and back into `queue`.  A = {origin.xyz, direction.x},
// B = {direction.yz, ...} per slot.  All work is queued on `stream`; nothing is allocated.
int rt_internal_sort_rays(const void* A, const void* B, uint32_t* queue, uint32_t* alt, uint32_t* keys0, uint32_t* keys1,
                          void* temp, size_t tempBytes, uint32_t n, const float rootLo[3], const float rootHi[3],
                          hipStream_t stream) {
    if (n < 2) return RT_OK;
    float scale[3];
    for (int k = 0; k < 3; ++k) {
        float const extent = rootHi[k] - rootLo[k];
        scale[k] = extent > 0.0f ? 32.0f / extent : 0.0f;
    }
    hipLaunchKernelGGL(k_sort_keys, dim3((n + 255) / 256), dim3(256), 0, stream, static_cast<const float4*>(A),
                       static_cast<const float4*>(B), queue, n, rootLo[0], rootLo[1], rootLo[2], scale[0], scale[1], scale[2], keys0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, static_cast<const uint32_t*>(keys0), keys1,
                                               static_cast<const uint32_t*>(queue), alt, static_cast<int>(n), 0, 18, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(queue, alt, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return rt_internal_fail(RT_ERR_HIP, (std::string("ray queue sort: ") + hipGetErrorString(e)).c_str());
    return RT_OK;
}
