// MI355X (gfx950) wavefront path tracer behind the C ABI of include/rt_amd.h.
//
// Numerics contract: every floating-point operation on the path from a ray to a hit record and from
// a hit record to radiance is IEEE binary32, evaluated in the same order and with the same fused
// multiply-adds as the reference (src/geometry.hpp:99-214, src/bsp.hpp:58-141, src/render.hpp:53-212,
// src/utility/vectorised.hpp:405-481).  This file must be compiled with -ffp-contract=off; fused
// operations are written explicitly with __fmaf_rn.  Division and square root use HIP's default
// correctly rounded forms.
#include "rt_amd.h"
#include "rt_host.h"
#include "rt_flatten.h"
#include "rt_leaf_bounds.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#define RT_STACK_MAX 64          // per-lane traversal stack entries (tree depth bound)
#define RT_BLOCK 256
#define RT_STAGE 256u                // survivors staged per wave before one append to the next queue
#ifndef RT_BVH_LDS_STACK
#define RT_BVH_LDS_STACK 20       // the same for the traversal of the bounding volume hierarchy
#endif
#define RT_LDS_STACK_DEPTH 20   // traversal stack entries kept in LDS (20 KiB per block: eight blocks fill a CU's 160 KiB); deeper ones spill to scratch

namespace {

thread_local std::string g_lastError;

int fail(int code, const std::string& msg) {
    g_lastError = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

#include "rt_dev_common.h"        // data layout, box test and filter, leaf cones and bounds
#include "rt_dev_trace_bsp.h"     // traceStream: the reference's BSP tree
#include "rt_dev_trace_bvh.h"     // traceStreamBvh: the alternative accelerator
#include "rt_dev_kernels.h"       // sampler, path state, kernels

}  // namespace

// ------------------------------------------------------------------------------------------------
// Host side of the C ABI
// ------------------------------------------------------------------------------------------------

// rt_sort.hip
int rt_internal_sort_temp_bytes(uint32_t maxN, size_t* bytes);
int rt_internal_sort_rays(const void* A, const void* B, uint32_t* queue, uint32_t* alt, uint32_t* keys0, uint32_t* keys1,
                          void* temp, size_t tempBytes, uint32_t n, const float rootLo[3], const float rootHi[3],
                          hipStream_t stream);

// Path-state buffers of one render call.  Two sets exist: the scene's own, allocated with hipMalloc and
// grown on demand (rt_render_device), and a fixed-size one carved out of the file-scope `scratch`
// array (rt_render_device_static), which never allocates.
struct Workspace {
    float4 *A = nullptr, *B = nullptr, *L = nullptr, *hit = nullptr;
    float* Tz = nullptr;
    uint32_t *q0 = nullptr, *q1 = nullptr;
    uint32_t* counts = nullptr;         // 1024 words: per-bounce queue sizes, then the trace kernels' fetch cursors
    Counters* counters = nullptr;
    float* accum = nullptr;
    uint64_t paths = 0, pixels = 0;     // capacity
    // RT_FLAG_SORT_RAYS only (allocated on first use): two key arrays and the radix sort's temporary storage
    uint32_t *sortKeys0 = nullptr, *sortKeys1 = nullptr;
    void* sortTemp = nullptr;
    size_t sortTempBytes = 0;
    uint64_t sortPaths = 0;
};

struct rt_scene {
    int device = 0;
    DevScene dev{};
    void* dNodes = nullptr;
    void* dLeafTris = nullptr;
    void* dLeafIds = nullptr;
    void* dTriNormals = nullptr;
    void* dMaterials = nullptr;
    uint64_t deviceBytes = 0;
    uint32_t nodeCount = 0, leafTriCount = 0, depth = 0, modelCount = 0, pretriCount = 0;
    int cuCount = 256;
    Workspace ws;                       // grown on demand by rt_render_device
    Workspace wsStatic;                 // views into this GPU's copy of `scratch`
    void* wsOut = nullptr; uint64_t wsOutPixels = 0;
    // Grid caps (resident blocks) of the persistent kernels, indexed [flag][spill]; filled at creation so
    // that no later call has to query or allocate.
    int gridTrace[2][2] = {{0, 0}, {0, 0}};
    int gridIntersect[2][2] = {{0, 0}, {0, 0}};
    bool leafAxesMany = false;          // form of the leaf bounds records (rt_leaf_bounds.h): selects the kernel variant
    // Blocks of k_shade per CU: no more than are resident at once (a grid of 8 per CU against 6 resident left a tail at a
    // third of the occupancy: 2.46 -> 2.32 ms per 16 spp on the 0.9 M-triangle scene), and 5 measure a little better still.
    int shadeBlocksPerCu = 5;
    // Alternative accelerator (rt_scene_attach_bvh4): same DevScene with nodes, leafTris and leafIds replaced.
    bool hasBvh = false;
    DevScene devBvh{};
    void* dBvhNodes = nullptr;
    void* dBvhTris = nullptr;
    void* dBvhIds = nullptr;
    void* dBvhShade = nullptr;
    uint32_t bvhStackNeed = 0, bvhNodeCount = 0;
    uint64_t bvhBytes = 0;               // the hierarchy's share of deviceBytes
    bool bvhMerged = false;              // which traversal shape the hierarchy's size calls for (see attachBvh)
    int gridTraceBvh[2][2] = {{0, 0}, {0, 0}};          // [count][spill], for the shape chosen
    int gridIntersectBvh[2][2] = {{0, 0}, {0, 0}};
    // Events of RT_FLAG_PROFILE_KERNELS, kept for reuse across calls.
    std::vector<hipEvent_t> eventPool;
    // All device work of a handle shares one workspace; a call on another stream first waits for the
    // stream of the previous call.
    hipStream_t lastStream = nullptr;
    bool hasLastStream = false;
    uint32_t lastBatches = 0, lastTraceLaunches = 0;      // of the latest render call (rt_render_multi reads them back)
    void* wsFrame = nullptr; uint64_t wsFramePixels = 0;   // rt_render_multi: the assembled frame on the first scene's GPU
};

// ---- allocation-free workspace -------------------------------------------------------------------
// rt_render_device_static keeps its whole path state in this array: no hipMalloc, hipFree or any other
// allocating call on its path.  Layout (bytes): queue sizes and fetch cursors 4096 | counters 4096 |
// A, B, L, hit 16*P each | two queues 4*P each | accumulator 12*P | throughput.z 4*P, with P = RT_STATIC_MAX_PATHS path
// slots (and at most P pixels per call).  Every GPU holds its own copy at its own address; the host-side
// name is only a handle for hipGetSymbolAddress, never a device address.
#define RT_STATIC_MAX_PATHS (1u << 18)
#define RT_STATIC_BYTES (8192ull + 88ull * RT_STATIC_MAX_PATHS)
#define RT_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_STATIC_BYTES];

namespace {

std::mutex g_scratchMutex;
void* g_scratchPtr[RT_MAX_DEVICES] = {};      // device address of `scratch` on each GPU, resolved once per device
hipStream_t g_scratchStream[RT_MAX_DEVICES] = {};      // the stream of the last rt_render_device_static call on each GPU
bool g_scratchHasStream[RT_MAX_DEVICES] = {};

// Resolves this GPU's address of `scratch` (the device must be current) and carves the workspace out of it.
int bindStaticWorkspace(rt_scene* s) {
    if (s->device < 0 || s->device >= RT_MAX_DEVICES) return RT_OK;      // rt_render_device_static reports it
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (!g_scratchPtr[s->device]) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RT_ERR_HIP, "rt_scene_create: hipGetSymbolAddress returned a null address");
        g_scratchPtr[s->device] = ptr;
    }
    unsigned char* base = static_cast<unsigned char*>(g_scratchPtr[s->device]);
    size_t const P = RT_STATIC_MAX_PATHS;
    Workspace& W = s->wsStatic;
    W.counts = reinterpret_cast<uint32_t*>(base);
    W.counters = reinterpret_cast<Counters*>(base + 4096);
    unsigned char* q = base + 8192;
    W.A = reinterpret_cast<float4*>(q); q += 16 * P;
    W.B = reinterpret_cast<float4*>(q); q += 16 * P;
    W.L = reinterpret_cast<float4*>(q); q += 16 * P;
    W.hit = reinterpret_cast<float4*>(q); q += 16 * P;
    W.q0 = reinterpret_cast<uint32_t*>(q); q += 4 * P;
    W.q1 = reinterpret_cast<uint32_t*>(q); q += 4 * P;
    W.accum = reinterpret_cast<float*>(q); q += 12 * P;
    W.Tz = reinterpret_cast<float*>(q);
    W.paths = P;
    W.pixels = P;
    return RT_OK;
}

void freeWorkspace(rt_scene* s) {
    Workspace& W = s->ws;
    for (void* p : {static_cast<void*>(W.A), static_cast<void*>(W.B), static_cast<void*>(W.L), static_cast<void*>(W.hit), static_cast<void*>(W.Tz),
                    static_cast<void*>(W.q0), static_cast<void*>(W.q1), static_cast<void*>(W.accum), static_cast<void*>(W.sortKeys0),
                    static_cast<void*>(W.sortKeys1), W.sortTemp})
        if (p) (void)hipFree(p);
    W.sortKeys0 = W.sortKeys1 = nullptr;
    W.sortTemp = nullptr;
    W.sortTempBytes = 0;
    W.sortPaths = 0;
    W.A = W.B = W.L = W.hit = nullptr;
    W.Tz = nullptr;
    W.q0 = W.q1 = nullptr;
    W.accum = nullptr;
    W.paths = 0;
    W.pixels = 0;
}

// Blocks of the traversal kernels' footprint that are resident per CU on this GPU, measured once per device with
// k_residency_probe (a grid of eight per CU; 0.1 ms when all of it is resident, a few time limits otherwise).
std::mutex g_probeMutex;
int g_probedPerCu[RT_MAX_DEVICES] = {};

int probedBlocksPerCu(const rt_scene* s) {
    int const want = 8;
    if (s->device < 0 || s->device >= RT_MAX_DEVICES) return want;
    std::lock_guard<std::mutex> lock(g_probeMutex);
    if (g_probedPerCu[s->device] > 0) return g_probedPerCu[s->device];
    uint32_t* d = nullptr;
    int result = want;
    if (hipMalloc(&d, 16) == hipSuccess) {
        uint32_t const grid = static_cast<uint32_t>(s->cuCount * want);
        size_t extra = 0;
        if (const char* text = std::getenv("RT_AMD_PROBE_LDS")) extra = std::strtoull(text, nullptr, 10);
        uint32_t h[4] = {0, 0, 0, 0};
        if (hipMemset(d, 0, 16) == hipSuccess) {
            hipLaunchKernelGGL(k_residency_probe, dim3(grid), dim3(RT_BLOCK), extra, nullptr, d, grid, 50000ull /* 0.5 ms */, extra ? 1 : 0);
            if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess
                && hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) == hipSuccess && h[1] > 0)
                result = std::max(1, static_cast<int>(h[2] ? want : h[1] / static_cast<uint32_t>(s->cuCount)));
        }
        (void)hipGetLastError();
        (void)hipFree(d);
        if (std::getenv("RT_AMD_DEBUG"))
            std::fprintf(stderr, "rt_amd: residency probe on device %d: %u of %u blocks alive at once -> %d per CU%s\n", s->device, h[1], grid,
                         result, result < want ? " (fewer than the occupancy query promises: the persistent grids shrink accordingly)" : "");
    }
    g_probedPerCu[s->device] = result;
    return result;
}

// Resident blocks of a persistent kernel over the whole chip: every block must be resident at once (the
// static share of the line stream assumes all waves run concurrently), so the cap is the kernel's own
// occupancy -- as the runtime reports it, and no more than the probe has seen.
template <class Kernel>
int residentBlocks(const rt_scene* s, Kernel kernel, bool deepTree) {
    int perCu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, RT_BLOCK, 0) != hipSuccess || perCu <= 0) perCu = 4;
    if (std::getenv("RT_AMD_DEBUG"))
        std::fprintf(stderr, "rt_amd: persistent kernel %p: %d resident blocks per CU\n", reinterpret_cast<const void*>(kernel), perCu);
    // Trees deeper than the LDS stack (large scenes: most node and triangle loads miss L1 whatever the
    // occupancy) run eight resident blocks per CU, which is what the 20 KiB of stack per block allow:
    // +2.6 % on a 0.9 M-triangle scene over seven.  Shallow trees (small scenes, L1-resident) run seven: with
    // eight, the 32 KB L1 of a CU is shared by more lines at once and its hit rate drops (-2.4 % on a
    // 4 k-triangle scene).  RT_AMD_BLOCKS_PER_CU overrides the cap for tuning.
    int cap = deepTree ? 8 : 7;
    if (const char* text = std::getenv("RT_AMD_BLOCKS_PER_CU")) cap = std::max(1, std::atoi(text));
    return s->cuCount * std::min(std::min(perCu, cap), probedBlocksPerCu(s));
}

int persistentGrid(int residentCap, uint64_t items) {
    static bool const full = std::getenv("RT_AMD_FULL_GRID") != nullptr;     // measurement aid
    if (full) return residentCap;
    uint64_t const blocks = (items + RT_BLOCK - 1) / RT_BLOCK;
    return static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(blocks, static_cast<uint64_t>(residentCap))));
}

int gridFor(const rt_scene* s, uint64_t items, int blocksPerCu = 8) {
    uint64_t const blocks = (items + RT_BLOCK - 1) / RT_BLOCK;
    static int const forced = std::getenv("RT_AMD_SHADE_BLOCKS") ? std::atoi(std::getenv("RT_AMD_SHADE_BLOCKS")) : 0;     // tuning aid
    if (forced > 0 && blocksPerCu != 8) blocksPerCu = forced;
    uint64_t const cap = static_cast<uint64_t>(s->cuCount) * blocksPerCu;     // 8 blocks of 256 per CU = 32 waves/CU
    return static_cast<int>(std::max<uint64_t>(1, std::min(blocks, cap)));
}

// Orders this call behind whatever the handle ran last on another stream.
int adoptStream(rt_scene* s, hipStream_t st) {
    if (s->hasLastStream && s->lastStream != st) HIP_TRY(hipStreamSynchronize(s->lastStream));
    s->lastStream = st;
    s->hasLastStream = true;
    return RT_OK;
}

// No C++ exception may cross the C ABI: every entry point that can allocate runs its body through this.
// (Building the message can itself run out of memory; then the status is returned with an empty message.)
int failQuietly(int code, const char* what, const char* why) noexcept {
    try { g_lastError.assign(what).append(": ").append(why); }
    catch (...) { g_lastError.clear(); }
    return code;
}
template <class Body>
int guarded(const char* what, Body&& body) noexcept {
    try { return body(); }
    catch (std::bad_alloc const&) { return failQuietly(RT_ERR_NOMEM, what, "out of host memory"); }
    catch (std::exception const& e) { return failQuietly(RT_ERR_INVALID, what, e.what()); }
    catch (...) { return failQuietly(RT_ERR_INVALID, what, "unexpected exception"); }
}

}  // namespace

// Error reporting for the other translation units of the library.
int rt_internal_fail(int code, const char* message) { return fail(code, message ? message : ""); }

namespace {

int sceneCreate(const rt_scene_desc* D, int device, rt_scene** out) {
    if (!D || !out) return fail(RT_ERR_INVALID, "rt_scene_create: null argument");
    *out = nullptr;
    rt_flat_scene F;                      // rt_flatten.cpp: validation, node and leaf bounds records, triangle and shading records
    {
        int const rc = rt_internal_flatten(D, F);
        if (rc != RT_OK) return rc;
    }
    if (F.depth > RT_STACK_MAX)
        return fail(RT_ERR_LIMIT, "rt_scene_create: tree depth " + std::to_string(F.depth) + " exceeds the "
                    + std::to_string(RT_STACK_MAX) + "-entry traversal stack");

    // ---- upload ---------------------------------------------------------------------------------
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(RT_ERR_NO_DEVICE, "rt_scene_create: no HIP device available");
    if (device < 0 || device >= count) return fail(RT_ERR_INVALID, "rt_scene_create: device index out of range");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<rt_scene, void (*)(rt_scene*)> holder(new rt_scene, rt_scene_destroy);
    rt_scene* s = holder.get();
    s->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        s->cuCount = prop.multiProcessorCount;
    auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 64));
        if (e != hipSuccess) return e;
        s->deviceBytes += bytes;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    size_t const matBytes = 48 * static_cast<size_t>(std::max<uint32_t>(D->model_count, 1));
    std::vector<float> mats(matBytes / 4, 0.0f);
    if (D->model_count) std::memcpy(mats.data(), D->model_materials, 48 * static_cast<size_t>(D->model_count));
    void* dCounts = nullptr;
    void* dCounters = nullptr;
    hipError_t e = upload(&s->dNodes, F.packed.data(), F.packed.size() * 4);
    if (e == hipSuccess) e = upload(&s->dLeafTris, F.leafTris.data(), F.leafTris.size() * 4);
    if (e == hipSuccess) e = upload(&s->dLeafIds, F.leafIds.data(), F.leafIds.size() * 4);
    if (e == hipSuccess) e = upload(&s->dTriNormals, F.shadeRecs.data(), F.shadeRecs.size() * 4);
    if (e == hipSuccess) e = upload(&s->dMaterials, mats.data(), matBytes);
    if (e == hipSuccess) e = hipMalloc(&dCounts, 4096);
    s->ws.counts = static_cast<uint32_t*>(dCounts);
    if (e == hipSuccess) e = hipMalloc(&dCounters, 1024);
    s->ws.counters = static_cast<Counters*>(dCounters);
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("rt_scene_create: upload failed: ") + hipGetErrorString(e));
    s->dev.nodes = static_cast<const float4*>(s->dNodes) + 4 * static_cast<size_t>(F.frontSlots);     // the leaf bounds lie in front of it
    s->dev.leafTris = static_cast<const float4*>(s->dLeafTris);
    s->dev.leafIds = static_cast<const uint4*>(s->dLeafIds);
    s->dev.shadeRecs = static_cast<const float4*>(s->dTriNormals);
    s->dev.materials = static_cast<const float4*>(s->dMaterials);
    s->dev.boxAbsMax = F.boxAbsMax;
    for (int k = 0; k < 3; ++k) { s->dev.rootLo[k] = F.rootLo[k]; s->dev.rootHi[k] = F.rootHi[k]; }
    s->nodeCount = F.nodeSlots;
    s->leafTriCount = F.refCount;
    s->depth = F.depth;
    s->modelCount = D->model_count;
    s->pretriCount = D->pretri_count;
    s->leafAxesMany = F.leafAxes == RT_LEAF_AXES_PACKED;
#define RT_GRIDS(AXES)                                                                                     \
    do {                                                                                                   \
        s->gridTrace[0][0] = residentBlocks(s, k_trace<false, false, AXES>, false);                        \
        s->gridTrace[0][1] = residentBlocks(s, k_trace<false, true, AXES>, true);                          \
        s->gridTrace[1][0] = residentBlocks(s, k_trace<true, false, AXES>, false);                         \
        s->gridTrace[1][1] = residentBlocks(s, k_trace<true, true, AXES>, true);                           \
        s->gridIntersect[0][0] = residentBlocks(s, k_intersect<false, false, false, AXES>, false);         \
        s->gridIntersect[0][1] = residentBlocks(s, k_intersect<false, true, false, AXES>, true);           \
        s->gridIntersect[1][0] = residentBlocks(s, k_intersect<true, false, false, AXES>, false);          \
        s->gridIntersect[1][1] = residentBlocks(s, k_intersect<true, true, false, AXES>, true);            \
    } while (0)
    if (s->leafAxesMany) RT_GRIDS(RT_LEAF_AXES_PACKED); else RT_GRIDS(RT_LEAF_AXES_ONE);
#undef RT_GRIDS
    {
        int perCu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_shade, RT_BLOCK, 0) == hipSuccess && perCu > 0) s->shadeBlocksPerCu = std::min(perCu, 5);
    }
    int const rc = bindStaticWorkspace(s);
    if (rc != RT_OK) return rc;
    *out = holder.release();
    return RT_OK;
}

void freeBvh(rt_scene* s) {
    for (void* p : {s->dBvhNodes, s->dBvhTris, s->dBvhIds, s->dBvhShade})
        if (p) (void)hipFree(p);
    s->dBvhNodes = s->dBvhTris = s->dBvhIds = s->dBvhShade = nullptr;
    s->hasBvh = false;
    s->deviceBytes -= s->bvhBytes;
    s->bvhBytes = 0;
}

int attachBvh(rt_scene* s, const rt_scene_desc* D, const rt_bvh4_node* nodes, uint32_t nodeCount, const uint32_t* order) {
    if (!s || !D || !nodes || nodeCount == 0) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: null argument");
    if (D->pretri_count && (!D->pretris || !order || !D->model_pretri_begin))
        return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: the description must carry pretris and the triangle order");
    if (D->model_count != s->modelCount || D->pretri_count != s->pretriCount)
        return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: the description is not the one the scene was created from");
    uint32_t const triCount = D->pretri_count;
    if (triCount >= (1u << 28) - 1u) return fail(RT_ERR_LIMIT, "rt_scene_attach_bvh4: more than 2^28 triangles");
    // ---- validate what the kernels assume ---------------------------------------------------------------
    std::vector<uint32_t> need(nodeCount, 0u);
    std::vector<uint8_t> referenced(nodeCount, 0);
    std::vector<uint8_t> covered(std::max<uint32_t>(triCount, 1), 0);
    for (uint32_t i = nodeCount; i-- > 0;) {
        uint32_t present = 0, deepest = 0;
        for (int k = 0; k < 4; ++k) {
            uint32_t const c = nodes[i].child[k];
            if (c == 0xFFFFFFFFu) continue;
            ++present;
            if (c & 0x80000000u) {
                uint32_t const first = c & 0x0FFFFFFFu, count = ((c >> 28) & 7u) + 1u;
                if (static_cast<uint64_t>(first) + count > triCount) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: leaf range outside the triangle order");
                for (uint32_t j = first; j < first + count; ++j) {
                    if (covered[j]) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: triangle in two leaves");
                    covered[j] = 1;
                }
            }
            else {
                if (c <= i || c >= nodeCount) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: child number not above its parent's or out of range");
                if (referenced[c]) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: node referenced twice");
                referenced[c] = 1;
                deepest = std::max(deepest, need[c]);
            }
        }
        need[i] = (present ? present - 1u : 0u) + deepest;
    }
    for (uint32_t i = 1; i < nodeCount; ++i)
        if (!referenced[i]) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: unreachable node");
    for (uint32_t j = 0; j < triCount; ++j)
        if (!covered[j]) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: triangle in no leaf");
    if (need[0] > RT_STACK_MAX)
        return fail(RT_ERR_LIMIT, "rt_scene_attach_bvh4: a descent can defer " + std::to_string(need[0]) + " children, more than the "
                    + std::to_string(RT_STACK_MAX) + "-entry traversal stack");
    // ---- triangle records and ids in the hierarchy's order -----------------------------------------------
    std::vector<float> tris(12 * static_cast<size_t>(std::max<uint32_t>(triCount, 1)), 0.0f);
    std::vector<uint32_t> ids(4 * static_cast<size_t>(std::max<uint32_t>(triCount, 1)), 0u);
    std::vector<float> shadeRecs(12 * static_cast<size_t>(std::max<uint32_t>(triCount, 1)), 0.0f);
    {
        if (D->model_count && (!D->model_vertex_begin || !D->model_tri_begin || (D->vertex_count && !D->vertex_normals)
                               || (D->base_tri_count && !D->base_tris)))
            return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: null array in description");
        std::vector<float> triNormals;
        int const rcn = rt_internal_gather_tri_normals(D, "rt_scene_attach_bvh4", triNormals);
        if (rcn != RT_OK) return rcn;
        std::vector<uint8_t> used(std::max<uint32_t>(triCount, 1), 0);
        uint32_t model = 0;
        std::vector<uint32_t> modelOf(std::max<uint32_t>(triCount, 1), 0u);
        for (uint32_t g = 0; g < triCount; ++g) {
            while (model + 1 < D->model_count && g >= D->model_pretri_begin[model + 1]) ++model;
            modelOf[g] = model;
        }
        for (uint32_t k = 0; k < triCount; ++k) {
            uint32_t const g = order[k];
            if (g >= triCount || used[g]) return fail(RT_ERR_INVALID, "rt_scene_attach_bvh4: tri_order is not a permutation");
            used[g] = 1;
            std::memcpy(tris.data() + 12 * static_cast<size_t>(k), D->pretris + 12 * static_cast<size_t>(g), 48);
            uint32_t const m = modelOf[g];
            uint32_t const rec[4] = {m, g - D->model_pretri_begin[m], g, 0u};
            std::memcpy(ids.data() + 4 * static_cast<size_t>(k), rec, 16);
            rt_internal_shade_record(shadeRecs.data() + 12 * static_cast<size_t>(k), m, triNormals.data() + 9 * static_cast<size_t>(g));
        }
    }
    // ---- upload ---------------------------------------------------------------------------------------------
    HIP_TRY(hipSetDevice(s->device));
    if (s->hasLastStream) HIP_TRY(hipStreamSynchronize(s->lastStream));
    freeBvh(s);
    auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 128));
        if (e != hipSuccess) return e;
        s->deviceBytes += bytes;
        s->bvhBytes += bytes;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    hipError_t e = upload(&s->dBvhNodes, nodes, sizeof(rt_bvh4_node) * static_cast<size_t>(nodeCount));
    if (e == hipSuccess) e = upload(&s->dBvhTris, tris.data(), tris.size() * 4);
    if (e == hipSuccess) e = upload(&s->dBvhIds, ids.data(), ids.size() * 4);
    if (e == hipSuccess) e = upload(&s->dBvhShade, shadeRecs.data(), shadeRecs.size() * 4);
    if (e != hipSuccess) {
        freeBvh(s);
        return fail(RT_ERR_HIP, std::string("rt_scene_attach_bvh4: upload failed: ") + hipGetErrorString(e));
    }
    s->devBvh = s->dev;
    s->devBvh.nodes = static_cast<const float4*>(s->dBvhNodes);
    s->devBvh.leafTris = static_cast<const float4*>(s->dBvhTris);
    s->devBvh.leafIds = static_cast<const uint4*>(s->dBvhIds);
    s->devBvh.shadeRecs = static_cast<const float4*>(s->dBvhShade);
    s->devBvh.refillMin = 16u;        // measured flat between 8 and 24 (3 % better than storing and fetching lane by lane)
    if (const char* text = std::getenv("RT_AMD_BVH_REFILL")) s->devBvh.refillMin = static_cast<uint32_t>(std::max(1, std::atoi(text)));     // tuning aid
    s->bvhStackNeed = need[0];
    s->bvhNodeCount = nodeCount;
    // Two shapes of the traversal loop.  Small hierarchies (nodes and triangles stay in L1/L2, the loop is paced by
    // instruction issue) run the two-phase loop; large ones (paced by memory latency) the one-step-per-iteration
    // loop, where no lane waits for another.  Measured, 1080p x 16 spp, ms per frame in k_trace_bvh, two-phase /
    // one-step: 3.9 k triangles 17.1 / 19.0, 0.9 M triangles 16.6 / 14.8, 9.9 M triangles (8 spp) 12.9 / 10.3.
    // RT_AMD_BVH_MERGED=0/1 overrides the choice for tuning.
    uint64_t const bvhBytes = sizeof(rt_bvh4_node) * static_cast<uint64_t>(nodeCount) + 48ull * triCount;
    s->bvhMerged = bvhBytes > (4ull << 20);
    if (const char* text = std::getenv("RT_AMD_BVH_MERGED")) s->bvhMerged = std::atoi(text) != 0;
    if (s->bvhMerged) {
        s->gridTraceBvh[0][0] = residentBlocks(s, k_trace_bvh<false, false, true>, true);
        s->gridTraceBvh[0][1] = residentBlocks(s, k_trace_bvh<false, true, true>, true);
        s->gridTraceBvh[1][0] = residentBlocks(s, k_trace_bvh<true, false, true>, true);
        s->gridTraceBvh[1][1] = residentBlocks(s, k_trace_bvh<true, true, true>, true);
        s->gridIntersectBvh[0][0] = residentBlocks(s, k_intersect_bvh<false, false, true>, true);
        s->gridIntersectBvh[0][1] = residentBlocks(s, k_intersect_bvh<false, true, true>, true);
        s->gridIntersectBvh[1][0] = residentBlocks(s, k_intersect_bvh<true, false, true>, true);
        s->gridIntersectBvh[1][1] = residentBlocks(s, k_intersect_bvh<true, true, true>, true);
    }
    else {
        s->gridTraceBvh[0][0] = residentBlocks(s, k_trace_bvh<false, false, false>, true);
        s->gridTraceBvh[0][1] = residentBlocks(s, k_trace_bvh<false, true, false>, true);
        s->gridTraceBvh[1][0] = residentBlocks(s, k_trace_bvh<true, false, false>, true);
        s->gridTraceBvh[1][1] = residentBlocks(s, k_trace_bvh<true, true, false>, true);
        s->gridIntersectBvh[0][0] = residentBlocks(s, k_intersect_bvh<false, false, false>, true);
        s->gridIntersectBvh[0][1] = residentBlocks(s, k_intersect_bvh<false, true, false>, true);
        s->gridIntersectBvh[1][0] = residentBlocks(s, k_intersect_bvh<true, false, false>, true);
        s->gridIntersectBvh[1][1] = residentBlocks(s, k_intersect_bvh<true, true, false>, true);
    }
    s->hasBvh = true;
    return RT_OK;
}

// The body of render(): batches of path slots through raygen, (trace, shade) per bounce and accumulate.
// Uses only the buffers of W and calls nothing that allocates unless `profile` asks for new events.
int renderCore(rt_scene* s, const rt_render_params* p, float* d_out, rt_stats* stats, hipStream_t st, const Workspace& W,
               uint32_t spb, bool profile) {
    auto const t0 = std::chrono::steady_clock::now();
    uint64_t const pixels = static_cast<uint64_t>(p->w) * p->h;
    uint32_t const bounceCount = p->bounce_limit + 1;      // intersections per path, src/render.hpp:88

    PathState P;
    P.A = W.A; P.B = W.B; P.L = W.L; P.Tz = W.Tz; P.hit = W.hit;
    P.queue[0] = W.q0; P.queue[1] = W.q1;
    P.counts = W.counts;
    Counters* gc = W.counters;

    RenderConst R;
    std::memcpy(R.camPos, p->camera_position, 12);
    std::memcpy(R.m, p->pixel_to_ray, 36);
    R.width = p->width; R.x0 = p->x0; R.y0 = p->y0; R.w = p->w; R.h = p->h;
    R.bounceLimit = p->bounce_limit; R.tMin = p->t_min;
    R.key0 = static_cast<uint32_t>(p->seed); R.key1 = static_cast<uint32_t>(p->seed >> 32);
    R.roulette = (p->flags & RT_FLAG_RUSSIAN_ROULETTE) ? 1u : 0u;
    R.lobes = (p->flags & RT_FLAG_LOBE_SAMPLING) ? 1u : 0u;
    bool const interleaved = p->row_stride > 1u && p->row_group > 0u;
    R.rowGroup = interleaved ? p->row_group : 1u;
    R.rowSkip = interleaved ? p->row_group * (p->row_stride - 1u) : 0u;

    bool const countWork = (p->flags & RT_FLAG_COUNT_WORK) != 0;
    bool const sortRays = (p->flags & RT_FLAG_SORT_RAYS) != 0;
    if (sortRays && !W.sortKeys0) return fail(RT_ERR_INVALID, "rt_render: RT_FLAG_SORT_RAYS needs buffers this entry point does not have");
    size_t eventsUsed = 0;
    std::vector<int8_t> evKind;           // kind of the interval an event opens, -1 for a closing event
    auto mark = [&](int kind) {
        if (!profile) return;
        if (eventsUsed == s->eventPool.size()) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) return;
            s->eventPool.push_back(ev);
        }
        (void)hipEventRecord(s->eventPool[eventsUsed++], st);
        evKind.push_back(static_cast<int8_t>(kind));
    };

    HIP_TRY(hipMemsetAsync(gc, 0, sizeof(Counters), st));
    HIP_TRY(hipMemsetAsync(W.accum, 0, pixels * 12, st));
    uint32_t batches = 0, traceLaunches = 0;
    // Pixels per slot group (see SlotMap).  RT_AMD_SLOT_GROUP overrides it for tuning.
    // One pixel per group: a wave then holds 64 samples of one pixel (or the samples of a few pixels), whose
    // primary rays walk the same nodes and whose later segments start within one pixel's footprint: measured
    // best (0.9 M-triangle scene, 256 spp: 2 009 Mrays/s with sample-major slots, 2 050 with groups of 64 pixels,
    // 2 072 with 4, 2 098 with 1).
    uint32_t slotGroup = 1u;
    if (const char* text = std::getenv("RT_AMD_SLOT_GROUP")) slotGroup = static_cast<uint32_t>(std::max(1, std::atoi(text)));
    bool const bvh = (p->flags & RT_FLAG_ACCEL_BVH) != 0;
    bool const spill = bvh ? s->bvhStackNeed >= RT_BVH_LDS_STACK : s->depth > RT_LDS_STACK_DEPTH;    // the hierarchy's kernels keep one slot free above the top
    int const traceCap = (bvh ? s->gridTraceBvh : s->gridTrace)[countWork ? 1 : 0][spill ? 1 : 0];
    DevScene const& dev = bvh ? s->devBvh : s->dev;
    for (uint32_t done = 0; done < p->spp; done += spb) {
        uint32_t const nb = std::min(spb, p->spp - done);
        uint32_t const paths = static_cast<uint32_t>(pixels * nb);
        int const grid = gridFor(s, paths);
        int const traceGrid = persistentGrid(traceCap, paths);
        HIP_TRY(hipMemsetAsync(P.counts, 0, 4096, st));
        mark(0);
        SlotMap M;
        M.pixels = static_cast<uint32_t>(pixels); M.samples = nb; M.group = slotGroup;
        hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(RT_BLOCK), 0, st, R, P, M, done, paths);
        mark(-1);
        for (uint32_t b = 0; b < bounceCount; ++b) {
            mark(1);
#define RT_LAUNCH_TRACE(KERNEL, ...)                                                                          \
    hipLaunchKernelGGL((KERNEL<__VA_ARGS__>), dim3(traceGrid), dim3(RT_BLOCK), 0, st, dev, P, b, p->t_min, gc)
#define RT_LAUNCH_TRACE_BVH(MERGED)                                                                            \
    do {                                                                                                       \
        if (countWork) { if (spill) RT_LAUNCH_TRACE(k_trace_bvh, true, true, MERGED); else RT_LAUNCH_TRACE(k_trace_bvh, true, false, MERGED); }     \
        else { if (spill) RT_LAUNCH_TRACE(k_trace_bvh, false, true, MERGED); else RT_LAUNCH_TRACE(k_trace_bvh, false, false, MERGED); }             \
    } while (0)
            if (bvh) {
                if (s->bvhMerged) RT_LAUNCH_TRACE_BVH(true); else RT_LAUNCH_TRACE_BVH(false);
            }
            else {
#define RT_LAUNCH_TRACE_BSP(AXES)                                                                              \
    do {                                                                                                       \
        if (countWork) { if (spill) RT_LAUNCH_TRACE(k_trace, true, true, AXES); else RT_LAUNCH_TRACE(k_trace, true, false, AXES); }     \
        else { if (spill) RT_LAUNCH_TRACE(k_trace, false, true, AXES); else RT_LAUNCH_TRACE(k_trace, false, false, AXES); }             \
    } while (0)
                if (s->leafAxesMany) RT_LAUNCH_TRACE_BSP(RT_LEAF_AXES_PACKED); else RT_LAUNCH_TRACE_BSP(RT_LEAF_AXES_ONE);
#undef RT_LAUNCH_TRACE_BSP
            }
#undef RT_LAUNCH_TRACE_BVH
#undef RT_LAUNCH_TRACE
            mark(-1);
            mark(2);
            hipLaunchKernelGGL(k_shade, dim3(gridFor(s, paths, s->shadeBlocksPerCu)), dim3(RT_BLOCK), 0, st, dev, R, P, b, done, M, gc);
            if (sortRays && b + 1 < bounceCount) {
                // the next bounce's queue, re-ordered by origin cell and direction octant (timed with the shading)
                uint32_t live = 0;
                HIP_TRY(hipMemcpyAsync(&live, P.counts + b + 1, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                int const rcSort = rt_internal_sort_rays(P.A, P.B, P.queue[(b + 1) & 1], P.queue[b & 1], W.sortKeys0, W.sortKeys1,
                                                         W.sortTemp, W.sortTempBytes, live, s->dev.rootLo, s->dev.rootHi, st);
                if (rcSort != RT_OK) return rcSort;
            }
            mark(-1);
            ++traceLaunches;
        }
        int const last = done + nb >= p->spp;
        mark(3);
        hipLaunchKernelGGL(k_accumulate, dim3(static_cast<unsigned>((pixels + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK),
                           0, st, P, W.accum, M, p->w, p->h, bounceCount, last,
                           static_cast<float>(p->spp), d_out, gc);
        mark(-1);
        ++batches;
    }
    HIP_TRY(hipGetLastError());
    s->lastBatches = batches;
    s->lastTraceLaunches = traceLaunches;
    if (stats) {
        Counters hc;
        HIP_TRY(hipMemcpyAsync(&hc, gc, sizeof hc, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::memset(stats, 0, sizeof *stats);
        stats->paths = hc.paths; stats->segments = hc.segments; stats->shaded_hits = hc.shadedHits;
        stats->box_tests = hc.boxTests; stats->tris_tested = hc.trisTested; stats->leaf_visits = hc.leafVisits;
        stats->inode_visits = hc.inodeVisits;
        stats->box_exact_fallbacks = hc.boxFallbacks; stats->box_filter_errors = hc.boxFilterErrors;
        stats->trace_launches = traceLaunches; stats->batches = batches;
        if (profile) {
            double acc[4] = {0, 0, 0, 0};
            for (size_t i = 0; i + 1 < eventsUsed; i += 2) {
                float ms = 0.0f;
                if (evKind[i] >= 0 && hipEventElapsedTime(&ms, s->eventPool[i], s->eventPool[i + 1]) == hipSuccess) acc[evKind[i]] += ms;
            }
            stats->ms_raygen = acc[0]; stats->ms_trace = acc[1]; stats->ms_shade = acc[2]; stats->ms_accumulate = acc[3];
        }
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (std::getenv("RT_AMD_DEBUG")) {           // queue sizes of the last batch, per bounce
            uint32_t counts[32] = {};
            (void)hipMemcpy(counts, P.counts, sizeof(uint32_t) * std::min<uint32_t>(bounceCount + 1, 32), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "rt_amd: paths entering each bounce of the last batch:");
            for (uint32_t b = 0; b < std::min<uint32_t>(bounceCount + 1, 32); ++b) std::fprintf(stderr, " %u", counts[b]);
            std::fprintf(stderr, "\n");
        }
    }
    return RT_OK;
}

int checkRenderParams(const rt_scene* s, const rt_render_params* p, const float* d_out) {
    if (!s || !p || !d_out) return fail(RT_ERR_INVALID, "rt_render: null argument");
    if (p->w == 0 || p->h == 0 || p->spp == 0) return fail(RT_ERR_INVALID, "rt_render: empty rectangle or zero spp");
    uint64_t lastRow = static_cast<uint64_t>(p->y0) + p->h - 1;
    if (p->row_stride > 1u) {
        if (p->row_group == 0u) return fail(RT_ERR_INVALID, "rt_render: row_stride without row_group");
        lastRow = static_cast<uint64_t>(p->y0) + (p->h - 1) + static_cast<uint64_t>((p->h - 1) / p->row_group) * p->row_group * (p->row_stride - 1u);
    }
    if (static_cast<uint64_t>(p->x0) + p->w > p->width || lastRow >= p->height)
        return fail(RT_ERR_INVALID, "rt_render: rectangle outside the image");
    if (static_cast<uint64_t>(p->width) * p->height > 0xFFFFFFFFull)
        return fail(RT_ERR_LIMIT, "rt_render: more than 2^32 pixels");
    if (p->bounce_limit > 500 || p->bounce_limit + 3 > RT_FETCH_BASE) return fail(RT_ERR_LIMIT, "rt_render: bounce_limit above 500");
    if ((p->flags & RT_FLAG_ACCEL_BVH) && !s->hasBvh)
        return fail(RT_ERR_INVALID, "rt_render: RT_FLAG_ACCEL_BVH without a hierarchy attached (rt_scene_attach_bvh4)");
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_lastError.c_str(); }
uint32_t rt_abi_version(void) { return 7; }

int rt_scene_create(const rt_scene_desc* D, int device, rt_scene** out) {
    return guarded("rt_scene_create", [&] { return sceneCreate(D, device, out); });
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->hasLastStream) (void)hipStreamSynchronize(s->lastStream);
    freeWorkspace(s);
    freeBvh(s);
    for (void* p : {s->dNodes, s->dLeafTris, s->dLeafIds, s->dTriNormals, s->dMaterials, static_cast<void*>(s->ws.counts),
                    static_cast<void*>(s->ws.counters), s->wsOut, s->wsFrame})
        if (p) (void)hipFree(p);
    for (hipEvent_t ev : s->eventPool) (void)hipEventDestroy(ev);
    delete s;
}

int rt_scene_attach_bvh4(rt_scene* s, const rt_scene_desc* D, const rt_bvh4_node* nodes, uint32_t node_count,
                         const uint32_t* tri_order) {
    return guarded("rt_scene_attach_bvh4", [&] { return attachBvh(s, D, nodes, node_count, tri_order); });
}

int rt_scene_build_bvh4(rt_scene* s, const rt_scene_desc* D, uint32_t max_leaf_tris, int threads) {
    return guarded("rt_scene_build_bvh4", [&]() -> int {
        if (!s || !D) return fail(RT_ERR_INVALID, "rt_scene_build_bvh4: null argument");
        if (D->pretri_count && !D->pretris) return fail(RT_ERR_INVALID, "rt_scene_build_bvh4: the description must carry pretris");
        rt_bvh4* bvh = nullptr;
        int rc = rt_bvh4_build(D->pretris, D->pretri_count, max_leaf_tris, threads, &bvh);
        if (rc != RT_OK) return rc;
        const rt_bvh4_node* nodes = nullptr;
        const uint32_t* order = nullptr;
        uint32_t counts[6] = {};
        rt_bvh4_view(bvh, &nodes, &order);
        rt_bvh4_counts(bvh, counts);
        rc = attachBvh(s, D, nodes, counts[0], order);
        rt_bvh4_destroy(bvh);
        return rc;
    });
}

int rt_scene_info(const rt_scene* s, uint64_t* device_bytes, uint32_t* node_count, uint32_t* leaf_tri_count,
                  uint32_t* depth) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_info: null scene");
    if (device_bytes) *device_bytes = s->deviceBytes;
    if (node_count) *node_count = s->nodeCount;
    if (leaf_tri_count) *leaf_tri_count = s->leafTriCount;
    if (depth) *depth = s->depth;
    return RT_OK;
}

namespace {
int intersectDevice(rt_scene* s, const float* d_origins, const float* d_directions, uint64_t n, float t_min,
                    int front_only, rt_hit* d_out, void* hip_stream) {
    if (!s || (n && (!d_origins || !d_directions || !d_out))) return fail(RT_ERR_INVALID, "rt_intersect: null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t const st = static_cast<hipStream_t>(hip_stream);
    int const rcs = adoptStream(s, st);
    if (rcs != RT_OK) return rcs;
    // The kernel hands out lines through a 32-bit cursor: larger batches go in slices.
    uint64_t const slice = 0xF0000000ull;
    bool const bvh = (front_only & RT_INTERSECT_BVH) != 0;
    bool const front = (front_only & ~RT_INTERSECT_BVH) != 0;
    if (bvh && !s->hasBvh) return fail(RT_ERR_INVALID, "rt_intersect: RT_INTERSECT_BVH without a hierarchy attached (rt_scene_attach_bvh4)");
    bool const spill = bvh ? s->bvhStackNeed >= RT_BVH_LDS_STACK : s->depth > RT_LDS_STACK_DEPTH;    // the hierarchy's kernels keep one slot free above the top
    int const cap = (bvh ? s->gridIntersectBvh : s->gridIntersect)[front ? 1 : 0][spill ? 1 : 0];
    DevScene const& dev = bvh ? s->devBvh : s->dev;
    for (uint64_t done = 0; done < n; done += slice) {
        uint32_t const m = static_cast<uint32_t>(std::min<uint64_t>(slice, n - done));
        uint32_t* cursor = reinterpret_cast<uint32_t*>(s->ws.counters) + 64;     // behind the Counters block
        HIP_TRY(hipMemsetAsync(cursor, 0, 4, st));
#define RT_LAUNCH_INTERSECT(KERNEL, ...)                                                                      \
    hipLaunchKernelGGL((KERNEL<__VA_ARGS__>), dim3(persistentGrid(cap, m)), dim3(RT_BLOCK), 0, st, dev,        \
                       d_origins + 3 * done, d_directions + 3 * done, m, t_min, d_out + done, cursor)
#define RT_LAUNCH_INTERSECT_GC(KERNEL, ...)                                                                   \
    hipLaunchKernelGGL((KERNEL<__VA_ARGS__>), dim3(persistentGrid(cap, m)), dim3(RT_BLOCK), 0, st, dev,        \
                       d_origins + 3 * done, d_directions + 3 * done, m, t_min, d_out + done, cursor, static_cast<Counters*>(nullptr))
#define RT_LAUNCH_INTERSECT_BVH(MERGED)                                                                        \
    do {                                                                                                       \
        if (front) { if (spill) RT_LAUNCH_INTERSECT(k_intersect_bvh, true, true, MERGED); else RT_LAUNCH_INTERSECT(k_intersect_bvh, true, false, MERGED); }     \
        else { if (spill) RT_LAUNCH_INTERSECT(k_intersect_bvh, false, true, MERGED); else RT_LAUNCH_INTERSECT(k_intersect_bvh, false, false, MERGED); }         \
    } while (0)
        if (bvh) {
            if (s->bvhMerged) RT_LAUNCH_INTERSECT_BVH(true); else RT_LAUNCH_INTERSECT_BVH(false);
        }
        else {
#define RT_LAUNCH_INTERSECT_BSP(AXES)                                                                          \
    do {                                                                                                       \
        if (front) { if (spill) RT_LAUNCH_INTERSECT_GC(k_intersect, true, true, false, AXES); else RT_LAUNCH_INTERSECT_GC(k_intersect, true, false, false, AXES); }     \
        else { if (spill) RT_LAUNCH_INTERSECT_GC(k_intersect, false, true, false, AXES); else RT_LAUNCH_INTERSECT_GC(k_intersect, false, false, false, AXES); }         \
    } while (0)
            if (s->leafAxesMany) RT_LAUNCH_INTERSECT_BSP(RT_LEAF_AXES_PACKED); else RT_LAUNCH_INTERSECT_BSP(RT_LEAF_AXES_ONE);
#undef RT_LAUNCH_INTERSECT_BSP
        }
#undef RT_LAUNCH_INTERSECT_BVH
#undef RT_LAUNCH_INTERSECT_GC
#undef RT_LAUNCH_INTERSECT
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}
}  // namespace

int rt_intersect_device(rt_scene* s, const float* d_origins, const float* d_directions, uint64_t n, float t_min,
                        int front_only, rt_hit* d_out, void* hip_stream) {
    return guarded("rt_intersect_device", [&] { return intersectDevice(s, d_origins, d_directions, n, t_min, front_only, d_out, hip_stream); });
}

namespace {
// Host-pointer form shared by rt_intersect and rt_intersect_check (stats != nullptr: the counting traversal of the BSP tree).
int intersectHost(rt_scene* s, const float* origins, const float* directions, uint64_t n, float t_min, int front_only,
                  rt_hit* out, rt_stats* stats) {
    if (!s || (n && (!origins || !directions || !out))) return fail(RT_ERR_INVALID, "rt_intersect: null argument");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    if (stats && (front_only & RT_INTERSECT_BVH)) return fail(RT_ERR_INVALID, "rt_intersect_check: the BSP tree only");
    if (stats && n > 0xF0000000ull) return fail(RT_ERR_LIMIT, "rt_intersect_check: too many lines");
    HIP_TRY(hipSetDevice(s->device));
    float *dO = nullptr, *dD = nullptr;
    rt_hit* dH = nullptr;
    int rc = RT_OK;
    hipError_t e = hipMalloc(&dO, n * 12);
    if (e == hipSuccess) e = hipMalloc(&dD, n * 12);
    if (e == hipSuccess) e = hipMalloc(&dH, n * sizeof(rt_hit));
    if (e == hipSuccess) e = hipMemcpy(dO, origins, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dD, directions, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess && !stats) {
        rc = intersectDevice(s, dO, dD, n, t_min, front_only, dH, nullptr);
        if (rc == RT_OK) e = hipDeviceSynchronize();
    }
    else if (e == hipSuccess) {
        if (s->hasLastStream) e = hipStreamSynchronize(s->lastStream);
        Counters* gc = s->ws.counters;
        uint32_t* cursor = reinterpret_cast<uint32_t*>(gc) + 64;     // behind the Counters block
        if (e == hipSuccess) e = hipMemset(gc, 0, 512);      // the counters and the cursor
        bool const front = (front_only & RT_INTERSECT_FRONT_ONLY) != 0, spill = s->depth > RT_LDS_STACK_DEPTH;
        uint32_t const m = static_cast<uint32_t>(n);
#define RT_LAUNCH_CHECK(F, S, AXES)                                                                                  \
    hipLaunchKernelGGL((k_intersect<F, S, true, AXES>), dim3(persistentGrid(residentBlocks(s, k_intersect<F, S, true, AXES>, S), m)), \
                       dim3(RT_BLOCK), 0, nullptr, s->dev, dO, dD, m, t_min, dH, cursor, gc)
#define RT_LAUNCH_CHECK_BSP(AXES)                                                                                    \
    do {                                                                                                             \
        if (front) { if (spill) RT_LAUNCH_CHECK(true, true, AXES); else RT_LAUNCH_CHECK(true, false, AXES); }        \
        else { if (spill) RT_LAUNCH_CHECK(false, true, AXES); else RT_LAUNCH_CHECK(false, false, AXES); }            \
    } while (0)
        if (e == hipSuccess) {
            if (s->leafAxesMany) RT_LAUNCH_CHECK_BSP(RT_LEAF_AXES_PACKED); else RT_LAUNCH_CHECK_BSP(RT_LEAF_AXES_ONE);
            e = hipGetLastError();
        }
#undef RT_LAUNCH_CHECK_BSP
#undef RT_LAUNCH_CHECK
        if (e == hipSuccess) e = hipDeviceSynchronize();
        Counters hc;
        if (e == hipSuccess) e = hipMemcpy(&hc, gc, sizeof hc, hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            stats->segments = n;
            stats->box_tests = hc.boxTests; stats->tris_tested = hc.trisTested; stats->leaf_visits = hc.leafVisits;
            stats->inode_visits = hc.inodeVisits;
            stats->box_exact_fallbacks = hc.boxFallbacks; stats->box_filter_errors = hc.boxFilterErrors;
        }
    }
    if (rc == RT_OK && e == hipSuccess) e = hipMemcpy(out, dH, n * sizeof(rt_hit), hipMemcpyDeviceToHost);
    if (dO) (void)hipFree(dO);
    if (dD) (void)hipFree(dD);
    if (dH) (void)hipFree(dH);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("rt_intersect: ") + hipGetErrorString(e));
    return RT_OK;
}
}  // namespace

int rt_intersect(rt_scene* s, const float* origins, const float* directions, uint64_t n, float t_min, int front_only,
                 rt_hit* out) {
    return guarded("rt_intersect", [&] { return intersectHost(s, origins, directions, n, t_min, front_only, out, nullptr); });
}

int rt_intersect_check(rt_scene* s, const float* origins, const float* directions, uint64_t n, float t_min, int front_only,
                       rt_hit* out, rt_stats* stats) {
    if (!stats) return fail(RT_ERR_INVALID, "rt_intersect_check: null stats");
    return guarded("rt_intersect_check", [&] { return intersectHost(s, origins, directions, n, t_min, front_only, out, stats); });
}

int rt_render_device(rt_scene* s, const rt_render_params* p, float* d_out, rt_stats* stats, void* hip_stream) {
    return guarded("rt_render_device", [&]() -> int {
        int const rc = checkRenderParams(s, p, d_out);
        if (rc != RT_OK) return rc;
        HIP_TRY(hipSetDevice(s->device));
        hipStream_t const st = static_cast<hipStream_t>(hip_stream);
        int const rcs = adoptStream(s, st);
        if (rcs != RT_OK) return rcs;
        uint64_t const pixels = static_cast<uint64_t>(p->w) * p->h;
        // Paths per batch.  Every bounce of a batch is one launch whose last rays set its duration (a few hundred
        // microseconds for the slowest line of a deep tree, however few lines are left), so batches should be as
        // large as memory allows: 80 bytes of HBM per path slot.  Default: the whole call in one batch, up to
        // 2^30 paths (82 GB) and half of the device's memory; halved again for as long as the allocation fails.
        uint64_t maxPaths = p->max_batch_paths;
        if (const char* text = std::getenv("RT_AMD_BATCH_PATHS")) maxPaths = std::max<uint64_t>(1, std::strtoull(text, nullptr, 10));
        bool const automatic = maxPaths == 0;
        if (automatic) {
            size_t freeBytes = 0, totalBytes = 0;
            maxPaths = 1ull << 30;
            if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && totalBytes > 0)
                maxPaths = std::min<uint64_t>(maxPaths, std::max<uint64_t>(16ull << 20, totalBytes / 2 / 80));
        }
        Workspace& W = s->ws;
        uint32_t spb = 0;
        for (;;) {
            spb = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(p->spp, maxPaths / pixels)));
            uint64_t const batchPaths = pixels * spb;
            if (batchPaths > 0xFFFFFFF0ull) return fail(RT_ERR_LIMIT, "rt_render: rectangle too large for one batch");
            if (W.paths >= batchPaths && W.pixels >= pixels) break;
            HIP_TRY(hipStreamSynchronize(st));
            freeWorkspace(s);
            void* ptr = nullptr;
            hipError_t e = hipMalloc(&ptr, batchPaths * 16); W.A = static_cast<float4*>(ptr); ptr = nullptr;
            if (e == hipSuccess) { e = hipMalloc(&ptr, batchPaths * 16); W.B = static_cast<float4*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) { e = hipMalloc(&ptr, batchPaths * 16); W.L = static_cast<float4*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) { e = hipMalloc(&ptr, batchPaths * 4); W.Tz = static_cast<float*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) { e = hipMalloc(&ptr, batchPaths * 16); W.hit = static_cast<float4*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) { e = hipMalloc(&ptr, batchPaths * 4); W.q0 = static_cast<uint32_t*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) { e = hipMalloc(&ptr, batchPaths * 4); W.q1 = static_cast<uint32_t*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) { e = hipMalloc(&ptr, pixels * 12); W.accum = static_cast<float*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) {
                W.paths = batchPaths;
                W.pixels = pixels;
                break;
            }
            freeWorkspace(s);
            (void)hipGetLastError();
            if (!automatic || spb == 1 || e != hipErrorOutOfMemory)
                return fail(RT_ERR_HIP, std::string("rt_render: workspace allocation failed: ") + hipGetErrorString(e));
            maxPaths = std::max<uint64_t>(pixels, batchPaths / 2);
        }
        if ((p->flags & RT_FLAG_SORT_RAYS) && W.sortPaths < W.paths) {
            HIP_TRY(hipStreamSynchronize(st));
            for (void* q : {static_cast<void*>(W.sortKeys0), static_cast<void*>(W.sortKeys1), W.sortTemp}) if (q) (void)hipFree(q);
            W.sortKeys0 = W.sortKeys1 = nullptr; W.sortTemp = nullptr; W.sortPaths = 0;
            int const rcT = rt_internal_sort_temp_bytes(static_cast<uint32_t>(W.paths), &W.sortTempBytes);
            if (rcT != RT_OK) return rcT;
            void* ptr = nullptr;
            hipError_t e = hipMalloc(&ptr, W.paths * 4); W.sortKeys0 = static_cast<uint32_t*>(ptr); ptr = nullptr;
            if (e == hipSuccess) { e = hipMalloc(&ptr, W.paths * 4); W.sortKeys1 = static_cast<uint32_t*>(ptr); ptr = nullptr; }
            if (e == hipSuccess) e = hipMalloc(&W.sortTemp, std::max<size_t>(W.sortTempBytes, 256));
            if (e != hipSuccess) {
                for (void* q : {static_cast<void*>(W.sortKeys0), static_cast<void*>(W.sortKeys1), W.sortTemp}) if (q) (void)hipFree(q);
                W.sortKeys0 = W.sortKeys1 = nullptr; W.sortTemp = nullptr;
                (void)hipGetLastError();
                return fail(RT_ERR_HIP, std::string("rt_render: allocation of the ray sort's buffers failed: ") + hipGetErrorString(e));
            }
            W.sortPaths = W.paths;
        }
        return renderCore(s, p, d_out, stats, st, W, spb, (p->flags & RT_FLAG_PROFILE_KERNELS) != 0);
    });
}

int rt_render_device_static(rt_scene* s, const rt_render_params* p, float* d_out, rt_stats* stats, void* hip_stream) {
    // (guarded: fail() builds a std::string, and nothing may cross the C ABI as an exception)
    return guarded("rt_render_device_static", [&]() -> int {
        int const rc = checkRenderParams(s, p, d_out);
        if (rc != RT_OK) return rc;
        uint64_t const pixels = static_cast<uint64_t>(p->w) * p->h;
        if (pixels > RT_STATIC_MAX_PATHS)
            return fail(RT_ERR_LIMIT, "rt_render_device_static: rectangle larger than 262144 pixels (RT_STATIC_MAX_PIXELS)");
        if (p->flags & RT_FLAG_PROFILE_KERNELS)
            return fail(RT_ERR_INVALID, "rt_render_device_static: RT_FLAG_PROFILE_KERNELS creates events and is not available here");
        if (p->flags & RT_FLAG_SORT_RAYS)
            return fail(RT_ERR_INVALID, "rt_render_device_static: RT_FLAG_SORT_RAYS needs buffers outside the static workspace and is not available here");
        if (!s->wsStatic.A) return fail(RT_ERR_LIMIT, "rt_render_device_static: no static workspace on this device index");
        HIP_TRY(hipSetDevice(s->device));
        hipStream_t const st = static_cast<hipStream_t>(hip_stream);
        int const rcs = adoptStream(s, st);
        if (rcs != RT_OK) return rcs;
        {
            // Every scene handle on this GPU shares the one array: a call that arrives on another stream than the array's
            // previous user (whichever handle that was) first waits for that stream.  Calls issued concurrently from
            // several host threads still race; that is the caller's to avoid.
            std::lock_guard<std::mutex> lock(g_scratchMutex);
            if (g_scratchHasStream[s->device] && g_scratchStream[s->device] != st) HIP_TRY(hipStreamSynchronize(g_scratchStream[s->device]));
            g_scratchStream[s->device] = st;
            g_scratchHasStream[s->device] = true;
        }
        uint64_t maxPaths = RT_STATIC_MAX_PATHS;
        if (p->max_batch_paths) maxPaths = std::min<uint64_t>(maxPaths, std::max<uint64_t>(p->max_batch_paths, pixels));
        uint32_t const spb = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(p->spp, maxPaths / pixels)));
        return renderCore(s, p, d_out, stats, st, s->wsStatic, spb, false);
    });
}

// render() over several GPUs from one process (the reference's render() is one call, src/render.hpp:215-234).  Rows are
// cut into bands of equal height dealt round-robin to the scenes (eight bands each: a scene's cost per row varies with
// what the rows show), every scene renders all of its bands with ONE call (row_group / row_stride) on its own GPU, all
// GPUs working at once, and the finished bands are copied GPU-to-GPU into the frame on the first scene's GPU.
namespace {
int renderMulti(rt_scene* const* scenes, uint32_t n, const rt_render_params* p, float* d_frame, float* host_frame, rt_stats* stats) {
    auto const t0 = std::chrono::steady_clock::now();
    if (!scenes || n == 0 || !p || (!d_frame && !host_frame)) return fail(RT_ERR_INVALID, "rt_render_multi: null argument");
    for (uint32_t k = 0; k < n; ++k)
        if (!scenes[k]) return fail(RT_ERR_INVALID, "rt_render_multi: null scene");
    if (p->row_stride > 1u) return fail(RT_ERR_INVALID, "rt_render_multi: the rectangle must be plain (row_stride is used internally)");
    if (p->w == 0 || p->h == 0 || p->spp == 0) return fail(RT_ERR_INVALID, "rt_render: empty rectangle or zero spp");
    rt_scene* const first = scenes[0];
    uint64_t const pixels = static_cast<uint64_t>(p->w) * p->h;
    size_t const rowBytes = static_cast<size_t>(p->w) * 12;
    if (!d_frame) {
        HIP_TRY(hipSetDevice(first->device));
        if (first->wsFramePixels < pixels) {
            if (first->hasLastStream) HIP_TRY(hipStreamSynchronize(first->lastStream));
            if (first->wsFrame) (void)hipFree(first->wsFrame);
            first->wsFrame = nullptr; first->wsFramePixels = 0;
            HIP_TRY(hipMalloc(&first->wsFrame, pixels * 12));
            first->wsFramePixels = pixels;
        }
        d_frame = static_cast<float*>(first->wsFrame);
    }
    uint32_t const bands = std::max<uint32_t>(1u, std::min<uint32_t>(p->h, n * 8u));
    uint32_t const group = (p->h + bands - 1) / bands;                  // rows per band (the frame's last band may be shorter)
    uint32_t const bandCount = (p->h + group - 1) / group;
    // ---- enqueue every scene's share (no synchronisation: stats are collected afterwards) ----
    std::vector<uint32_t> rowsOf(n, 0u);
    for (uint32_t b = 0; b < bandCount; ++b) rowsOf[b % n] += std::min(group, p->h - b * group);
    for (uint32_t k = 0; k < n; ++k) {
        if (rowsOf[k] == 0) continue;
        rt_scene* const s = scenes[k];
        HIP_TRY(hipSetDevice(s->device));
        uint64_t const need = static_cast<uint64_t>(rowsOf[k]) * p->w;
        if (s->wsOutPixels < need) {
            if (s->hasLastStream) HIP_TRY(hipStreamSynchronize(s->lastStream));
            if (s->wsOut) (void)hipFree(s->wsOut);
            s->wsOut = nullptr; s->wsOutPixels = 0;
            HIP_TRY(hipMalloc(&s->wsOut, need * 12));
            s->wsOutPixels = need;
        }
        rt_render_params q = *p;
        q.y0 = p->y0 + k * group;
        q.h = rowsOf[k];
        q.row_group = group;
        q.row_stride = n;
        int const rc = rt_render_device(s, &q, static_cast<float*>(s->wsOut), nullptr, nullptr);
        if (rc != RT_OK) return rc;
        // this scene's bands into the frame, behind its render on its own GPU's stream
        size_t cursor = 0;
        for (uint32_t b = k; b < bandCount; b += n) {
            size_t const bytes = static_cast<size_t>(std::min(group, p->h - b * group)) * rowBytes;
            char* dst = reinterpret_cast<char*>(d_frame) + static_cast<size_t>(b) * group * rowBytes;
            const char* src = static_cast<const char*>(s->wsOut) + cursor;
            if (s->device == first->device) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nullptr));
            else HIP_TRY(hipMemcpyPeerAsync(dst, first->device, src, s->device, bytes, nullptr));
            cursor += bytes;
        }
    }
    // ---- wait for every GPU, then add up what they counted ----
    if (stats) std::memset(stats, 0, sizeof *stats);
    for (uint32_t k = 0; k < n; ++k) {
        if (rowsOf[k] == 0) continue;
        rt_scene* const s = scenes[k];
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipStreamSynchronize(nullptr));
        if (stats) {
            Counters hc;
            HIP_TRY(hipMemcpy(&hc, s->ws.counters, sizeof hc, hipMemcpyDeviceToHost));
            stats->paths += hc.paths; stats->segments += hc.segments; stats->shaded_hits += hc.shadedHits;
            stats->box_tests += hc.boxTests; stats->tris_tested += hc.trisTested; stats->leaf_visits += hc.leafVisits;
            stats->inode_visits += hc.inodeVisits;
            stats->box_exact_fallbacks += hc.boxFallbacks; stats->box_filter_errors += hc.boxFilterErrors;
            stats->trace_launches += s->lastTraceLaunches; stats->batches += s->lastBatches;
        }
    }
    if (host_frame) {
        HIP_TRY(hipSetDevice(first->device));
        HIP_TRY(hipMemcpy(host_frame, d_frame, pixels * 12, hipMemcpyDeviceToHost));
    }
    if (stats) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}
}  // namespace

int rt_render_multi(rt_scene* const* scenes, uint32_t scene_count, const rt_render_params* p, float* out_rgb, rt_stats* stats) {
    return guarded("rt_render_multi", [&] { return renderMulti(scenes, scene_count, p, nullptr, out_rgb, stats); });
}

int rt_render_multi_device(rt_scene* const* scenes, uint32_t scene_count, const rt_render_params* p, float* d_out_rgb, rt_stats* stats) {
    return guarded("rt_render_multi_device", [&] { return renderMulti(scenes, scene_count, p, d_out_rgb, nullptr, stats); });
}

namespace {
int renderHost(rt_scene* s, const rt_render_params* p, float* out_rgb, rt_stats* stats) {
    if (!s || !p || !out_rgb) return fail(RT_ERR_INVALID, "rt_render: null argument");
    HIP_TRY(hipSetDevice(s->device));
    uint64_t const pixels = static_cast<uint64_t>(p->w) * p->h;
    if (s->wsOutPixels < pixels) {
        if (s->hasLastStream) HIP_TRY(hipStreamSynchronize(s->lastStream));
        if (s->wsOut) (void)hipFree(s->wsOut);
        s->wsOut = nullptr;
        s->wsOutPixels = 0;
        HIP_TRY(hipMalloc(&s->wsOut, std::max<uint64_t>(pixels, 1) * 12));
        s->wsOutPixels = pixels;
    }
    rt_stats local;
    int const rc = rt_render_device(s, p, static_cast<float*>(s->wsOut), stats ? stats : &local, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpy(out_rgb, s->wsOut, pixels * 12, hipMemcpyDeviceToHost));
    return RT_OK;
}
}  // namespace

int rt_render(rt_scene* s, const rt_render_params* p, float* out_rgb, rt_stats* stats) {
    return guarded("rt_render", [&] { return renderHost(s, p, out_rgb, stats); });
}

#ifdef RT_DIAG
// Diagnostic build only: the tallies of traceStream since the last reset.  [0..4] wave cycles in refill, node
// walk, leaf fetch, triangle scan, store; [6,7] walk iterations and lanes active in them; [8,9] the same for the
// first triangle pass, [10,11] for the second; [12,13] outer iterations and lanes holding a line.
int rt_diag_read(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_diag), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag), zero, sizeof zero) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// Test hook: evaluates one scalar building block on the GPU for n inputs (see k_selftest).
int rt_selftest_math(int op, const float* a, const float* b, uint32_t n, float* out) {
  return guarded("rt_selftest_math", [&]() -> int {
    if (!a || !b || !out || n == 0) return fail(RT_ERR_INVALID, "rt_selftest_math: null argument");
    uint32_t const outPer = (op == 2 || op == 3) ? 2 : 1;
    float *dA = nullptr, *dB = nullptr, *dO = nullptr;
    hipError_t e = hipMalloc(&dA, n * 4ull);
    if (e == hipSuccess) e = hipMalloc(&dB, n * 4ull);
    if (e == hipSuccess) e = hipMalloc(&dO, n * 4ull * outPer);
    if (e == hipSuccess) e = hipMemcpy(dA, a, n * 4ull, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dB, b, n * 4ull, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, dA, dB, n, dO);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dO, n * 4ull * outPer, hipMemcpyDeviceToHost);
    if (dA) (void)hipFree(dA);
    if (dB) (void)hipFree(dB);
    if (dO) (void)hipFree(dO);
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("rt_selftest_math: ") + hipGetErrorString(e));
    return RT_OK;
  });
}

}  // extern "C"
