// BSP construction with the heavy work on the GPU: the same tree as rt_bsp_build (rt_host.cpp) and therefore as the
// reference's BSPTree constructor (src/bsp.hpp:35-48, 174-285), array for array.
//
// The reference recurses depth-first and tests every triangle of the scene against every node box; rt_bsp_build
// recurses the same way over candidate lists.  Here the tree grows one LEVEL at a time: all (node, candidate triangle)
// pairs of a level are evaluated by one kernel -- the exact triangle/box predicate (src/geometry.hpp:217-428) on the
// node's own box, and the widened bounds test that selects the candidates of the two halves -- three prefix sums turn
// the verdicts into positions, the host reads one count triple per node and decides leaf / split / empty exactly as
// the sequential builder does, and a scatter kernel writes the leaves' triangle lists and the next level's pairs, order
// preserved.  The levels are then walked depth-first on the host, negative side first, which numbers interior nodes
// and leaves in the order of the sequential build.
//
// Compiled with -ffp-contract=off like the rest of the library: the predicate must round exactly as on the host.
#include "rt_bsp_internal.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

int rt_internal_fail(int code, const char* message);

namespace {

#define BSPG_TRY(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return rt_internal_fail(RT_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

constexpr uint32_t NONE = 0xFFFFFFFFu;

// A device array that grows (contents are not kept) and is released on every exit path.
struct DevArray {
    void* p = nullptr;
    size_t cap = 0;
    ~DevArray() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t const want = std::max<size_t>(bytes + bytes / 4, 256);
        hipError_t const e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct V3 { float x, y, z; };
__host__ __device__ inline V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline float min3f(float a, float b, float c) { float m = b < a ? b : a; return c < m ? c : m; }
__device__ inline float max3f(float a, float b, float c) { float m = a < b ? b : a; return m < c ? c : m; }
__device__ inline bool apart(float lo, float hi, float r) { return hi < -r || r < lo; }
__device__ inline bool apart2(float p, float q, float r) { return p < q ? apart(p, q, r) : apart(q, p, r); }

// triIntersectsBox, src/geometry.hpp:217-428: the thirteen separating axes in the reference's order and with its
// expressions (the same text as triBox in rt_host.cpp).
__device__ bool triBoxDevice(V3 a, V3 b, V3 c, const float* box) {
    V3 h;
    h.x = (box[3] - box[0]) / 2.0f;
    {
        float const m = box[0] + h.x;
        a.x -= m; b.x -= m; c.x -= m;
        if (apart(min3f(a.x, b.x, c.x), max3f(a.x, b.x, c.x), h.x)) return false;
    }
    h.y = (box[4] - box[1]) / 2.0f;
    {
        float const m = box[1] + h.y;
        a.y -= m; b.y -= m; c.y -= m;
        if (apart(min3f(a.y, b.y, c.y), max3f(a.y, b.y, c.y), h.y)) return false;
    }
    h.z = (box[5] - box[2]) / 2.0f;
    {
        float const m = box[2] + h.z;
        a.z -= m; b.z -= m; c.z -= m;
        if (apart(min3f(a.z, b.z, c.z), max3f(a.z, b.z, c.z), h.z)) return false;
    }
    V3 const ab = sub3(b, a);
    if (apart2(a.z * b.y - a.y * b.z, c.z * ab.y - c.y * ab.z, fabsf(h.y * ab.z) + fabsf(h.z * ab.y))) return false;
    if (apart2(a.x * b.z - a.z * b.x, c.x * ab.z - c.z * ab.x, fabsf(h.x * ab.z) + fabsf(h.z * ab.x))) return false;
    if (apart2(a.y * b.x - a.x * b.y, c.y * ab.x - c.x * ab.y, fabsf(h.x * ab.y) + fabsf(h.y * ab.x))) return false;
    V3 const ac = sub3(c, a);
    if (apart2(a.z * c.y - a.y * c.z, b.z * ac.y - b.y * ac.z, fabsf(h.y * ac.z) + fabsf(h.z * ac.y))) return false;
    if (apart2(a.x * c.z - a.z * c.x, b.x * ac.z - b.z * ac.x, fabsf(h.x * ac.z) + fabsf(h.z * ac.x))) return false;
    if (apart2(a.y * c.x - a.x * c.y, b.y * ac.x - b.x * ac.y, fabsf(h.x * ac.y) + fabsf(h.y * ac.x))) return false;
    V3 const bc = sub3(c, b);
    if (apart2(a.z * bc.y - a.y * bc.z, b.z * c.y - b.y * c.z, fabsf(h.y * bc.z) + fabsf(h.z * bc.y))) return false;
    if (apart2(a.x * bc.z - a.z * bc.x, b.x * c.z - b.z * c.x, fabsf(h.x * bc.z) + fabsf(h.z * bc.x))) return false;
    if (apart2(a.y * bc.x - a.x * bc.y, b.y * c.x - b.x * c.y, fabsf(h.x * bc.y) + fabsf(h.y * bc.x))) return false;
    V3 const n = {ab.y * ac.z - ac.y * ab.z, ab.z * ac.x - ac.z * ab.x, ab.x * ac.y - ac.x * ab.y};
    float const offset = a.x * n.x + a.y * n.y + a.z * n.z;
    float const reach = fabsf(n.x * h.x) + fabsf(n.y * h.y) + fabsf(n.z * h.z);
    return !apart(offset, offset, reach);
}

// One lane per (node, candidate) pair of the level.  inside: the exact predicate on the node's box; keepNeg / keepPos:
// the triangle's bounds, widened by `slack`, meet the lower / upper half of the box cut at its centre on the node's
// axis (the split of src/bsp.hpp:254-275) -- a superset of what the exact predicate accepts there.  Entry `pairs` of
// each flag array is a zero, so that the prefix sums end with the totals.
__global__ void k_level_eval(const float* __restrict__ tri9, const float* __restrict__ bounds6,
                             const float* __restrict__ nodeBox, const uint8_t* __restrict__ nodeAxis,
                             const uint32_t* __restrict__ cand, const uint32_t* __restrict__ pairNode, uint32_t pairs,
                             float slack, uint32_t* __restrict__ inside, uint32_t* __restrict__ keepNeg,
                             uint32_t* __restrict__ keepPos) {
    uint32_t const p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > pairs) return;
    if (p == pairs) { inside[p] = 0u; keepNeg[p] = 0u; keepPos[p] = 0u; return; }
    uint32_t const g = cand[p], nd = pairNode[p];
    const float* t = tri9 + 9 * static_cast<size_t>(g);
    const float* box = nodeBox + 6 * static_cast<size_t>(nd);
    V3 const a = {t[0], t[1], t[2]}, b = {t[3], t[4], t[5]}, c = {t[6], t[7], t[8]};
    inside[p] = triBoxDevice(a, b, c, box) ? 1u : 0u;
    const float* tb = bounds6 + 6 * static_cast<size_t>(g);
    int const axis = nodeAxis[nd];
    float const m = (box[axis] + box[3 + axis]) / 2.0f;
    bool neg = true, pos = true;
    for (int k = 0; k < 3; ++k) {
        float const lo = box[k], hi = box[3 + k];
        float const negHi = k == axis ? m : hi, posLo = k == axis ? m : lo;
        neg = neg && tb[3 + k] >= lo - slack && tb[k] <= negHi + slack;
        pos = pos && tb[3 + k] >= posLo - slack && tb[k] <= hi + slack;
    }
    keepNeg[p] = neg ? 1u : 0u;
    keepPos[p] = pos ? 1u : 0u;
}

// The prefix sums at the first pair of every node (and at `pairs` for the entry after the last node).
__global__ void k_level_bounds(const uint32_t* __restrict__ begin, uint32_t nodesPlusOne, const uint32_t* __restrict__ sIn,
                               const uint32_t* __restrict__ sNeg, const uint32_t* __restrict__ sPos,
                               uint32_t* __restrict__ bIn, uint32_t* __restrict__ bNeg, uint32_t* __restrict__ bPos) {
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodesPlusOne) return;
    uint32_t const at = begin[i];
    bIn[i] = sIn[at]; bNeg[i] = sNeg[at]; bPos[i] = sPos[at];
}

// dest: per node {leaf list base, negative child's base, positive child's base, negative child's number, positive
// child's number} (NONE where there is none).  Order within a list = order of the pairs = ascending triangle index.
__global__ void k_level_scatter(const uint32_t* __restrict__ cand, const uint32_t* __restrict__ pairNode, uint32_t pairs,
                                const uint32_t* __restrict__ inside, const uint32_t* __restrict__ keepNeg,
                                const uint32_t* __restrict__ keepPos, const uint32_t* __restrict__ sIn,
                                const uint32_t* __restrict__ sNeg, const uint32_t* __restrict__ sPos,
                                const uint32_t* __restrict__ bIn, const uint32_t* __restrict__ bNeg,
                                const uint32_t* __restrict__ bPos, const uint32_t* __restrict__ dest,
                                uint32_t* __restrict__ leafOut, uint32_t* __restrict__ nextCand,
                                uint32_t* __restrict__ nextPairNode) {
    uint32_t const p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    uint32_t const nd = pairNode[p], g = cand[p];
    const uint32_t* d = dest + 5 * static_cast<size_t>(nd);
    if (d[0] != NONE && inside[p]) leafOut[d[0] + (sIn[p] - bIn[nd])] = g;
    if (d[1] != NONE && keepNeg[p]) {
        uint32_t const at = d[1] + (sNeg[p] - bNeg[nd]);
        nextCand[at] = g; nextPairNode[at] = d[3];
    }
    if (d[2] != NONE && keepPos[p]) {
        uint32_t const at = d[2] + (sPos[p] - bPos[nd]);
        nextCand[at] = g; nextPairNode[at] = d[4];
    }
}

// One lane per box: how many triangles of the whole scene pass the exact predicate on it.  Triangles go through LDS a
// tile at a time, every lane of the block tests the same triangle against its own box.
__global__ __launch_bounds__(256) void k_exhaustive_counts(const float* __restrict__ tri9, uint32_t tris,
                                                           const float* __restrict__ boxes6, uint32_t boxes,
                                                           uint32_t* __restrict__ counts) {
    __shared__ float tile[256 * 9];
    uint32_t const b = blockIdx.x * blockDim.x + threadIdx.x;
    float box[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (b < boxes)
        for (int k = 0; k < 6; ++k) box[k] = boxes6[6 * static_cast<size_t>(b) + k];
    uint32_t count = 0;
    for (uint32_t first = 0; first < tris; first += 256u) {
        uint32_t const n = min(256u, tris - first);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < 9u * n; k += 256u) tile[k] = tri9[9 * static_cast<size_t>(first) + k];
        __syncthreads();
        if (b < boxes)
            for (uint32_t j = 0; j < n; ++j) {
                const float* t = tile + 9 * j;
                V3 const a = {t[0], t[1], t[2]}, bb = {t[3], t[4], t[5]}, c = {t[6], t[7], t[8]};
                count += triBoxDevice(a, bb, c, box) ? 1u : 0u;
            }
    }
    if (b < boxes) counts[b] = count;
}

struct LevelNode {
    float box[6];
    uint8_t axis;
    uint8_t kind;                 // 0 empty, 1 leaf, 2 split
    uint32_t negChild, posChild;  // number in the next level, NONE: empty child
    uint32_t leafFirst, leafCount;   // in the level's leaf list
};
struct Level {
    std::vector<LevelNode> nodes;
    std::vector<uint32_t> leafTris;     // scene-wide triangle numbers of the level's leaves, leaf after leaf
};

// Depth-first walk over the levels, negative side first: the numbering of the sequential build.
struct Emitter {
    std::vector<Level> const& levels;
    std::vector<uint32_t> const& triMesh;
    std::vector<uint32_t> const& triTri;
    rt_bsp& out;
    uint32_t maxLeaf;

    int32_t emit(uint32_t depth, uint32_t number) {
        LevelNode const& n = levels[depth].nodes[number];
        if (n.kind == 0) return 0;
        if (n.kind == 1) {
            if (n.leafCount > maxLeaf) ++out.oversized;
            const uint32_t* g = levels[depth].leafTris.data() + n.leafFirst;
            for (uint32_t k = 0; k < n.leafCount; ++k) out.refs.push_back({triMesh[g[k]], triTri[g[k]]});
            out.leaf_first.push_back(static_cast<uint32_t>(out.refs.size()));
            return -static_cast<int32_t>(out.leaf_first.size() - 1);
        }
        out.depth = std::max(out.depth, depth + 1);
        size_t const self = out.inodes.size();
        rt_inode node{};
        node.axis = n.axis;
        std::memcpy(node.neg_box, n.box, sizeof node.neg_box);
        std::memcpy(node.pos_box, n.box, sizeof node.pos_box);
        float const m = (n.box[n.axis] + n.box[3 + n.axis]) / 2.0f;          // src/bsp.hpp:254-275
        node.neg_box[3 + n.axis] = m;
        node.pos_box[n.axis] = m;
        out.inodes.push_back(node);
        int32_t const neg = n.negChild == NONE ? 0 : emit(depth + 1, n.negChild);
        int32_t const pos = n.posChild == NONE ? 0 : emit(depth + 1, n.posChild);
        out.inodes[self].neg_index = neg;
        out.inodes[self].pos_index = pos;
        return static_cast<int32_t>(self + 1);
    }
};

int buildOnDevice(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                  const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                  const float root_box[6], uint32_t maxLeaf, uint32_t maxDepth, int device, rt_bsp& bsp) {
    auto const t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (std::getenv("RT_AMD_DEBUG"))
            std::fprintf(stderr, "rt_amd: rt_bsp_build_device: %s at %.3f s\n", what,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    // ---- triangles, flattened as in rt_bsp_build ------------------------------------------------------
    size_t total = 0;
    for (uint32_t m = 0; m < model_count; ++m) total += mesh_tri_ranges[2 * model_meshes[m] + 1];
    if (total >= (1ull << 31)) return rt_internal_fail(RT_ERR_LIMIT, "rt_bsp_build_device: more than 2^31 triangles");
    std::vector<float> tri9(9 * total), bounds6(6 * total);
    std::vector<uint32_t> triMesh(total), triTri(total);
    float scale = 0.0f;
    for (int k = 0; k < 6; ++k) scale = std::max(scale, std::fabs(root_box[k]));
    size_t g = 0;
    for (uint32_t m = 0; m < model_count; ++m) {
        uint32_t const begin = mesh_tri_ranges[2 * model_meshes[m]], count = mesh_tri_ranges[2 * model_meshes[m] + 1];
        const float* verts = inst_positions + 3 * static_cast<size_t>(model_vertex_begin[m]);
        for (uint32_t k = 0; k < count; ++k, ++g) {
            const uint32_t* idx = base_tris + 3 * static_cast<size_t>(begin + k);
            float* t = tri9.data() + 9 * g;
            for (int v = 0; v < 3; ++v) std::memcpy(t + 3 * v, verts + 3 * static_cast<size_t>(idx[v]), 12);
            float* b = bounds6.data() + 6 * g;
            for (int a = 0; a < 3; ++a) {
                b[a] = std::min({t[a], t[3 + a], t[6 + a]});
                b[3 + a] = std::max({t[a], t[3 + a], t[6 + a]});
                scale = std::max(scale, std::max(std::fabs(b[a]), std::fabs(b[3 + a])));
            }
            triMesh[g] = m;
            triTri[g] = k;
        }
    }
    float const slack = 1e-4f * scale;          // as in rt_bsp_build
    lap("triangles flattened");

    BSPG_TRY(hipSetDevice(device));
    DevArray dTri, dBounds, dNodeBox, dNodeAxis, dBegin, dDest, dCand[2], dPairNode[2], dIn, dNeg, dPos, dSIn, dSNeg, dSPos,
             dBIn, dBNeg, dBPos, dLeafOut, dTemp;
    BSPG_TRY(dTri.reserve(tri9.size() * 4));
    BSPG_TRY(dBounds.reserve(bounds6.size() * 4));
    if (total) {
        BSPG_TRY(hipMemcpy(dTri.p, tri9.data(), tri9.size() * 4, hipMemcpyHostToDevice));
        BSPG_TRY(hipMemcpy(dBounds.p, bounds6.data(), bounds6.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<float>().swap(tri9);
    std::vector<float>().swap(bounds6);

    // ---- level 0: the root with every triangle of the scene as a candidate ---------------------------------
    std::vector<Level> levels(1);
    {
        LevelNode root{};
        std::memcpy(root.box, root_box, sizeof root.box);
        root.axis = 0;
        levels[0].nodes.push_back(root);
    }
    std::vector<uint32_t> begin{0u, static_cast<uint32_t>(total)};        // first pair of every node, then `pairs`
    {
        std::vector<uint32_t> all(total), zeros(total, 0u);
        for (size_t i = 0; i < total; ++i) all[i] = static_cast<uint32_t>(i);
        BSPG_TRY(dCand[0].reserve(total * 4));
        BSPG_TRY(dPairNode[0].reserve(total * 4));
        if (total) {
            BSPG_TRY(hipMemcpy(dCand[0].p, all.data(), total * 4, hipMemcpyHostToDevice));
            BSPG_TRY(hipMemcpy(dPairNode[0].p, zeros.data(), total * 4, hipMemcpyHostToDevice));
        }
    }
    int cur = 0;
    std::vector<uint32_t> bIn, bNeg, bPos, dest;
    std::vector<float> boxes;
    std::vector<uint8_t> axes;
    for (uint32_t depth = 0;; ++depth) {
        Level& L = levels[depth];
        size_t const nodes = L.nodes.size();
        uint64_t const pairs64 = begin.back();
        if (pairs64 >= (1ull << 31)) return rt_internal_fail(RT_ERR_LIMIT, "rt_bsp_build_device: more than 2^31 candidate pairs on one level");
        uint32_t const pairs = static_cast<uint32_t>(pairs64);
        // node boxes and axes of this level
        boxes.resize(6 * nodes);
        axes.resize(nodes);
        for (size_t i = 0; i < nodes; ++i) { std::memcpy(boxes.data() + 6 * i, L.nodes[i].box, 24); axes[i] = L.nodes[i].axis; }
        BSPG_TRY(dNodeBox.reserve(boxes.size() * 4));
        BSPG_TRY(dNodeAxis.reserve(axes.size()));
        BSPG_TRY(dBegin.reserve(begin.size() * 4));
        BSPG_TRY(hipMemcpy(dNodeBox.p, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
        BSPG_TRY(hipMemcpy(dNodeAxis.p, axes.data(), axes.size(), hipMemcpyHostToDevice));
        BSPG_TRY(hipMemcpy(dBegin.p, begin.data(), begin.size() * 4, hipMemcpyHostToDevice));
        size_t const flagBytes = (static_cast<size_t>(pairs) + 1) * 4;
        for (DevArray* a : {&dIn, &dNeg, &dPos, &dSIn, &dSNeg, &dSPos}) BSPG_TRY(a->reserve(flagBytes));
        for (DevArray* a : {&dBIn, &dBNeg, &dBPos}) BSPG_TRY(a->reserve((nodes + 1) * 4));
        uint32_t const evalBlocks = (pairs + 1 + 255) / 256;
        hipLaunchKernelGGL(k_level_eval, dim3(evalBlocks), dim3(256), 0, nullptr, dTri.as<float>(), dBounds.as<float>(),
                           dNodeBox.as<float>(), dNodeAxis.as<uint8_t>(), dCand[cur].as<uint32_t>(), dPairNode[cur].as<uint32_t>(),
                           pairs, slack, dIn.as<uint32_t>(), dNeg.as<uint32_t>(), dPos.as<uint32_t>());
        BSPG_TRY(hipGetLastError());
        size_t tempBytes = 0;
        BSPG_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, dIn.as<uint32_t>(), dSIn.as<uint32_t>(), static_cast<int>(pairs + 1)));
        BSPG_TRY(dTemp.reserve(tempBytes));
        BSPG_TRY(hipcub::DeviceScan::ExclusiveSum(dTemp.p, tempBytes, dIn.as<uint32_t>(), dSIn.as<uint32_t>(), static_cast<int>(pairs + 1)));
        BSPG_TRY(hipcub::DeviceScan::ExclusiveSum(dTemp.p, tempBytes, dNeg.as<uint32_t>(), dSNeg.as<uint32_t>(), static_cast<int>(pairs + 1)));
        BSPG_TRY(hipcub::DeviceScan::ExclusiveSum(dTemp.p, tempBytes, dPos.as<uint32_t>(), dSPos.as<uint32_t>(), static_cast<int>(pairs + 1)));
        uint32_t const boundBlocks = static_cast<uint32_t>((nodes + 1 + 255) / 256);
        hipLaunchKernelGGL(k_level_bounds, dim3(boundBlocks), dim3(256), 0, nullptr, dBegin.as<uint32_t>(),
                           static_cast<uint32_t>(nodes + 1), dSIn.as<uint32_t>(), dSNeg.as<uint32_t>(), dSPos.as<uint32_t>(),
                           dBIn.as<uint32_t>(), dBNeg.as<uint32_t>(), dBPos.as<uint32_t>());
        BSPG_TRY(hipGetLastError());
        bIn.resize(nodes + 1); bNeg.resize(nodes + 1); bPos.resize(nodes + 1);
        BSPG_TRY(hipMemcpy(bIn.data(), dBIn.p, (nodes + 1) * 4, hipMemcpyDeviceToHost));
        BSPG_TRY(hipMemcpy(bNeg.data(), dBNeg.p, (nodes + 1) * 4, hipMemcpyDeviceToHost));
        BSPG_TRY(hipMemcpy(bPos.data(), dBPos.p, (nodes + 1) * 4, hipMemcpyDeviceToHost));

        // ---- the sequential builder's decisions, node by node (Builder::node in rt_host.cpp) -------------------
        bool const mustStop = depth >= maxDepth;
        dest.assign(5 * nodes, NONE);
        Level next;
        std::vector<uint32_t> nextBegin{0u};
        uint64_t leafTotal = 0, nextPairs = 0;
        for (size_t i = 0; i < nodes; ++i) {
            LevelNode& n = L.nodes[i];
            uint32_t const nIn = bIn[i + 1] - bIn[i];
            n.negChild = n.posChild = NONE;
            if (!mustStop && nIn > maxLeaf) {                  // the (maxLeaf+1)-th triangle: subdivide, src/bsp.hpp:206-208
                n.kind = 2;
                float const m = (n.box[n.axis] + n.box[3 + n.axis]) / 2.0f;
                for (int side = 0; side < 2; ++side) {
                    uint32_t const count = side == 0 ? bNeg[i + 1] - bNeg[i] : bPos[i + 1] - bPos[i];
                    if (count == 0) continue;                  // no candidates: the child is empty whatever its box
                    LevelNode child{};
                    std::memcpy(child.box, n.box, sizeof child.box);
                    if (side == 0) child.box[3 + n.axis] = m; else child.box[n.axis] = m;
                    child.axis = static_cast<uint8_t>((n.axis + 1) % 3);
                    uint32_t const number = static_cast<uint32_t>(next.nodes.size());
                    next.nodes.push_back(child);
                    (side == 0 ? n.negChild : n.posChild) = number;
                    dest[5 * i + 1 + side] = static_cast<uint32_t>(nextPairs);
                    dest[5 * i + 3 + side] = number;
                    nextPairs += count;
                    if (nextPairs >= (1ull << 31)) return rt_internal_fail(RT_ERR_LIMIT, "rt_bsp_build_device: more than 2^31 candidate pairs on one level");
                    nextBegin.push_back(static_cast<uint32_t>(nextPairs));
                }
            }
            else if (nIn == 0) n.kind = 0;
            else {
                n.kind = 1;
                n.leafFirst = static_cast<uint32_t>(leafTotal);
                n.leafCount = nIn;
                dest[5 * i] = static_cast<uint32_t>(leafTotal);
                leafTotal += nIn;
            }
        }
        int const nxt = cur ^ 1;
        BSPG_TRY(dDest.reserve(dest.size() * 4));
        BSPG_TRY(hipMemcpy(dDest.p, dest.data(), dest.size() * 4, hipMemcpyHostToDevice));
        BSPG_TRY(dLeafOut.reserve(leafTotal * 4));
        BSPG_TRY(dCand[nxt].reserve(nextPairs * 4));
        BSPG_TRY(dPairNode[nxt].reserve(nextPairs * 4));
        if (pairs) {
            hipLaunchKernelGGL(k_level_scatter, dim3((pairs + 255) / 256), dim3(256), 0, nullptr, dCand[cur].as<uint32_t>(),
                               dPairNode[cur].as<uint32_t>(), pairs, dIn.as<uint32_t>(), dNeg.as<uint32_t>(), dPos.as<uint32_t>(),
                               dSIn.as<uint32_t>(), dSNeg.as<uint32_t>(), dSPos.as<uint32_t>(), dBIn.as<uint32_t>(),
                               dBNeg.as<uint32_t>(), dBPos.as<uint32_t>(), dDest.as<uint32_t>(), dLeafOut.as<uint32_t>(),
                               dCand[nxt].as<uint32_t>(), dPairNode[nxt].as<uint32_t>());
            BSPG_TRY(hipGetLastError());
        }
        L.leafTris.resize(leafTotal);
        if (leafTotal) BSPG_TRY(hipMemcpy(L.leafTris.data(), dLeafOut.p, leafTotal * 4, hipMemcpyDeviceToHost));
        else BSPG_TRY(hipDeviceSynchronize());
        if (next.nodes.empty()) break;
        levels.push_back(std::move(next));
        begin.swap(nextBegin);
        cur = nxt;
    }

    lap("levels built");
    std::memcpy(bsp.root_box, root_box, sizeof bsp.root_box);
    size_t splitNodes = 0, leafNodes = 0, leafRefs = 0;
    for (Level const& L : levels) {
        leafRefs += L.leafTris.size();
        for (LevelNode const& n : L.nodes) { splitNodes += n.kind == 2; leafNodes += n.kind == 1; }
    }
    bsp.inodes.reserve(splitNodes);
    bsp.leaf_first.reserve(leafNodes + 1);
    bsp.refs.reserve(leafRefs);
    bsp.leaf_first.push_back(0);
    Emitter e{levels, triMesh, triTri, bsp, maxLeaf};
    bsp.root_index = e.emit(0, 0);
    lap("numbered depth-first");
    return RT_OK;
}

// Flattens the scene's triangles (9 floats each, model after model) as both builders do.
void flattenTriangles(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                      const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                      std::vector<float>& tri9) {
    size_t total = 0;
    for (uint32_t m = 0; m < model_count; ++m) total += mesh_tri_ranges[2 * model_meshes[m] + 1];
    tri9.resize(9 * total);
    size_t g = 0;
    for (uint32_t m = 0; m < model_count; ++m) {
        uint32_t const begin = mesh_tri_ranges[2 * model_meshes[m]], count = mesh_tri_ranges[2 * model_meshes[m] + 1];
        const float* verts = inst_positions + 3 * static_cast<size_t>(model_vertex_begin[m]);
        for (uint32_t k = 0; k < count; ++k, ++g) {
            const uint32_t* idx = base_tris + 3 * static_cast<size_t>(begin + k);
            for (int v = 0; v < 3; ++v) std::memcpy(tri9.data() + 9 * g + 3 * v, verts + 3 * static_cast<size_t>(idx[v]), 12);
        }
    }
}

int exhaustiveCounts(const std::vector<float>& tri9, const float* boxes6, uint32_t box_count, int device, uint32_t* counts) {
    if (tri9.size() / 9 >= (1ull << 32)) return rt_internal_fail(RT_ERR_LIMIT, "rt_bsp_exhaustive_counts_device: more than 2^32 triangles");
    BSPG_TRY(hipSetDevice(device));
    DevArray dTri, dBoxes, dCounts;
    BSPG_TRY(dTri.reserve(tri9.size() * 4));
    BSPG_TRY(dBoxes.reserve(static_cast<size_t>(box_count) * 24));
    BSPG_TRY(dCounts.reserve(static_cast<size_t>(box_count) * 4));
    if (!tri9.empty()) BSPG_TRY(hipMemcpy(dTri.p, tri9.data(), tri9.size() * 4, hipMemcpyHostToDevice));
    BSPG_TRY(hipMemcpy(dBoxes.p, boxes6, static_cast<size_t>(box_count) * 24, hipMemcpyHostToDevice));
    // in launches of 262 144 boxes (four blocks per CU), so that no single launch runs for minutes on a large scene
    for (uint32_t first = 0; first < box_count; first += 262144u) {
        uint32_t const n = std::min(262144u, box_count - first);
        hipLaunchKernelGGL(k_exhaustive_counts, dim3((n + 255) / 256), dim3(256), 0, nullptr, dTri.as<float>(),
                           static_cast<uint32_t>(tri9.size() / 9), dBoxes.as<float>() + 6 * static_cast<size_t>(first), n,
                           dCounts.as<uint32_t>() + first);
        BSPG_TRY(hipGetLastError());
        BSPG_TRY(hipDeviceSynchronize());
    }
    BSPG_TRY(hipMemcpy(counts, dCounts.p, static_cast<size_t>(box_count) * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // namespace

extern "C" int rt_bsp_exhaustive_counts_device(const float* inst_positions, const uint32_t* model_vertex_begin,
                                               const uint32_t* base_tris, const uint32_t* mesh_tri_ranges,
                                               const uint32_t* model_meshes, uint32_t model_count, const float* boxes6,
                                               uint32_t box_count, int device, uint32_t* counts) {
    if (!inst_positions || !model_vertex_begin || !base_tris || !mesh_tri_ranges || !model_meshes || (box_count && (!boxes6 || !counts)))
        return rt_internal_fail(RT_ERR_INVALID, "rt_bsp_exhaustive_counts_device: null argument");
    if (box_count == 0) return RT_OK;
    try {
        std::vector<float> tri9;
        flattenTriangles(inst_positions, model_vertex_begin, base_tris, mesh_tri_ranges, model_meshes, model_count, tri9);
        return exhaustiveCounts(tri9, boxes6, box_count, device, counts);
    }
    catch (std::bad_alloc const&) { return rt_internal_fail(RT_ERR_NOMEM, "rt_bsp_exhaustive_counts_device: out of host memory"); }
    catch (...) { return rt_internal_fail(RT_ERR_INVALID, "rt_bsp_exhaustive_counts_device: unexpected exception"); }
}

extern "C" int rt_bsp_build_device(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                                   const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                                   const float root_box[6], uint32_t max_leaf_tris, uint32_t max_depth, int device,
                                   rt_bsp** out) {
    if (!inst_positions || !model_vertex_begin || !base_tris || !mesh_tri_ranges || !model_meshes || !root_box || !out
        || max_leaf_tris == 0)
        return rt_internal_fail(RT_ERR_INVALID, "rt_bsp_build_device: null argument or max_leaf_tris == 0");
    if (max_depth > 64) return rt_internal_fail(RT_ERR_LIMIT, "rt_bsp_build_device: max_depth above 64");
    rt_bsp* bsp = new (std::nothrow) rt_bsp;
    if (!bsp) return rt_internal_fail(RT_ERR_NOMEM, "rt_bsp_build_device: out of host memory");
    int rc;
    try {
        rc = buildOnDevice(inst_positions, model_vertex_begin, base_tris, mesh_tri_ranges, model_meshes, model_count, root_box,
                           max_leaf_tris, max_depth, device, *bsp);
    }
    catch (std::bad_alloc const&) { rc = rt_internal_fail(RT_ERR_NOMEM, "rt_bsp_build_device: out of host memory"); }
    catch (...) { rc = rt_internal_fail(RT_ERR_INVALID, "rt_bsp_build_device: unexpected exception"); }
    if (rc != RT_OK) { delete bsp; return rc; }
    *out = bsp;
    return RT_OK;
}
