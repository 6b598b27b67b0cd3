// Host side of rt_scene_create: validates a scene description (BSPTree::_root/_inodes/_leaves, src/bsp.hpp:170-172,
// and RayTraceData, src/render.hpp:28-35) and flattens it into the arrays the traversal and shading kernels read.
// No GPU code here; rt_device.hip uploads the result.
#include "rt_flatten.h"
#include "rt_leaf_bounds.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {

int fail(int code, const std::string& msg) { return rt_internal_fail(code, msg.c_str()); }

// Node record as the kernels read it, 32 bytes: {lo.xyz, hi.x} {hi.yz, index, aux}.
struct HNode { float lo[3], hi[3]; int32_t index; uint32_t aux; };
static_assert(sizeof(HNode) == 32, "node record must be 32 bytes");
// Leaf bounds record, 48 bytes: {lo.xyz, hi.x} {hi.yz, cone word, leaf index} {cK, cE, 0, 0} (rt_leaf_bounds.h).
struct BRec { float lo[3], hi[3]; uint32_t cone; int32_t index; float cK, cE, pad0, pad1; };
static_assert(sizeof(BRec) == 48, "bounds record must be 48 bytes");

// Cone code of a leaf (bits 8-30 of the leaf record's aux word; 0: none), from the normals that head the leaf's triangle
// records (12 floats each).  The axis is the integer point nearest to 31 times the mean unit normal; the threshold is
// derived from that integer axis, with 0.002 rad to spare, and rounded up to an integer.
uint32_t leafCone(const float* records, uint32_t count) {
    double mean[3] = {0.0, 0.0, 0.0};
    uint32_t used = 0;
    for (uint32_t i = 0; i < count; ++i) {
        float const* n = records + 12 * static_cast<size_t>(i);
        double const len = std::sqrt(double(n[0]) * n[0] + double(n[1]) * n[1] + double(n[2]) * n[2]);
        if (!(len > 0.0) || !(len < 1e300)) {
            if (len == 0.0) continue;            // a zero normal fails the facing test whatever the direction
            return 0u;                           // NaN or infinite normal: no cone
        }
        for (int k = 0; k < 3; ++k) mean[k] += n[k] / len;
        ++used;
    }
    double const meanLen = std::sqrt(mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]);
    if (used == 0 || !(meanLen > 1e-6)) return 0u;
    int axis[3];
    for (int k = 0; k < 3; ++k)
        axis[k] = static_cast<int>(std::min(31.0, std::max(-31.0, std::floor(mean[k] / meanLen * 31.0 + 0.5))));
    double const axisLen = std::sqrt(double(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]));
    if (!(axisLen > 0.0)) return 0u;
    double spread = 0.0;
    for (uint32_t i = 0; i < count; ++i) {
        float const* n = records + 12 * static_cast<size_t>(i);
        double const len = std::sqrt(double(n[0]) * n[0] + double(n[1]) * n[1] + double(n[2]) * n[2]);
        if (len == 0.0) continue;
        double const c = (n[0] * double(axis[0]) + n[1] * double(axis[1]) + n[2] * double(axis[2])) / (len * axisLen);
        spread = std::max(spread, std::acos(std::min(1.0, std::max(-1.0, c))));
    }
    double const angle = spread + 0.002;
    if (!(angle <= 1.53)) return 0u;
    double const needed = std::sin(angle) * axisLen * (1.0 + 1e-6);
    uint32_t const q = std::max(1u, static_cast<uint32_t>(std::ceil(needed)));
    if (q > 31u) return 0u;
    return (q << 8) | ((static_cast<uint32_t>(axis[2]) & 0x3Fu) << 13) | ((static_cast<uint32_t>(axis[1]) & 0x3Fu) << 19)
         | ((static_cast<uint32_t>(axis[0]) & 0x3Fu) << 25);
}

float roundedUp(double x) {
    float f = static_cast<float>(x);
    if (double(f) < x) f = std::nextafter(f, INFINITY);
    return f;
}
float roundedDown(double x) {
    float f = static_cast<float>(x);
    if (double(f) > x) f = std::nextafter(f, -INFINITY);
    return f;
}

// The bounds record of one leaf (rt_leaf_bounds.h): the box around the triangles as the triangle test sees them (v1,
// v1 + v1ToV2, v1 + v1ToV3), cut to the leaf's box and padded by 2^-22 of the largest box coordinate, and the two
// coefficients of the widening, from the longest first-corner edge and the flattest first corner among the triangles.
// `unbounded` (measurements): bounds no line misses.
BRec leafBounds(const HNode& node, const float* records, uint32_t count, double absMax, bool unbounded) {
    BRec out{};
    out.cone = node.aux;                  // the leaf's cone word and triangle index: the kernels start the scan from
    out.index = node.index;               // this record alone
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool finite = !unbounded && absMax >= 0x1p-60 && absMax <= 0x1p100;
    double eMax = 0.0, kappaMax = 0.0;
    for (uint32_t i = 0; i < count; ++i) {
        float const* rec = records + 12 * static_cast<size_t>(i);       // normal, v1, v1ToV3, v1ToV2
        double e13 = 0.0, e12 = 0.0, nn = 0.0;
        for (int k = 0; k < 3; ++k) {
            double const a = rec[3 + k], b = a + double(rec[6 + k]), c = a + double(rec[9 + k]);
            if (!std::isfinite(a) || !std::isfinite(b) || !std::isfinite(c) || !std::isfinite(double(rec[k]))) finite = false;
            lo[k] = std::min(lo[k], std::min(a, std::min(b, c)));
            hi[k] = std::max(hi[k], std::max(a, std::max(b, c)));
            e13 += double(rec[6 + k]) * rec[6 + k];
            e12 += double(rec[9 + k]) * rec[9 + k];
            nn += double(rec[k]) * rec[k];
        }
        if (nn == 0.0) continue;            // a zero normal never passes the facing test (its determinant is an exact zero)
        double const e2 = std::max(e13, e12);          // squared length of the longer edge
        eMax = std::max(eMax, std::sqrt(e2));
        kappaMax = std::max(kappaMax, e2 / std::sqrt(nn));
    }
    double const eps = 0x1p-24;
    double const cK = std::sqrt(3.0) * eps * (32.0 * kappaMax + 8.0) * (1.0 + 1e-6);
    double const cE = 2.0 * eMax * cK * (1.0 + 1e-6);
    if (!std::isfinite(cK) || !std::isfinite(cE)) finite = false;
    double const pad = std::ldexp(absMax, -22);
    for (int k = 0; k < 3; ++k) {
        double l = -INFINITY, h = INFINITY;              // non-finite vertices: no bounds
        if (finite) {
            l = std::max(lo[k], double(node.lo[k])) - pad;
            h = std::min(hi[k], double(node.hi[k])) + pad;
        }
        out.lo[k] = roundedDown(l);
        out.hi[k] = roundedUp(h);
    }
    out.cK = finite ? roundedUp(cK) : INFINITY;
    out.cE = finite ? roundedUp(cE) : INFINITY;
    return out;
}

}  // namespace

int rt_internal_gather_tri_normals(const rt_scene_desc* D, const char* who, std::vector<float>& out) {
    out.assign(9 * static_cast<size_t>(std::max<uint32_t>(D->pretri_count, 1)), 0.0f);
    for (uint32_t m = 0; m < D->model_count; ++m) {
        uint32_t const tb = D->model_pretri_begin[m], te = D->model_pretri_begin[m + 1];
        uint32_t const vb = D->model_vertex_begin[m], ve = D->model_vertex_begin[m + 1];
        if (te < tb || ve < vb || te > D->pretri_count || ve > D->vertex_count)
            return fail(RT_ERR_INVALID, std::string(who) + ": offsets not ascending");
        if (static_cast<uint64_t>(D->model_tri_begin[m]) + (te - tb) > D->base_tri_count)
            return fail(RT_ERR_INVALID, std::string(who) + ": model triangle range outside base_tris");
        for (uint32_t k = 0; k < te - tb; ++k) {
            uint32_t const* idx = D->base_tris + 3 * (static_cast<size_t>(D->model_tri_begin[m]) + k);
            float* o = out.data() + 9 * static_cast<size_t>(tb + k);
            for (int j = 0; j < 3; ++j) {
                if (idx[j] >= ve - vb) return fail(RT_ERR_INVALID, std::string(who) + ": vertex index outside model");
                std::memcpy(o + 3 * j, D->vertex_normals + 3 * static_cast<size_t>(vb + idx[j]), 12);
            }
        }
    }
    return RT_OK;
}

void rt_internal_shade_record(float* rec, uint32_t model, const float* normals9) {
    std::memcpy(rec, &model, 4);
    std::memcpy(rec + 1, normals9, 36);
    rec[10] = 0.0f;
    rec[11] = 0.0f;
}

int rt_internal_flatten(const rt_scene_desc* D, rt_flat_scene& F) {
    if (!D) return fail(RT_ERR_INVALID, "rt_scene_create: null argument");
    if (!D->model_pretri_begin || !D->model_vertex_begin || !D->leaf_first_ref)
        return fail(RT_ERR_INVALID, "rt_scene_create: null offset array in description");
    if ((D->model_count && (!D->model_tri_begin || !D->model_materials)) || (D->pretri_count && !D->pretris && !D->leaf_pretris)
        || (D->vertex_count && !D->vertex_normals) || (D->base_tri_count && !D->base_tris))
        return fail(RT_ERR_INVALID, "rt_scene_create: null array in description");
    if (D->inode_count && !D->inodes) return fail(RT_ERR_INVALID, "rt_scene_create: inodes missing");
    uint32_t const refCount = D->leaf_first_ref[D->leaf_count];
    if (refCount && !D->leaf_refs) return fail(RT_ERR_INVALID, "rt_scene_create: leaf_refs missing");
    if (D->model_pretri_begin[D->model_count] != D->pretri_count)
        return fail(RT_ERR_INVALID, "rt_scene_create: model_pretri_begin does not end at pretri_count");
    if (D->model_vertex_begin[D->model_count] != D->vertex_count)
        return fail(RT_ERR_INVALID, "rt_scene_create: model_vertex_begin does not end at vertex_count");

    // ---- flatten the tree -------------------------------------------------------------------
    std::vector<HNode> nodes;
    nodes.reserve(2 * static_cast<size_t>(D->inode_count) + 2);
    auto leafRecord = [&](int32_t refIndex, HNode& n) -> int {
        uint32_t const leaf = static_cast<uint32_t>(-(refIndex + 1));
        if (leaf >= D->leaf_count) return fail(RT_ERR_INVALID, "rt_scene_create: leaf index out of range");
        uint32_t const a = D->leaf_first_ref[leaf], b = D->leaf_first_ref[leaf + 1];
        if (b < a || b > refCount || b == a) return fail(RT_ERR_INVALID, "rt_scene_create: bad leaf range");
        n.index = -static_cast<int32_t>(a) - 1;
        n.aux = b - a;
        return RT_OK;
    };
    uint32_t maxDepth = 0;
    {
        HNode root{};
        std::memcpy(root.lo, D->root_box, 12);
        std::memcpy(root.hi, D->root_box + 3, 12);
        nodes.push_back(root);
        nodes.push_back(HNode{});          // padding: child pairs then start at even slots, one pair per 64-byte line
        struct Item { uint32_t nodeSlot; int32_t refIndex; uint32_t depth; };
        std::vector<Item> todo;
        todo.push_back({0u, D->root_index, 0u});
        std::vector<uint8_t> seen(D->inode_count, 0);
        while (!todo.empty()) {
            Item const it = todo.back();
            todo.pop_back();
            maxDepth = std::max(maxDepth, it.depth);
            if (it.refIndex == 0) { nodes[it.nodeSlot].index = 0; nodes[it.nodeSlot].aux = 0; continue; }
            if (it.refIndex < 0) {
                HNode tmp = nodes[it.nodeSlot];
                int const rc = leafRecord(it.refIndex, tmp);
                if (rc != RT_OK) return rc;
                nodes[it.nodeSlot] = tmp;
                continue;
            }
            uint32_t const inode = static_cast<uint32_t>(it.refIndex - 1);
            if (inode >= D->inode_count) return fail(RT_ERR_INVALID, "rt_scene_create: inode index out of range");
            if (seen[inode]) return fail(RT_ERR_INVALID, "rt_scene_create: inode referenced twice");
            seen[inode] = 1;
            rt_inode const& in = D->inodes[inode];
            if (in.axis > 2) return fail(RT_ERR_INVALID, "rt_scene_create: bad division axis");
            // The kernels recompute the split plane as the centre of the node's own box.
            HNode const self = nodes[it.nodeSlot];
            float const plane = (self.lo[in.axis] + self.hi[in.axis]) / 2.0f;
            if (!(in.pos_box[in.axis] == plane))
                return fail(RT_ERR_INVALID, "rt_scene_create: tree is not split at box centres");
            uint32_t const pair = static_cast<uint32_t>(nodes.size());
            nodes[it.nodeSlot].index = static_cast<int32_t>(pair);
            nodes[it.nodeSlot].aux = in.axis | (in.neg_index == 0 ? 4u : 0u) | (in.pos_index == 0 ? 8u : 0u);
            HNode a{}, b{};
            std::memcpy(a.lo, in.neg_box, 12); std::memcpy(a.hi, in.neg_box + 3, 12);
            std::memcpy(b.lo, in.pos_box, 12); std::memcpy(b.hi, in.pos_box + 3, 12);
            nodes.push_back(a);
            nodes.push_back(b);
            // Depth-first, negative side first, so a subtree's records stay close together.
            todo.push_back({pair + 1, in.pos_index, it.depth + 1});
            todo.push_back({pair, in.neg_index, it.depth + 1});
        }
    }
    if (nodes.size() > 0x7FFFFFF0ull) return fail(RT_ERR_LIMIT, "rt_scene_create: too many nodes");

    // ---- leaf triangle records and ids ----------------------------------------------------------
    // one spare record at the end: the leaf scan fetches the head of the next record while it works on one
    F.leafTris.assign(12 * (static_cast<size_t>(std::max<uint32_t>(refCount, 1)) + 1), 0.0f);
    F.leafIds.assign(4 * static_cast<size_t>(std::max<uint32_t>(refCount, 1)), 0u);
    for (uint32_t r = 0; r < refCount; ++r) {
        rt_tri_ref const ref = D->leaf_refs[r];
        if (ref.mesh >= D->model_count) return fail(RT_ERR_INVALID, "rt_scene_create: leaf ref model out of range");
        uint32_t const b = D->model_pretri_begin[ref.mesh], e = D->model_pretri_begin[ref.mesh + 1];
        if (ref.tri >= e - b) return fail(RT_ERR_INVALID, "rt_scene_create: leaf ref triangle out of range");
        uint32_t const gtri = b + ref.tri;
        float const* pt = D->leaf_pretris ? D->leaf_pretris + 12 * static_cast<size_t>(r)
                                          : D->pretris + 12 * static_cast<size_t>(gtri);    // normal, v1, v1ToV3, v1ToV2
        std::memcpy(F.leafTris.data() + 12 * static_cast<size_t>(r), pt, 48);
        uint32_t const ids[4] = {ref.mesh, ref.tri, gtri, 0u};
        std::memcpy(F.leafIds.data() + 4 * static_cast<size_t>(r), ids, 16);
    }
    // ---- leaf normal cones (see coneAllBackFacing in rt_device.hip) -----------------------------------------
    unsigned long long leavesSeen = 0, leavesWithCone = 0, thresholdSum = 0;
    for (HNode& node : nodes) {
        if (node.index >= 0) continue;
        uint32_t const first = static_cast<uint32_t>(-(node.index + 1)), leafCount = node.aux;
        node.aux = leafCount >= 256u ? (0x80000000u | leafCount) : (leafCount | leafCone(F.leafTris.data() + 12 * static_cast<size_t>(first), leafCount));
        ++leavesSeen;
        if ((node.aux >> 8) != 0u && (node.aux & 0x80000000u) == 0u) { ++leavesWithCone; thresholdSum += (node.aux >> 8) & 0x1Fu; }
    }
    if (std::getenv("RT_AMD_DEBUG"))
        std::fprintf(stderr, "rt_amd: %llu of %llu leaves carry a normal cone, mean threshold %.3f\n", leavesWithCone, leavesSeen,
                     leavesWithCone ? double(thresholdSum) / 31.0 / double(leavesWithCone) : 0.0);
    // ---- node records behind the leaf bounds records (layout: rt_flatten.h) -------------------------------------
    size_t const N = nodes.size();
    size_t const front = (N + 3) & ~static_cast<size_t>(3);
    F.nodeSlots = static_cast<uint32_t>(N);
    F.frontSlots = static_cast<uint32_t>(front);
    F.packed.assign(12 * front + 8 * N, 0.0f);
    double absMax = 0.0;
    for (int k = 0; k < 3; ++k) absMax = std::max(absMax, std::max(std::fabs(double(nodes[0].lo[k])), std::fabs(double(nodes[0].hi[k]))));
    // RT_AMD_LEAF_BOUNDS=0 (measurements): every leaf gets unbounded bounds, which no line misses
    bool const unbounded = std::getenv("RT_AMD_LEAF_BOUNDS") && std::atoi(std::getenv("RT_AMD_LEAF_BOUNDS")) == 0;
    for (size_t slot = 0; slot < N; ++slot) {
        HNode const& node = nodes[slot];
        std::memcpy(F.packed.data() + 12 * front + 8 * slot, &node, 32);
        if (node.index >= 0) continue;
        uint32_t const first = static_cast<uint32_t>(-(node.index + 1));
        uint32_t const leafCount = (node.aux & 0x80000000u) ? (node.aux & 0x7FFFFFFFu) : (node.aux & 0xFFu);
        BRec const rec = leafBounds(node, F.leafTris.data() + 12 * static_cast<size_t>(first), leafCount, absMax, unbounded);
        std::memcpy(F.packed.data() + 12 * (front - 1 - slot), &rec, 48);
    }
    // ---- shading records in the order of the leaf triangle records ------------------------------
    std::vector<float> triNormals;
    {
        int const rc = rt_internal_gather_tri_normals(D, "rt_scene_create", triNormals);
        if (rc != RT_OK) return rc;
    }
    F.shadeRecs.assign(12 * static_cast<size_t>(std::max<uint32_t>(refCount, 1)), 0.0f);
    for (uint32_t r = 0; r < refCount; ++r)
        rt_internal_shade_record(F.shadeRecs.data() + 12 * static_cast<size_t>(r), F.leafIds[4 * static_cast<size_t>(r)],
                                 triNormals.data() + 9 * static_cast<size_t>(F.leafIds[4 * static_cast<size_t>(r) + 2]));
    F.refCount = refCount;
    F.depth = maxDepth;
    F.boxAbsMax = 0.0f;
    for (int k = 0; k < 3; ++k) {
        F.rootLo[k] = nodes[0].lo[k];
        F.rootHi[k] = nodes[0].hi[k];
        F.boxAbsMax = std::max(F.boxAbsMax, std::max(std::fabs(nodes[0].lo[k]), std::fabs(nodes[0].hi[k])));
    }
    return RT_OK;
}

extern "C" {

// Test hooks, host only (declared in rt_amd.h).
int rt_debug_flatten(const rt_scene_desc* D, float* packed, uint64_t* packed_floats, uint32_t* node_slots,
                     uint32_t* front_slots, float* box_abs_max) {
    try {
        if (!D || !packed_floats) return fail(RT_ERR_INVALID, "rt_debug_flatten: null argument");
        rt_flat_scene F;
        int const rc = rt_internal_flatten(D, F);
        if (rc != RT_OK) return rc;
        uint64_t const need = F.packed.size();
        if (node_slots) *node_slots = F.nodeSlots;
        if (front_slots) *front_slots = F.frontSlots;
        if (box_abs_max) *box_abs_max = F.boxAbsMax;
        if (!packed) { *packed_floats = need; return RT_OK; }
        if (*packed_floats < need) { *packed_floats = need; return fail(RT_ERR_LIMIT, "rt_debug_flatten: buffer too small"); }
        std::memcpy(packed, F.packed.data(), need * 4);
        *packed_floats = need;
        return RT_OK;
    }
    catch (std::bad_alloc const&) { return fail(RT_ERR_NOMEM, "rt_debug_flatten: out of host memory"); }
    catch (...) { return fail(RT_ERR_INVALID, "rt_debug_flatten: unexpected exception"); }
}

int rt_debug_leaf_drop(const float* packed, uint32_t front_slots, float box_abs_max, const uint32_t* slots,
                       const float* origins, const float* directions, uint64_t n, uint8_t* out_drop) {
    if (!packed || !slots || !origins || !directions || !out_drop) return fail(RT_ERR_INVALID, "rt_debug_leaf_drop: null argument");
    for (uint64_t i = 0; i < n; ++i) {
        if (slots[i] >= front_slots) return fail(RT_ERR_INVALID, "rt_debug_leaf_drop: slot out of range");
        BRec rec;
        std::memcpy(&rec, packed + 12 * (static_cast<size_t>(front_slots) - 1 - slots[i]), 48);
        float const* o = origins + 3 * i;
        float const* d = directions + 3 * i;
        float const rx = 1.0f / d[0], ry = 1.0f / d[1], rz = 1.0f / d[2];
        bool const rayOk = rtFilterMargin(o[0], o[1], o[2], d[0], d[1], d[2], rx, ry, rz, box_abs_max) < INFINITY;
        out_drop[i] = rtLeafBoundsMiss(rec.lo[0], rec.lo[1], rec.lo[2], rec.hi[0], rec.hi[1], rec.hi[2], rec.cone, rec.cK, rec.cE,
                                       o[0], o[1], o[2], d[0], d[1], d[2], rx, ry, rz, rayOk) ? 1 : 0;
    }
    return RT_OK;
}

}  // extern "C"
