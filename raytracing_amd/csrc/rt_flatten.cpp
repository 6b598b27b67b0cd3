// Host side of rt_scene_create: validates a scene description (BSPTree::_root/_inodes/_leaves, src/bsp.hpp:170-172,
// and RayTraceData, src/render.hpp:28-35) and flattens it into the arrays the traversal and shading kernels read.
// No GPU code here; rt_device.hip uploads the result.
#include "rt_flatten.h"
#include "rt_leaf_bounds.h"

#include <algorithm>
#include <array>
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
// Leaf bounds record (rt_leaf_bounds.h), 64 bytes: {lo.xyz, hi.x} {hi.yz, cone word, leaf index} {cK, cE, cN, n0} {n1 .. n4}
// with the normals' description in n0..n4: {sin b, cos b, A1.xyz} as floats (cone form) or ten 16-bit halves (axes form).
struct BRec { float lo[3], hi[3]; uint32_t cone; int32_t index; float cK, cE, cN; uint32_t n[5]; };
static_assert(sizeof(BRec) == 64, "bounds record must be 64 bytes");
// What leafBounds works out for a leaf, before it is laid out for one of the two record forms.
struct LeafDescription {
    BRec base{};                          // box, words and coefficients; the normals' fields are filled by layOut
    bool finite = false;
    int axisCount = 0;                    // 0: no axes form; else the axes form with this many axes
    float axes[RT_LEAF_AXES_PACKED][3] = {};
    double axesSpread = 4.0;
    bool hasCone = false;                 // one cone around all normals
    float coneAxis[3] = {0, 0, 0};
    double coneSpread = 4.0;
};

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

// Largest angle between a unit vector of `units` and the nearest of +-axes[0..count) (axes as stored, any length).
double spreadAbout(const std::vector<std::array<double, 3>>& units, const float (*axes)[3], int count) {
    double spread = 0.0;
    for (auto const& u : units) {
        double best = -1.0;
        for (int a = 0; a < count; ++a) {
            double const len = std::sqrt(double(axes[a][0]) * axes[a][0] + double(axes[a][1]) * axes[a][1] + double(axes[a][2]) * axes[a][2]);
            if (!(len > 0.0)) continue;
            best = std::max(best, std::fabs(u[0] * axes[a][0] + u[1] * axes[a][1] + u[2] * axes[a][2]) / len);
        }
        if (best < 0.0) return 4.0;
        spread = std::max(spread, std::acos(std::min(1.0, best)));
    }
    return spread;
}

// What the bounds record of one leaf is made from (rt_leaf_bounds.h): the box around the triangles as the triangle test
// sees them (v1, v1 + v1ToV2, v1 + v1ToV3), cut to the leaf's box and padded by 2^-22 of the largest box coordinate; the
// coefficients of the widening, from the longest first-corner edge, the flattest first corner and the longest normal
// among the triangles; and two descriptions of the normals: axes found greedily (a normal joins the axis it lies
// within 0.12 rad of, up to sign, else founds a new one; at most three), refined to the mean of their members,
// and one cone around the mean of all normals (signs folded onto the first).
// `unbounded` (measurements): bounds no line misses.
LeafDescription describeLeaf(const HNode& node, const float* records, uint32_t count, double absMax, bool unbounded) {
    LeafDescription L;
    BRec& out = L.base;
    out.cone = node.aux;                  // the leaf's cone word and triangle index: the kernels start the scan from
    out.index = node.index;               // this record alone
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool finite = !unbounded && absMax >= 0x1p-60 && absMax <= 0x1p100;
    double eMax = 0.0, kappaMax = 0.0, nMax = 0.0;
    std::vector<std::array<double, 3>> units;
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
        // The rule's proof takes the stored normal for the rounded cross product of the stored edges (either order), as
        // preprocessTri makes it (src/geometry.hpp:77-82).  A caller's record that is anything else switches the rule off
        // for its leaf.
        {
            double const a[3] = {rec[9], rec[10], rec[11]}, b[3] = {rec[6], rec[7], rec[8]};          // v1ToV2, v1ToV3
            bool asMade = true, reversed = true;
            for (int k = 0; k < 3; ++k) {
                double const p = a[(k + 1) % 3] * b[(k + 2) % 3], q = a[(k + 2) % 3] * b[(k + 1) % 3];
                double const slack = 2.0 * 0x1p-24 * (std::fabs(p) + std::fabs(q)) * (1.0 + 1e-6) + 0x1p-140;
                asMade = asMade && std::fabs(double(rec[k]) - (p - q)) <= slack;
                reversed = reversed && std::fabs(double(rec[k]) + (p - q)) <= slack;
            }
            if (!asMade && !reversed) finite = false;
        }
        double const e2 = std::max(e13, e12);          // squared length of the longer edge
        double const len = std::sqrt(nn);
        eMax = std::max(eMax, std::sqrt(e2));
        kappaMax = std::max(kappaMax, e2 / len);
        nMax = std::max(nMax, len);
        units.push_back({rec[0] / len, rec[1] / len, rec[2] / len});
    }
    double const eps = 0x1p-24;
    double const cK = std::sqrt(3.0) * eps * (32.0 * kappaMax + 8.0) * (1.0 + 1e-6);
    double const cE = 2.0 * eMax * cK * (1.0 + 1e-6);
    if (!std::isfinite(cK) || !std::isfinite(cE) || !(nMax > 0.0)) finite = false;
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
    out.cK = finite ? roundedUp(cK) : INFINITY;           // an infinite cK switches the rule off for the leaf
    out.cE = finite ? roundedUp(cE) : INFINITY;
    out.cN = finite ? roundedDown(0.999e-6 / nMax) : 0.0f;
    L.finite = finite;
    if (!finite || units.empty() || count >= 4096) return L;
    // ---- the normals ----
    double sums[RT_LEAF_AXES_PACKED][3] = {}, all[3] = {0, 0, 0};
    int found = 0;
    bool fits = true;
    double const joinCos = std::cos(0.12);
    for (auto const& u : units) {
        int best = -1;
        double bestDot = 0.0;
        for (int a = 0; a < found; ++a) {
            double const len = std::sqrt(sums[a][0] * sums[a][0] + sums[a][1] * sums[a][1] + sums[a][2] * sums[a][2]);
            double const dot = (u[0] * sums[a][0] + u[1] * sums[a][1] + u[2] * sums[a][2]) / len;
            if (std::fabs(dot) > std::fabs(bestDot)) { bestDot = dot; best = a; }
        }
        if (best >= 0 && std::fabs(bestDot) >= joinCos) {
            double const sign = bestDot < 0.0 ? -1.0 : 1.0;
            for (int k = 0; k < 3; ++k) sums[best][k] += sign * u[k];
        }
        else if (found < RT_LEAF_AXES_PACKED) { for (int k = 0; k < 3; ++k) sums[found][k] = u[k]; ++found; }
        else fits = false;
        double const signAll = (u[0] * units[0][0] + u[1] * units[0][1] + u[2] * units[0][2]) < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < 3; ++k) all[k] += signAll * u[k];
    }
    auto store = [](float* dst, const double* v) -> bool {
        double const len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (!(len > 1e-9)) return false;
        for (int k = 0; k < 3; ++k) dst[k] = static_cast<float>(v[k] / len);
        return true;
    };
    if (fits) {
        bool ok = true;
        for (int a = 0; a < found; ++a) ok = ok && store(L.axes[a], sums[a]);
        if (ok) {
            L.axesSpread = spreadAbout(units, L.axes, found) + 1e-6;
            L.axisCount = found;
            for (int a = found; a < RT_LEAF_AXES_PACKED; ++a) for (int k = 0; k < 3; ++k) L.axes[a][k] = L.axes[found - 1][k];
        }
    }
    float cone[1][3];
    if (store(cone[0], all)) {
        L.coneSpread = spreadAbout(units, cone, 1) + 1e-6;
        L.hasCone = L.coneSpread < 1.57;
        for (int k = 0; k < 3; ++k) L.coneAxis[k] = cone[0][k];
    }
    return L;
}

// Is the leaf one of several flat faces (what the axes form is for)?  Two or three axes, every normal within 0.01 rad of
// one of them, and no cone that narrow around all of them.
bool prefersAxes(const LeafDescription& L) {
    return L.axisCount >= 2 && L.axesSpread < 0.01 && !(L.hasCone && L.coneSpread < 0.02);
}

// Fills the normals' words of the record for the cone form (packed == false) or the axes form.
BRec layOut(const LeafDescription& L, bool packed) {
    BRec r = L.base;
    auto sine = [](double b) { return roundedUp(std::min(1.0, std::sin(b) * (1.0 + 0x1p-18) + 0x1p-20)); };
    auto cosine = [](double b) { return roundedDown(std::max(0.0, std::cos(b) * (1.0 - 0x1p-18))); };
    if (!packed) {
        float f[5] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (L.finite && L.hasCone) { f[0] = sine(L.coneSpread); f[1] = cosine(L.coneSpread); std::memcpy(f + 2, L.coneAxis, 12); }
        std::memcpy(r.n, f, 20);
        return r;
    }
    // axes form: the leaf's own axes, or its cone's axis three times with the cone's half-angle
    float axes[3][3] = {};
    double spread = 4.0;
    if (L.finite && L.axisCount >= 1 && (prefersAxes(L) || (L.axesSpread <= L.coneSpread))) { std::memcpy(axes, L.axes, sizeof axes); spread = L.axesSpread; }
    else if (L.finite && L.hasCone) { for (auto& axis : axes) std::memcpy(axis, L.coneAxis, 12); spread = L.coneSpread; }
    uint16_t h[10] = {};
    double const sinB2 = spread < 1.57 ? std::sin(spread + 0x1p-14) * (1.0 + 0x1p-18) + 0x1p-20 : 1.0;
    h[0] = static_cast<uint16_t>(std::min(65535.0, std::ceil(std::min(sinB2, 1.0) * 65536.0)));     // 65535: 2 sin b2 just below 2, no bound
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < 3; ++k)
            h[1 + 3 * a + k] = static_cast<uint16_t>(static_cast<int16_t>(std::lrint(std::max(-1.0, std::min(1.0, double(axes[a][k]))) * 32767.0)));
    for (int w = 0; w < 5; ++w) r.n[w] = static_cast<uint32_t>(h[2 * w]) | (static_cast<uint32_t>(h[2 * w + 1]) << 16);
    return r;
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
    double absMax = 0.0;
    for (int k = 0; k < 3; ++k) absMax = std::max(absMax, std::max(std::fabs(double(nodes[0].lo[k])), std::fabs(double(nodes[0].hi[k]))));
    // RT_AMD_LEAF_BOUNDS=0 (measurements): every leaf gets unbounded bounds, which no line misses
    bool const unbounded = std::getenv("RT_AMD_LEAF_BOUNDS") && std::atoi(std::getenv("RT_AMD_LEAF_BOUNDS")) == 0;
    std::vector<LeafDescription> described;
    described.reserve(leavesSeen);
    size_t wantAxes = 0;
    for (size_t slot = 0; slot < N; ++slot) {
        HNode const& node = nodes[slot];
        if (node.index >= 0) continue;
        uint32_t const first = static_cast<uint32_t>(-(node.index + 1));
        uint32_t const leafCount = (node.aux & 0x80000000u) ? (node.aux & 0x7FFFFFFFu) : (node.aux & 0xFFu);
        described.push_back(describeLeaf(node, F.leafTris.data() + 12 * static_cast<size_t>(first), leafCount, absMax, unbounded));
        if (described.back().finite && prefersAxes(described.back())) ++wantAxes;
    }
    // Scenes in which at least one leaf in ten holds several flat faces (boxes, rooms) get the axes form in all their
    // records; the others (smooth surfaces) the cone form, whose angle sum is the sharper bound for a curved patch.
    // RT_AMD_LEAF_AXES=1 / 3 forces the form (measurements).
    bool many = wantAxes * 10 >= described.size() && wantAxes > 0;
    if (const char* text = std::getenv("RT_AMD_LEAF_AXES")) many = std::atoi(text) > 1;
    F.leafAxes = many ? RT_LEAF_AXES_PACKED : RT_LEAF_AXES_ONE;
    size_t const recFloats = 16;
    F.nodeSlots = static_cast<uint32_t>(N);
    F.frontSlots = static_cast<uint32_t>(N);
    F.packed.assign(recFloats * N + 8 * N, 0.0f);
    size_t leaf = 0;
    for (size_t slot = 0; slot < N; ++slot) {
        HNode const& node = nodes[slot];
        std::memcpy(F.packed.data() + recFloats * N + 8 * slot, &node, 32);
        if (node.index >= 0) continue;
        BRec const rec = layOut(described[leaf++], many);
        std::memcpy(F.packed.data() + recFloats * (N - 1 - slot), &rec, 4 * recFloats);
    }
    if (std::getenv("RT_AMD_DEBUG"))
        std::fprintf(stderr, "rt_amd: %zu of %zu leaves hold several flat faces: bounds records in the %s form\n",
                     wantAxes, described.size(), many ? "axes" : "cone");
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
                     uint32_t* leaf_axes, float* box_abs_max) {
    try {
        if (!D || !packed_floats) return fail(RT_ERR_INVALID, "rt_debug_flatten: null argument");
        rt_flat_scene F;
        int const rc = rt_internal_flatten(D, F);
        if (rc != RT_OK) return rc;
        uint64_t const need = F.packed.size();
        if (node_slots) *node_slots = F.nodeSlots;
        if (leaf_axes) *leaf_axes = F.leafAxes;
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

int rt_debug_leaf_drop(const float* packed, uint32_t node_slots, uint32_t leaf_axes, float box_abs_max, const uint32_t* slots,
                       const float* origins, const float* directions, uint64_t n, uint8_t* out_drop) {
    if (!packed || !slots || !origins || !directions || !out_drop) return fail(RT_ERR_INVALID, "rt_debug_leaf_drop: null argument");
    if (leaf_axes != RT_LEAF_AXES_ONE && leaf_axes != RT_LEAF_AXES_PACKED) return fail(RT_ERR_INVALID, "rt_debug_leaf_drop: bad record form");
    size_t const recFloats = 16;
    for (uint64_t i = 0; i < n; ++i) {
        if (slots[i] >= node_slots) return fail(RT_ERR_INVALID, "rt_debug_leaf_drop: slot out of range");
        float const* rec = packed + recFloats * (static_cast<size_t>(node_slots) - 1 - slots[i]);
        float const* o = origins + 3 * i;
        float const* d = directions + 3 * i;
        float const rx = 1.0f / d[0], ry = 1.0f / d[1], rz = 1.0f / d[2];
        bool const rayOk = rtFilterMargin(o[0], o[1], o[2], d[0], d[1], d[2], rx, ry, rz, box_abs_max) < INFINITY;
        out_drop[i] = rtLeafRecordMiss(rec, static_cast<int>(leaf_axes), o[0], o[1], o[2], d[0], d[1], d[2], rx, ry, rz, rayOk) ? 1 : 0;
    }
    return RT_OK;
}

// Would the traversal skip a leaf whose aux word is cone_words[i] for direction i, on the strength of its normal cone?
int rt_debug_cone_skip(const uint32_t* cone_words, const float* directions, uint64_t n, uint8_t* out_skip) {
    if (!cone_words || !directions || !out_skip) return fail(RT_ERR_INVALID, "rt_debug_cone_skip: null argument");
    for (uint64_t i = 0; i < n; ++i)
        out_skip[i] = rtConeAllBackFacing(cone_words[i], directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]) ? 1 : 0;
    return RT_OK;
}

}  // extern "C"
