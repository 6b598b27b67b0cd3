// Host-side scene preprocessing and BSP construction for librt_amd.so (declared in include/rt_host.h).
//
// The arithmetic of every step matches the reference's so that a scene prepared here is
// bit-identical to one prepared by the reference (src/main.cpp:191-216).  That requires this file to
// be compiled with -ffp-contract=off: the reference's scalar code never fuses a multiply with an add.
#include "rt_host.h"
#include "rt_bsp_internal.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

// Sets the calling thread's error message (rt_last_error) and returns `code`; defined beside the device code.
int rt_internal_fail(int code, const char* message);

namespace {

struct Vec3 {
    float x, y, z;
    float operator[](int k) const { return k == 0 ? x : (k == 1 ? y : z); }
};
inline Vec3 load3(const float* p) { return {p[0], p[1], p[2]}; }
inline void store3(float* p, Vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// Same operation order as glm's scalar code paths (libraries/glm/glm/detail/func_geometric.inl).
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline Vec3 normalize(Vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

// q = {w,x,y,z}; rotation of v as glm computes it (libraries/glm/glm/detail/type_quat.inl:343-350).
Vec3 rotate(const float q[4], Vec3 v) {
    Vec3 const u{q[1], q[2], q[3]};
    Vec3 const uv = cross(u, v);
    Vec3 const uuv = cross(u, uv);
    return v + (uv * q[0] + uuv) * 2.0f;
}

struct Box { Vec3 lo, hi; };

}  // namespace

extern "C" {

void rt_preprocess_material(const float m[8], float out[12]) {
    // src/material.hpp:32-45
    float const roughness = m[3], metalness = m[4];
    float r4 = 1.0f;
    for (int i = 0; i < 4; ++i) r4 *= roughness;
    float const dielectric = 1.0f - metalness;
    float const pi = 3.14159265358979323846264338327950288f;
    out[0] = r4;
    out[1] = r4 / 4.0f;
    for (int c = 0; c < 3; ++c) {
        out[2 + c] = dielectric * 0.04f + metalness * m[c];
        out[5 + c] = dielectric * m[c] / pi;
        out[8 + c] = m[5 + c];
    }
    out[11] = 0.0f;
}

float rt_srgb_to_linear(float srgb) {
    // src/image.hpp:34-41
    return srgb <= 0.04045f ? srgb / 12.92f : std::pow((srgb + 0.055f) / 1.055f, 2.4f);
}

float rt_linear_to_srgb(float linear) {
    // src/image.hpp:20-27
    return linear <= 0.0031308f ? linear * 12.92f : 1.055f * std::pow(linear, 1.0f / 2.4f) - 0.055f;
}

int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    // src/main.cpp:272-280
    if (!path || (!rgb8 && width && height)) return -1;
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return -1;
    size_t const bytes = static_cast<size_t>(width) * height * 3;
    bool ok = std::fprintf(f, "P6\n%u %u\n255\n", width, height) > 0;
    ok = ok && (bytes == 0 || std::fwrite(rgb8, 1, bytes, f) == bytes);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}

void rt_quat_from_euler(const float e[3], float q[4]) {
    // libraries/glm/glm/detail/type_quat.inl:204-213
    float const cx = std::cos(e[0] * 0.5f), cy = std::cos(e[1] * 0.5f), cz = std::cos(e[2] * 0.5f);
    float const sx = std::sin(e[0] * 0.5f), sy = std::sin(e[1] * 0.5f), sz = std::sin(e[2] * 0.5f);
    q[0] = cx * cy * cz + sx * sy * sz;
    q[1] = sx * cy * cz - cx * sy * sz;
    q[2] = cx * sy * cz + sx * cy * sz;
    q[3] = cx * cy * sz - sx * sy * cz;
}

void rt_camera_basis(const float q[4], float forward[3], float down[3], float right[3]) {
    store3(forward, rotate(q, {0.0f, 0.0f, 1.0f}));
    store3(down, rotate(q, {0.0f, -1.0f, 0.0f}));
    store3(right, rotate(q, {-1.0f, 0.0f, 0.0f}));
}

void rt_pixel_to_ray_transform(const float f[3], const float d[3], const float r[3], float fov, uint32_t width,
                               uint32_t height, float out[9]) {
    // src/camera.hpp:40-45: the offset column is evaluated in double and narrowed once.
    double const focal = width / std::tan(fov / 2.0);
    for (int c = 0; c < 3; ++c) {
        out[c] = r[c];
        out[3 + c] = d[c];
        double const o = (focal * static_cast<double>(f[c]) - static_cast<double>(width) * static_cast<double>(r[c])
                          - static_cast<double>(height) * static_cast<double>(d[c])) / 2.0;
        out[6 + c] = static_cast<float>(o);
    }
}

void rt_instantiate_meshes(const float* positions, const float* normals, const uint32_t* mesh_vertex_ranges,
                           const float* transforms, const uint32_t* model_meshes, uint32_t model_count,
                           float* out_positions, float* out_normals, uint32_t* out_model_vertex_begin) {
    uint32_t cursor = 0;
    for (uint32_t model = 0; model < model_count; ++model) {
        float const* t = transforms + 10 * static_cast<size_t>(model);
        // MeshTransform::matrix() (src/mesh.hpp:62-69) over glm::mat3_cast (gtc/quaternion.inl:41-66).
        float const w = t[3], x = t[4], y = t[5], z = t[6];
        float const xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z;
        float const wx = w * x, wy = w * y, wz = w * z;
        Vec3 const c0 = Vec3{1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy)} * t[7];
        Vec3 const c1 = Vec3{2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx)} * t[8];
        Vec3 const c2 = Vec3{2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy)} * t[9];
        Vec3 const c3{t[0], t[1], t[2]};
        // Normal matrix = inverse transpose of the 3x3 part (src/mesh.hpp:74-76;
        // libraries/glm/glm/gtc/matrix_inverse.inl:43-63): cofactors divided by the determinant.
        float const det = c0.x * (c1.y * c2.z - c1.z * c2.y) - c0.y * (c1.x * c2.z - c1.z * c2.x)
                        + c0.z * (c1.x * c2.y - c1.y * c2.x);
        Vec3 n0{+(c1.y * c2.z - c2.y * c1.z), -(c1.x * c2.z - c2.x * c1.z), +(c1.x * c2.y - c2.x * c1.y)};
        Vec3 n1{-(c0.y * c2.z - c2.y * c0.z), +(c0.x * c2.z - c2.x * c0.z), -(c0.x * c2.y - c2.x * c0.y)};
        Vec3 n2{+(c0.y * c1.z - c1.y * c0.z), -(c0.x * c1.z - c1.x * c0.z), +(c0.x * c1.y - c1.x * c0.y)};
        n0 = {n0.x / det, n0.y / det, n0.z / det};
        n1 = {n1.x / det, n1.y / det, n1.z / det};
        n2 = {n2.x / det, n2.y / det, n2.z / det};

        uint32_t const begin = mesh_vertex_ranges[2 * model_meshes[model]];
        uint32_t const count = mesh_vertex_ranges[2 * model_meshes[model] + 1];
        out_model_vertex_begin[model] = cursor;
        for (uint32_t k = 0; k < count; ++k) {
            Vec3 const p = load3(positions + 3 * static_cast<size_t>(begin + k));
            Vec3 const n = load3(normals + 3 * static_cast<size_t>(begin + k));
            // mat4x3 * vec4{p, 1} and mat3 * n, summed left to right as glm does.
            Vec3 const wp{c0.x * p.x + c1.x * p.y + c2.x * p.z + c3.x * 1.0f,
                          c0.y * p.x + c1.y * p.y + c2.y * p.z + c3.y * 1.0f,
                          c0.z * p.x + c1.z * p.y + c2.z * p.z + c3.z * 1.0f};
            Vec3 const wn{n0.x * n.x + n1.x * n.y + n2.x * n.z,
                          n0.y * n.x + n1.y * n.y + n2.y * n.z,
                          n0.z * n.x + n1.z * n.y + n2.z * n.z};
            store3(out_positions + 3 * static_cast<size_t>(cursor + k), wp);
            store3(out_normals + 3 * static_cast<size_t>(cursor + k), normalize(wn));
        }
        cursor += count;
    }
    out_model_vertex_begin[model_count] = cursor;
}

void rt_preprocess_tris(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                        const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                        float* out_pretris, uint32_t* out_model_pretri_begin) {
    uint32_t cursor = 0;
    for (uint32_t model = 0; model < model_count; ++model) {
        uint32_t const begin = mesh_tri_ranges[2 * model_meshes[model]];
        uint32_t const count = mesh_tri_ranges[2 * model_meshes[model] + 1];
        float const* verts = inst_positions + 3 * static_cast<size_t>(model_vertex_begin[model]);
        out_model_pretri_begin[model] = cursor;
        for (uint32_t k = 0; k < count; ++k) {
            uint32_t const* idx = base_tris + 3 * static_cast<size_t>(begin + k);
            Vec3 const v1 = load3(verts + 3 * static_cast<size_t>(idx[0]));
            Vec3 const e12 = load3(verts + 3 * static_cast<size_t>(idx[1])) - v1;
            Vec3 const e13 = load3(verts + 3 * static_cast<size_t>(idx[2])) - v1;
            float* o = out_pretris + 12 * static_cast<size_t>(cursor + k);
            store3(o, cross(e12, e13));     // normal
            store3(o + 3, v1);
            store3(o + 6, e13);             // v1ToV3
            store3(o + 9, e12);             // v1ToV2
        }
        cursor += count;
    }
    out_model_pretri_begin[model_count] = cursor;
}

void rt_scene_bounding_box(const float* p, uint64_t vertex_count, int inflate_mode, float out_box[6]) {
    float const inf = std::numeric_limits<float>::infinity();
    Vec3 lo{inf, inf, inf}, hi{-inf, -inf, -inf};
    for (uint64_t i = 0; i < vertex_count; ++i) {
        Vec3 const v = load3(p + 3 * i);
        lo = {v.x < lo.x ? v.x : lo.x, v.y < lo.y ? v.y : lo.y, v.z < lo.z ? v.z : lo.z};
        hi = {hi.x < v.x ? v.x : hi.x, hi.y < v.y ? v.y : hi.y, hi.z < v.z ? v.z : hi.z};
    }
    if (inflate_mode == 0) {
        lo = lo * 1.001f;
        hi = hi * 1.001f;
    }
    else {
        Vec3 const e = hi - lo;
        float const pad = 0.001f * std::max(e.x, std::max(e.y, e.z));
        lo = lo - Vec3{pad, pad, pad};
        hi = hi + Vec3{pad, pad, pad};
    }
    store3(out_box, lo);
    store3(out_box + 3, hi);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Predicates
// ------------------------------------------------------------------------------------------------
namespace {

// One face of lineIntersectsBox (src/geometry.hpp:136-212): the line meets the face plane at
// t = numer/denom with |denom| >= 1e-6 and t >= 0, and the hit lies inside the face rectangle
// spanned by the two other axes a and b (bounds inclusive, point = origin + t*direction unfused).
inline bool faceHit(float denom, float numer, float oa, float da, float loa, float hia, float ob, float db, float lob,
                    float hib) {
    if (std::fabs(denom) < 1e-6f) return false;
    float const t = numer / denom;
    if (!(t >= 0.0f)) return false;
    float const pa = oa + t * da;
    if (!(pa >= loa && pa <= hia)) return false;
    float const pb = ob + t * db;
    return pb >= lob && pb <= hib;
}

bool lineBox(Vec3 o, Vec3 d, Box const& b) {
    return faceHit(-d.x, o.x - b.lo.x, o.y, d.y, b.lo.y, b.hi.y, o.z, d.z, b.lo.z, b.hi.z)
        || faceHit(d.x, b.hi.x - o.x, o.y, d.y, b.lo.y, b.hi.y, o.z, d.z, b.lo.z, b.hi.z)
        || faceHit(-d.y, o.y - b.lo.y, o.x, d.x, b.lo.x, b.hi.x, o.z, d.z, b.lo.z, b.hi.z)
        || faceHit(d.y, b.hi.y - o.y, o.x, d.x, b.lo.x, b.hi.x, o.z, d.z, b.lo.z, b.hi.z)
        || faceHit(-d.z, o.z - b.lo.z, o.x, d.x, b.lo.x, b.hi.x, o.y, d.y, b.lo.y, b.hi.y)
        || faceHit(d.z, b.hi.z - o.z, o.x, d.x, b.lo.x, b.hi.x, o.y, d.y, b.lo.y, b.hi.y);
}

// Separating-axis test of src/geometry.hpp:217-428.  The triangle is translated so the box centre is
// the origin; `h` is the box half-extent.  For the nine edge axes only two of the three vertex
// projections are distinct, and the projection expressions differ per edge exactly as in the
// reference, so they are spelled out rather than generated.
inline bool apart(float lo, float hi, float r) { return hi < -r || r < lo; }
inline bool apart2(float p, float q, float r) { return p < q ? apart(p, q, r) : apart(q, p, r); }

bool triBox(Vec3 a, Vec3 b, Vec3 c, Box const& box) {
    Vec3 h;
    h.x = (box.hi.x - box.lo.x) / 2.0f;
    {
        float const m = box.lo.x + h.x;
        a.x -= m; b.x -= m; c.x -= m;
        if (apart(std::min({a.x, b.x, c.x}), std::max({a.x, b.x, c.x}), h.x)) return false;
    }
    h.y = (box.hi.y - box.lo.y) / 2.0f;
    {
        float const m = box.lo.y + h.y;
        a.y -= m; b.y -= m; c.y -= m;
        if (apart(std::min({a.y, b.y, c.y}), std::max({a.y, b.y, c.y}), h.y)) return false;
    }
    h.z = (box.hi.z - box.lo.z) / 2.0f;
    {
        float const m = box.lo.z + h.z;
        a.z -= m; b.z -= m; c.z -= m;
        if (apart(std::min({a.z, b.z, c.z}), std::max({a.z, b.z, c.z}), h.z)) return false;
    }
    Vec3 const ab = b - a;
    if (apart2(a.z * b.y - a.y * b.z, c.z * ab.y - c.y * ab.z, std::fabs(h.y * ab.z) + std::fabs(h.z * ab.y))) return false;
    if (apart2(a.x * b.z - a.z * b.x, c.x * ab.z - c.z * ab.x, std::fabs(h.x * ab.z) + std::fabs(h.z * ab.x))) return false;
    if (apart2(a.y * b.x - a.x * b.y, c.y * ab.x - c.x * ab.y, std::fabs(h.x * ab.y) + std::fabs(h.y * ab.x))) return false;
    Vec3 const ac = c - a;
    if (apart2(a.z * c.y - a.y * c.z, b.z * ac.y - b.y * ac.z, std::fabs(h.y * ac.z) + std::fabs(h.z * ac.y))) return false;
    if (apart2(a.x * c.z - a.z * c.x, b.x * ac.z - b.z * ac.x, std::fabs(h.x * ac.z) + std::fabs(h.z * ac.x))) return false;
    if (apart2(a.y * c.x - a.x * c.y, b.y * ac.x - b.x * ac.y, std::fabs(h.x * ac.y) + std::fabs(h.y * ac.x))) return false;
    Vec3 const bc = c - b;
    if (apart2(a.z * bc.y - a.y * bc.z, b.z * c.y - b.y * c.z, std::fabs(h.y * bc.z) + std::fabs(h.z * bc.y))) return false;
    if (apart2(a.x * bc.z - a.z * bc.x, b.x * c.z - b.z * c.x, std::fabs(h.x * bc.z) + std::fabs(h.z * bc.x))) return false;
    if (apart2(a.y * bc.x - a.x * bc.y, b.y * c.x - b.x * c.y, std::fabs(h.x * bc.y) + std::fabs(h.y * bc.x))) return false;
    Vec3 const n = cross(ab, ac);
    float const offset = dot(a, n);
    float const reach = std::fabs(n.x * h.x) + std::fabs(n.y * h.y) + std::fabs(n.z * h.z);
    return !apart(offset, offset, reach);
}

// ------------------------------------------------------------------------------------------------
// BSP builder
// ------------------------------------------------------------------------------------------------

struct TriGeom {
    Vec3 a, b, c;        // instantiated vertices
    Vec3 lo, hi;         // their bounds
    uint32_t mesh, tri;  // model index, triangle index within the model
};

}  // namespace


namespace {

struct Builder {
    std::vector<TriGeom> const& tris;
    rt_bsp& out;
    uint32_t maxLeaf, maxDepth;
    float slack;    // absolute slack of the candidate filter, see below

    // Returns the Node index value for the node covering `box`.
    //
    // `cand` lists, in ascending order, every triangle that could pass the exact triangle/box
    // predicate for `box`: it is a superset kept by a cheap bounds test widened by `slack`, so the
    // exact predicate is always evaluated on the node's own box, never inherited from an ancestor.
    // That keeps the result identical to the reference, which re-tests every triangle of the scene
    // against every node box (src/bsp.hpp:188-211), even if rounding made the exact predicate
    // non-monotonic between a box and its half.
    //
    // With `tasks` set, subtrees at depth `cutDepth` or below whose candidate list has shrunk to `taskMax`
    // (a fraction of the scene per worker, so that unevenly filled space still balances) are not built:
    // their inputs are recorded and the returned index is a placeholder (TASK_BASE + task number); they
    // are built independently afterwards and spliced in by Stitcher.
    struct Task { Box box; std::vector<uint32_t> cand; int axis; uint32_t depth; };
    static constexpr int32_t TASK_BASE = 0x40000000;
    std::vector<Task>* tasks = nullptr;
    uint32_t cutDepth = 0;
    size_t taskMax = 0;

    int32_t node(Box const& box, std::vector<uint32_t> const& cand, int axis, uint32_t depth) {
        if (tasks && depth >= cutDepth && cand.size() > 4 * static_cast<size_t>(maxLeaf)
            && (cand.size() <= taskMax || depth >= cutDepth + 12)) {
            tasks->push_back(Task{box, cand, axis, depth});
            return TASK_BASE + static_cast<int32_t>(tasks->size() - 1);
        }
        out.depth = std::max(out.depth, depth);
        std::vector<uint32_t> inside;
        inside.reserve(std::min<size_t>(cand.size(), maxLeaf + 1));
        bool split = false;
        bool const mustStop = depth >= maxDepth;
        for (uint32_t g : cand) {
            TriGeom const& t = tris[g];
            if (!triBox(t.a, t.b, t.c, box)) continue;
            if (inside.size() < maxLeaf || mustStop) inside.push_back(g);
            else { split = true; break; }           // the (maxLeaf+1)-th triangle: subdivide, src/bsp.hpp:206-208
        }
        if (!split) {
            if (inside.empty()) return 0;
            if (inside.size() > maxLeaf) ++out.oversized;
            for (uint32_t g : inside) out.refs.push_back({tris[g].mesh, tris[g].tri});
            out.leaf_first.push_back(static_cast<uint32_t>(out.refs.size()));
            return -static_cast<int32_t>(out.leaf_first.size() - 1);
        }
        Box lower = box, upper = box;
        // Split at the box centre (src/bsp.hpp:254-275).
        if (axis == 0) { float const m = (box.lo.x + box.hi.x) / 2.0f; lower.hi.x = m; upper.lo.x = m; }
        else if (axis == 1) { float const m = (box.lo.y + box.hi.y) / 2.0f; lower.hi.y = m; upper.lo.y = m; }
        else { float const m = (box.lo.z + box.hi.z) / 2.0f; lower.hi.z = m; upper.lo.z = m; }

        size_t const self = out.inodes.size();
        out.inodes.push_back(rt_inode{});
        out.inodes[self].axis = static_cast<uint8_t>(axis);
        int const next = (axis + 1) % 3;
        for (int side = 0; side < 2; ++side) {
            Box const& child = side == 0 ? lower : upper;
            std::vector<uint32_t> sub;
            sub.reserve(cand.size() / 2 + 8);
            for (uint32_t g : cand) {
                TriGeom const& t = tris[g];
                if (t.hi.x >= child.lo.x - slack && t.lo.x <= child.hi.x + slack
                    && t.hi.y >= child.lo.y - slack && t.lo.y <= child.hi.y + slack
                    && t.hi.z >= child.lo.z - slack && t.lo.z <= child.hi.z + slack) sub.push_back(g);
            }
            int32_t const index = node(child, sub, next, depth + 1);
            rt_inode& me = out.inodes[self];
            float* dstBox = side == 0 ? me.neg_box : me.pos_box;
            store3(dstBox, child.lo);
            store3(dstBox + 3, child.hi);
            (side == 0 ? me.neg_index : me.pos_index) = index;
        }
        return static_cast<int32_t>(self + 1);
    }
};

// Re-emits the top of the tree in the order the sequential build would have produced (interior nodes in
// pre-order, negative side first; leaves in creation order) and splices the independently built
// subtrees in where their placeholders stand, shifting their node and leaf numbers.
struct Stitcher {
    rt_bsp const& top;
    std::vector<rt_bsp> const& parts;
    rt_bsp& out;

    void appendLeaf(rt_bsp const& src, uint32_t leaf) {            // leaf numbers are 1-based (index = -leaf)
        for (uint32_t r = src.leaf_first[leaf - 1]; r < src.leaf_first[leaf]; ++r) out.refs.push_back(src.refs[r]);
        out.leaf_first.push_back(static_cast<uint32_t>(out.refs.size()));
    }

    int32_t splice(rt_bsp const& part) {
        int32_t const inodeShift = static_cast<int32_t>(out.inodes.size());
        int32_t const leafShift = static_cast<int32_t>(out.leaf_first.size() - 1);
        auto shifted = [&](int32_t index) { return index > 0 ? index + inodeShift : (index < 0 ? index - leafShift : 0); };
        for (rt_inode n : part.inodes) {
            n.neg_index = shifted(n.neg_index);
            n.pos_index = shifted(n.pos_index);
            out.inodes.push_back(n);
        }
        uint32_t const refShift = static_cast<uint32_t>(out.refs.size());
        out.refs.insert(out.refs.end(), part.refs.begin(), part.refs.end());
        for (size_t l = 1; l < part.leaf_first.size(); ++l) out.leaf_first.push_back(part.leaf_first[l] + refShift);
        out.depth = std::max(out.depth, part.depth);
        out.oversized += part.oversized;
        return shifted(part.root_index);
    }

    int32_t emit(int32_t index) {
        if (index == 0) return 0;
        if (index >= Builder::TASK_BASE) return splice(parts[static_cast<size_t>(index - Builder::TASK_BASE)]);
        if (index < 0) {
            appendLeaf(top, static_cast<uint32_t>(-index));
            return -static_cast<int32_t>(out.leaf_first.size() - 1);
        }
        size_t const self = out.inodes.size();
        out.inodes.push_back(top.inodes[static_cast<size_t>(index - 1)]);
        int32_t const neg = emit(top.inodes[static_cast<size_t>(index - 1)].neg_index);
        int32_t const pos = emit(top.inodes[static_cast<size_t>(index - 1)].pos_index);
        out.inodes[self].neg_index = neg;
        out.inodes[self].pos_index = pos;
        return static_cast<int32_t>(self + 1);
    }
};

}  // namespace

extern "C" {

int rt_line_intersects_box(const float origin[3], const float direction[3], const float box[6]) {
    return lineBox(load3(origin), load3(direction), Box{load3(box), load3(box + 3)}) ? 1 : 0;
}

int rt_tri_intersects_box(const float tri[9], const float box[6]) {
    return triBox(load3(tri), load3(tri + 3), load3(tri + 6), Box{load3(box), load3(box + 3)}) ? 1 : 0;
}

int rt_bsp_build(const float* inst_positions, const uint32_t* model_vertex_begin, const uint32_t* base_tris,
                 const uint32_t* mesh_tri_ranges, const uint32_t* model_meshes, uint32_t model_count,
                 const float root_box[6], uint32_t max_leaf_tris, uint32_t max_depth, int threads, rt_bsp** out) {
    if (!inst_positions || !model_vertex_begin || !base_tris || !mesh_tri_ranges || !model_meshes || !root_box || !out
        || max_leaf_tris == 0) return rt_internal_fail(RT_ERR_INVALID, "rt_bsp_build: null argument or max_leaf_tris == 0");
    rt_bsp* bsp = new (std::nothrow) rt_bsp;
    if (!bsp) return rt_internal_fail(RT_ERR_NOMEM, "rt_bsp_build: out of host memory");
    try {
        std::vector<TriGeom> tris;
        size_t total = 0;
        for (uint32_t m = 0; m < model_count; ++m) total += mesh_tri_ranges[2 * model_meshes[m] + 1];
        tris.reserve(total);
        float scale = 0.0f;
        for (int k = 0; k < 6; ++k) scale = std::max(scale, std::fabs(root_box[k]));
        for (uint32_t m = 0; m < model_count; ++m) {
            uint32_t const begin = mesh_tri_ranges[2 * model_meshes[m]];
            uint32_t const count = mesh_tri_ranges[2 * model_meshes[m] + 1];
            float const* verts = inst_positions + 3 * static_cast<size_t>(model_vertex_begin[m]);
            for (uint32_t k = 0; k < count; ++k) {
                uint32_t const* idx = base_tris + 3 * static_cast<size_t>(begin + k);
                TriGeom t;
                t.a = load3(verts + 3 * static_cast<size_t>(idx[0]));
                t.b = load3(verts + 3 * static_cast<size_t>(idx[1]));
                t.c = load3(verts + 3 * static_cast<size_t>(idx[2]));
                t.lo = {std::min({t.a.x, t.b.x, t.c.x}), std::min({t.a.y, t.b.y, t.c.y}), std::min({t.a.z, t.b.z, t.c.z})};
                t.hi = {std::max({t.a.x, t.b.x, t.c.x}), std::max({t.a.y, t.b.y, t.c.y}), std::max({t.a.z, t.b.z, t.c.z})};
                t.mesh = m;
                t.tri = k;
                for (float v : {t.lo.x, t.lo.y, t.lo.z, t.hi.x, t.hi.y, t.hi.z}) scale = std::max(scale, std::fabs(v));
                tris.push_back(t);
            }
        }
        std::memcpy(bsp->root_box, root_box, sizeof bsp->root_box);
        bsp->leaf_first.push_back(0);
        std::vector<uint32_t> all(tris.size());
        for (size_t i = 0; i < all.size(); ++i) all[i] = static_cast<uint32_t>(i);
        // The exact predicate's first three axes compare translated coordinates whose rounding
        // error is a few ulps of the scene's coordinate magnitude; 1e-4 of that magnitude is far
        // outside it, so no triangle that the exact test would accept is ever filtered out.
        float const slack = 1e-4f * scale;
        // threads <= 0: OMP_NUM_THREADS if set (the usual statement of a process's CPU share), else the
        // hardware concurrency, capped at 16.
        unsigned workers = threads > 0 ? static_cast<unsigned>(threads) : 0u;
        if (workers == 0) {
            if (const char* env = std::getenv("OMP_NUM_THREADS")) workers = static_cast<unsigned>(std::max(0, std::atoi(env)));
            if (workers == 0) workers = std::min(std::thread::hardware_concurrency(), 16u);
        }
        workers = std::max(1u, std::min(workers, 64u));
        if (workers == 1 || tris.size() < 200000) {
            Builder b{tris, *bsp, max_leaf_tris, max_depth, slack};
            bsp->root_index = b.node(Box{load3(root_box), load3(root_box + 3)}, all, 0, 0);
        }
        else {
            // Top of the tree on this thread, down to subtrees holding at most 1/(8 workers) of the scene;
            // the subtrees in parallel, largest first; then one pass that restores the sequential order.
            uint32_t cut = 2;
            while ((1u << cut) < workers && cut < 8) ++cut;
            rt_bsp top;
            top.leaf_first.push_back(0);
            std::vector<Builder::Task> tasks;
            Builder tb{tris, top, max_leaf_tris, max_depth, slack};
            tb.tasks = &tasks;
            tb.cutDepth = std::min(cut, max_depth);
            tb.taskMax = std::max<size_t>(tris.size() / (8 * static_cast<size_t>(workers)), 1024);
            top.root_index = tb.node(Box{load3(root_box), load3(root_box + 3)}, all, 0, 0);
            std::vector<uint32_t>().swap(all);
            std::vector<rt_bsp> parts(tasks.size());
            std::vector<size_t> order(tasks.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return tasks[a].cand.size() > tasks[b].cand.size(); });
            std::atomic<size_t> next{0};
            std::atomic<bool> failed{false};
            auto work = [&]() {
                try {
                    for (size_t k = next.fetch_add(1); k < order.size(); k = next.fetch_add(1)) {
                        size_t const i = order[k];
                        parts[i].leaf_first.push_back(0);
                        Builder pb{tris, parts[i], max_leaf_tris, max_depth, slack};
                        parts[i].root_index = pb.node(tasks[i].box, tasks[i].cand, tasks[i].axis, tasks[i].depth);
                        std::vector<uint32_t>().swap(tasks[i].cand);
                    }
                }
                catch (std::bad_alloc const&) { failed = true; }
            };
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
            if (failed) throw std::bad_alloc();
            bsp->depth = top.depth;
            bsp->oversized = top.oversized;
            Stitcher st{top, parts, *bsp};
            bsp->root_index = st.emit(top.root_index);
        }
    }
    catch (std::bad_alloc const&) {
        delete bsp;
        return rt_internal_fail(RT_ERR_NOMEM, "rt_bsp_build: out of host memory");
    }
    catch (...) {
        delete bsp;
        return rt_internal_fail(RT_ERR_INVALID, "rt_bsp_build: unexpected exception");
    }
    *out = bsp;
    return RT_OK;
}

void rt_bsp_destroy(rt_bsp* bsp) { delete bsp; }

void rt_bsp_counts(const rt_bsp* bsp, uint32_t counts[5]) {
    counts[0] = static_cast<uint32_t>(bsp->inodes.size());
    counts[1] = static_cast<uint32_t>(bsp->leaf_first.size() - 1);
    counts[2] = static_cast<uint32_t>(bsp->refs.size());
    counts[3] = bsp->depth;
    counts[4] = bsp->oversized;
}

void rt_bsp_view(const rt_bsp* bsp, float root_box[6], int32_t* root_index, const rt_inode** inodes,
                 const uint32_t** leaf_first_ref, const rt_tri_ref** leaf_refs) {
    std::memcpy(root_box, bsp->root_box, sizeof bsp->root_box);
    *root_index = bsp->root_index;
    *inodes = bsp->inodes.data();
    *leaf_first_ref = bsp->leaf_first.data();
    *leaf_refs = bsp->refs.data();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Scene description <-> one contiguous byte buffer (for shipping a prepared scene to other processes)
// ------------------------------------------------------------------------------------------------
namespace {

struct BlobHeader {
    uint32_t magic, version;
    float root_box[6];
    int32_t root_index;
    uint32_t inode_count, leaf_count, ref_count, pretri_count, vertex_count, base_tri_count, model_count;
    uint32_t has_pretris, has_leaf_pretris;
    uint64_t total_bytes;
};
constexpr uint32_t kBlobMagic = 0x42535452u;      // "RTSB"
constexpr uint32_t kBlobVersion = 1;
inline size_t pad16(size_t n) { return (n + 15) & ~static_cast<size_t>(15); }

// The arrays of a description in blob order, with their byte sizes.
struct BlobLayout {
    size_t size[11];
    size_t offset[11];
    size_t total;
};
BlobLayout blobLayout(BlobHeader const& h) {
    BlobLayout L{};
    size_t const models = h.model_count;
    L.size[0] = sizeof(rt_inode) * static_cast<size_t>(h.inode_count);
    L.size[1] = 4 * (static_cast<size_t>(h.leaf_count) + 1);
    L.size[2] = sizeof(rt_tri_ref) * static_cast<size_t>(h.ref_count);
    L.size[3] = h.has_pretris ? 48 * static_cast<size_t>(h.pretri_count) : 0;
    L.size[4] = 4 * (models + 1);
    L.size[5] = 12 * static_cast<size_t>(h.vertex_count);
    L.size[6] = 4 * (models + 1);
    L.size[7] = 12 * static_cast<size_t>(h.base_tri_count);
    L.size[8] = 4 * models;
    L.size[9] = 48 * models;
    L.size[10] = h.has_leaf_pretris ? 48 * static_cast<size_t>(h.ref_count) : 0;
    size_t cursor = pad16(sizeof(BlobHeader));
    for (int k = 0; k < 11; ++k) { L.offset[k] = cursor; cursor += pad16(L.size[k]); }
    L.total = cursor;
    return L;
}

}  // namespace

extern "C" {

int rt_scene_serialize(const rt_scene_desc* D, void* blob, uint64_t* size) {
    if (!D || !size) return rt_internal_fail(RT_ERR_INVALID, "rt_scene_serialize: null argument");
    if (!D->leaf_first_ref || !D->model_pretri_begin || !D->model_vertex_begin)
        return rt_internal_fail(RT_ERR_INVALID, "rt_scene_serialize: null offset array in description");
    BlobHeader h{};
    h.magic = kBlobMagic; h.version = kBlobVersion;
    std::memcpy(h.root_box, D->root_box, sizeof h.root_box);
    h.root_index = D->root_index;
    h.inode_count = D->inode_count; h.leaf_count = D->leaf_count; h.ref_count = D->leaf_first_ref[D->leaf_count];
    h.pretri_count = D->pretri_count; h.vertex_count = D->vertex_count; h.base_tri_count = D->base_tri_count;
    h.model_count = D->model_count;
    h.has_pretris = D->pretris ? 1u : 0u;
    h.has_leaf_pretris = D->leaf_pretris ? 1u : 0u;
    BlobLayout const L = blobLayout(h);
    h.total_bytes = L.total;
    if (!blob) { *size = L.total; return RT_OK; }                 // size query
    if (*size < L.total) { *size = L.total; return rt_internal_fail(RT_ERR_LIMIT, "rt_scene_serialize: buffer too small"); }
    const void* src[11] = {D->inodes, D->leaf_first_ref, D->leaf_refs, D->pretris, D->model_pretri_begin, D->vertex_normals,
                           D->model_vertex_begin, D->base_tris, D->model_tri_begin, D->model_materials, D->leaf_pretris};
    unsigned char* out = static_cast<unsigned char*>(blob);
    std::memset(out, 0, pad16(sizeof h));
    std::memcpy(out, &h, sizeof h);
    for (int k = 0; k < 11; ++k) {
        if (L.size[k] == 0) continue;
        if (!src[k]) return rt_internal_fail(RT_ERR_INVALID, "rt_scene_serialize: null array in description");
        std::memcpy(out + L.offset[k], src[k], L.size[k]);
        std::memset(out + L.offset[k] + L.size[k], 0, pad16(L.size[k]) - L.size[k]);
    }
    *size = L.total;
    return RT_OK;
}

int rt_scene_desc_from_blob(const void* blob, uint64_t size, rt_scene_desc* D) {
    if (!blob || !D) return rt_internal_fail(RT_ERR_INVALID, "rt_scene_desc_from_blob: null argument");
    if (size < sizeof(BlobHeader)) return rt_internal_fail(RT_ERR_INVALID, "rt_scene_desc_from_blob: buffer shorter than its header");
    if (reinterpret_cast<uintptr_t>(blob) % 8 != 0) return rt_internal_fail(RT_ERR_INVALID, "rt_scene_desc_from_blob: buffer not 8-byte aligned");
    BlobHeader h;
    std::memcpy(&h, blob, sizeof h);
    if (h.magic != kBlobMagic || h.version != kBlobVersion)
        return rt_internal_fail(RT_ERR_INVALID, "rt_scene_desc_from_blob: not a scene buffer of this library version");
    BlobLayout const L = blobLayout(h);
    if (L.total != h.total_bytes || size < L.total)
        return rt_internal_fail(RT_ERR_INVALID, "rt_scene_desc_from_blob: truncated or inconsistent buffer");
    const unsigned char* in = static_cast<const unsigned char*>(blob);
    auto at = [&](int k) -> const void* { return L.size[k] ? in + L.offset[k] : nullptr; };
    std::memset(D, 0, sizeof *D);
    std::memcpy(D->root_box, h.root_box, sizeof h.root_box);
    D->root_index = h.root_index;
    D->inodes = static_cast<const rt_inode*>(at(0)); D->inode_count = h.inode_count;
    D->leaf_first_ref = static_cast<const uint32_t*>(at(1)); D->leaf_count = h.leaf_count;
    D->leaf_refs = static_cast<const rt_tri_ref*>(at(2));
    D->pretris = static_cast<const float*>(at(3)); D->pretri_count = h.pretri_count;
    D->model_pretri_begin = static_cast<const uint32_t*>(at(4));
    D->vertex_normals = static_cast<const float*>(at(5)); D->vertex_count = h.vertex_count;
    D->model_vertex_begin = static_cast<const uint32_t*>(at(6));
    D->base_tris = static_cast<const uint32_t*>(at(7)); D->base_tri_count = h.base_tri_count;
    D->model_tri_begin = static_cast<const uint32_t*>(at(8));
    D->model_materials = static_cast<const float*>(at(9)); D->model_count = h.model_count;
    D->leaf_pretris = static_cast<const float*>(at(10));
    if (D->leaf_first_ref[h.leaf_count] != h.ref_count)
        return rt_internal_fail(RT_ERR_INVALID, "rt_scene_desc_from_blob: leaf offsets do not end at the reference count");
    return RT_OK;
}

int rt_scene_create_from_blob(const void* blob, uint64_t size, int device, rt_scene** out) {
    rt_scene_desc D;
    int const rc = rt_scene_desc_from_blob(blob, size, &D);
    if (rc != RT_OK) return rc;
    return rt_scene_create(&D, device, out);
}

}  // extern "C"
