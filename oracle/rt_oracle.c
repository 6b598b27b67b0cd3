/* TEST INFRASTRUCTURE ONLY -- see rt_oracle.h.  Plain-C restatement of the reference hot path.
 * Build: gcc -std=c11 -O2 -mavx2 -mfma -ffp-contract=off -fopenmp (oracle/Makefile).
 * All `file:line` citations are relative to /root/reference. */
#include "rt_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add3(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scale3(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float axis3(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

/* glm scalar paths (libraries/glm/glm/detail/func_geometric.inl:48-55, :82-90, :66-76;
 * func_exponential.inl:134-139): dot is a left-to-right sum of products, normalize multiplies by
 * 1/sqrt(dot), cross is a.y*b.z - b.y*a.z etc.  No FMA (contraction is off). */
static inline float glm_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 glm_normalize(v3 a) { return scale3(a, 1.0f / sqrtf(glm_dot(a, a))); }
static inline v3 glm_cross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}

/* AVX wrapper semantics, one lane (src/utility/vectorised.hpp:466-481):
 * dot = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)); cross component = fms(a.y,b.z, a.z*b.y). */
static inline float simd_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 simd_cross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

typedef struct { v3 min, max; } box3;
typedef struct { v3 normal, v1, v1ToV3, v1ToV2; } pretri;         /* src/geometry.hpp:44-49 */
typedef struct { float ndfAlphaSq, geometryAlphaSq; v3 f0, adjustedColour, emission; } premat; /* src/material.hpp:23-29 */

/* ------------------------------------------------------------------------------------------ */
/* lineIntersectsBox, src/geometry.hpp:135-214                                                  */
/* ------------------------------------------------------------------------------------------ */

/* src/geometry.hpp:136-149: plane hit parameter, valid when |n.d| >= 1e-6 and t >= 0. */
static inline int plane_hit(float nDotD, float p0MinusL0DotN, float* t) {
    if (fabsf(nDotD) < 1e-6f) return 0;
    *t = p0MinusL0DotN / nDotD;
    return *t >= 0.0f;
}

static int line_intersects_box(v3 o, v3 d, const box3* b, orc_stats* st) {
    float t;
    if (st) st->boxTests++;
    /* -X, +X, -Y, +Y, -Z, +Z faces in the reference's order (:153-212). */
    if (plane_hit(-d.x, o.x - b->min.x, &t)) {
        float y = o.y + t * d.y;
        if (y >= b->min.y && y <= b->max.y) {
            float z = o.z + t * d.z;
            if (z >= b->min.z && z <= b->max.z) return 1;
        }
    }
    if (plane_hit(d.x, b->max.x - o.x, &t)) {
        float y = o.y + t * d.y;
        if (y >= b->min.y && y <= b->max.y) {
            float z = o.z + t * d.z;
            if (z >= b->min.z && z <= b->max.z) return 1;
        }
    }
    if (plane_hit(-d.y, o.y - b->min.y, &t)) {
        float x = o.x + t * d.x;
        if (x >= b->min.x && x <= b->max.x) {
            float z = o.z + t * d.z;
            if (z >= b->min.z && z <= b->max.z) return 1;
        }
    }
    if (plane_hit(d.y, b->max.y - o.y, &t)) {
        float x = o.x + t * d.x;
        if (x >= b->min.x && x <= b->max.x) {
            float z = o.z + t * d.z;
            if (z >= b->min.z && z <= b->max.z) return 1;
        }
    }
    if (plane_hit(-d.z, o.z - b->min.z, &t)) {
        float x = o.x + t * d.x;
        if (x >= b->min.x && x <= b->max.x) {
            float y = o.y + t * d.y;
            if (y >= b->min.y && y <= b->max.y) return 1;
        }
    }
    if (plane_hit(d.z, b->max.z - o.z, &t)) {
        float x = o.x + t * d.x;
        if (x >= b->min.x && x <= b->max.x) {
            float y = o.y + t * d.y;
            if (y >= b->min.y && y <= b->max.y) return 1;
        }
    }
    return 0;
}

int orc_line_intersects_box1(const float origin[3], const float direction[3], const float box[6]) {
    box3 b = {ld3(box), ld3(box + 3)};
    return line_intersects_box(ld3(origin), ld3(direction), &b, NULL);
}

void orc_line_intersects_box(const float* origins, const float* directions, const float* boxes, size_t n,
                             uint8_t* out) {
    for (size_t i = 0; i < n; ++i)
        out[i] = (uint8_t)orc_line_intersects_box1(origins + 3 * i, directions + 3 * i, boxes + 6 * i);
}

/* ------------------------------------------------------------------------------------------ */
/* triIntersectsBox, src/geometry.hpp:217-428 (13-axis separating-axis test)                   */
/* ------------------------------------------------------------------------------------------ */

/* src/geometry.hpp:220-222: strict comparison, touching intervals overlap. */
static inline int disjoint(float aMin, float aMax, float bMin, float bMax) { return aMax < bMin || bMax < aMin; }
static inline float min2(float a, float b) { return b < a ? b : a; }      /* std::min / std::minmax semantics */
static inline float max2(float a, float b) { return a < b ? b : a; }
static inline float min3(float a, float b, float c) { return min2(min2(a, b), c); }
static inline float max3(float a, float b, float c) { return max2(max2(a, b), c); }

/* One "box axis x tri edge" test (:283-410).  p and q are the two distinct vertex projections,
 * r1,e1,r2,e2 give the box projection radius |r1*e1| + |r2*e2|. */
static inline int edge_axis_separates(float p, float q, float r1, float e1, float r2, float e2) {
    float lo = min2(p, q), hi = max2(p, q);
    float boxMax = fabsf(r1 * e1) + fabsf(r2 * e2);
    return disjoint(lo, hi, -boxMax, boxMax);
}

static int tri_intersects_box(v3 a, v3 b, v3 c, const box3* box) {
    v3 r, ctr;
    /* The reference translates and tests one axis at a time (:225-279). */
    r.x = (box->max.x - box->min.x) / 2.0f;
    ctr.x = box->min.x + r.x;
    a.x -= ctr.x; b.x -= ctr.x; c.x -= ctr.x;
    if (disjoint(min3(a.x, b.x, c.x), max3(a.x, b.x, c.x), -r.x, r.x)) return 0;
    r.y = (box->max.y - box->min.y) / 2.0f;
    ctr.y = box->min.y + r.y;
    a.y -= ctr.y; b.y -= ctr.y; c.y -= ctr.y;
    if (disjoint(min3(a.y, b.y, c.y), max3(a.y, b.y, c.y), -r.y, r.y)) return 0;
    r.z = (box->max.z - box->min.z) / 2.0f;
    ctr.z = box->min.z + r.z;
    a.z -= ctr.z; b.z -= ctr.z; c.z -= ctr.z;
    if (disjoint(min3(a.z, b.z, c.z), max3(a.z, b.z, c.z), -r.z, r.z)) return 0;

    /* Edge 1 = v2 - v1 (:282-324). */
    v3 e1 = sub3(b, a);
    if (edge_axis_separates(a.z * b.y - a.y * b.z, c.z * e1.y - c.y * e1.z, r.y, e1.z, r.z, e1.y)) return 0;
    if (edge_axis_separates(a.x * b.z - a.z * b.x, c.x * e1.z - c.z * e1.x, r.x, e1.z, r.z, e1.x)) return 0;
    if (edge_axis_separates(a.y * b.x - a.x * b.y, c.y * e1.x - c.x * e1.y, r.x, e1.y, r.y, e1.x)) return 0;
    /* Edge 2 = v3 - v1 (:325-367). */
    v3 e2 = sub3(c, a);
    if (edge_axis_separates(a.z * c.y - a.y * c.z, b.z * e2.y - b.y * e2.z, r.y, e2.z, r.z, e2.y)) return 0;
    if (edge_axis_separates(a.x * c.z - a.z * c.x, b.x * e2.z - b.z * e2.x, r.x, e2.z, r.z, e2.x)) return 0;
    if (edge_axis_separates(a.y * c.x - a.x * c.y, b.y * e2.x - b.x * e2.y, r.x, e2.y, r.y, e2.x)) return 0;
    /* Edge 3 = v3 - v2 (:368-410). */
    v3 e3 = sub3(c, b);
    if (edge_axis_separates(a.z * e3.y - a.y * e3.z, b.z * c.y - b.y * c.z, r.y, e3.z, r.z, e3.y)) return 0;
    if (edge_axis_separates(a.x * e3.z - a.z * e3.x, b.x * c.z - b.z * c.x, r.x, e3.z, r.z, e3.x)) return 0;
    if (edge_axis_separates(a.y * e3.x - a.x * e3.y, b.y * c.x - b.x * c.y, r.x, e3.y, r.y, e3.x)) return 0;

    /* Triangle normal (:413-425). */
    v3 n = glm_cross(e1, e2);
    float proj = glm_dot(a, n);
    float boxMax = fabsf(n.x * r.x) + fabsf(n.y * r.y) + fabsf(n.z * r.z);
    if (disjoint(proj, proj, -boxMax, boxMax)) return 0;
    return 1;
}

int orc_tri_intersects_box1(const float tri[9], const float box[6]) {
    box3 b = {ld3(box), ld3(box + 3)};
    return tri_intersects_box(ld3(tri), ld3(tri + 3), ld3(tri + 6), &b);
}

void orc_tri_intersects_box(const float* tris, const float* boxes, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)orc_tri_intersects_box1(tris + 9 * i, boxes + 6 * i);
}

/* ------------------------------------------------------------------------------------------ */
/* FastRNG, src/utility/random.hpp:10-34                                                       */
/* ------------------------------------------------------------------------------------------ */

typedef struct { uint64_t state; } fast_rng;
static inline uint32_t rng_value(fast_rng* r) {
    r->state = 214013 * r->state + 2531011;
    return (uint32_t)(r->state >> 16);
}
static inline float rng_unit(fast_rng* r) { return (float)rng_value(r) / 0x100000000; }  /* can round to 1.0f */
static inline float rng_angle(fast_rng* r) { return 6.28318530717958647692528676655900576f * rng_unit(r); }

void orc_fast_rng_floats(uint64_t seed, size_t n, float* out) {
    fast_rng r = {seed};
    for (size_t i = 0; i < n; ++i) out[i] = rng_unit(&r);
}

/* ------------------------------------------------------------------------------------------ */
/* Scene                                                                                        */
/* ------------------------------------------------------------------------------------------ */

typedef struct { box3 box; int32_t index; } bsp_node;                  /* src/bsp.hpp:148-153 */
typedef struct { bsp_node neg, pos; uint8_t axis; } bsp_inode;         /* src/bsp.hpp:155-159 */
/* Leaf, src/bsp.hpp:161-168: the reference holds up to 32 triangles in four 8-wide blocks.  Here a leaf is a
 * range of the scene-wide reference arrays (refMesh/refTri/refTris), so that trees with larger leaves (which
 * only the product's depth-capped builder can make) can be traversed with the same scan; `firstBlock` indexes
 * the 8-wide structure-of-arrays copies used by the AVX2 leaf test. */
typedef struct { uint32_t first, count, firstBlock; } bsp_leaf;
/* PreprocessedTriBlock, src/geometry.hpp:53-58: normal, v1, v1ToV3, v1ToV2 of 8 triangles, one array per component. */
typedef struct { float c[12][8]; } tri_block;

struct orc_scene {
    uint32_t modelCount, vertexCount, triCount, baseVertexCount, baseTriCount, meshCount, materialCount;
    uint32_t width, height;
    /* base meshes */
    uint32_t* baseTris;            /* 3 per tri */
    uint32_t* meshTriRanges;       /* {begin,size} per mesh */
    uint32_t* modelMeshes;
    /* preprocessed */
    v3* positions; v3* normals;    /* instantiated */
    uint32_t* vertexRanges;        /* per model {begin,size} */
    pretri* pretris; uint32_t* triRanges;   /* per model {begin,size} */
    premat* modelMaterials;        /* permutation applied: per model */
    v3 cameraPosition; float pixelToRay[9]; box3 rootBox;
    /* tree */
    bsp_node root;
    bsp_inode* inodes; uint32_t inodeCount, inodeCap;
    bsp_leaf* leaves; uint32_t leafCount, leafCap;
    uint32_t* refMesh; uint32_t* refTri; pretri* refTris; uint32_t refCount, refCap;
    tri_block* blocks; uint32_t blockCount;
    int simdLeaves;                /* leaf test: 0 = one triangle at a time, 1 = AVX2, 8 triangles per step */
    int buildFailed;
    /* the product's alternative accelerator (orc_scene_set_bvh4); not part of the reference */
    float* bvhNodes; uint32_t bvhNodeCount;        /* 32 words per node */
    pretri* bvhTris; uint32_t* bvhMesh; uint32_t* bvhTri; uint32_t* bvhGlobal; uint32_t bvhTriCount;
};

/* src/material.hpp:32-45 */
static premat preprocess_material(const float* m) {
    premat p;
    float r = m[3], r4 = 1.0f;
    for (int i = 0; i < 4; ++i) r4 *= r;                        /* iPow, src/utility/math.hpp:19-25 */
    p.ndfAlphaSq = r4;
    p.geometryAlphaSq = r4 / 4.0f;
    float omm = 1.0f - m[4];
    p.f0 = V(omm * 0.04f + m[4] * m[0], omm * 0.04f + m[4] * m[1], omm * 0.04f + m[4] * m[2]);
    float pi = 3.14159265358979323846264338327950288f;
    p.adjustedColour = V(omm * m[0] / pi, omm * m[1] / pi, omm * m[2] / pi);
    p.emission = V(m[5], m[6], m[7]);
    return p;
}

/* glm quaternion * vec3, libraries/glm/glm/detail/type_quat.inl:343-350 */
static v3 quat_rotate(const float q[4] /* w,x,y,z */, v3 v) {
    v3 qv = V(q[1], q[2], q[3]);
    v3 uv = glm_cross(qv, v);
    v3 uuv = glm_cross(qv, uv);
    v3 s = add3(scale3(uv, q[0]), uuv);
    return add3(v, scale3(s, 2.0f));
}

/* src/camera.hpp:33-46: o computed in double then narrowed; columns {right, down, o}. */
static void pixel_to_ray(v3 fwd, v3 down, v3 right, float fov, uint32_t w, uint32_t h, float* m) {
    double k = w / tan(fov / 2.0);
    double ox = (k * (double)fwd.x - (double)w * (double)right.x - (double)h * (double)down.x) / 2.0;
    double oy = (k * (double)fwd.y - (double)w * (double)right.y - (double)h * (double)down.y) / 2.0;
    double oz = (k * (double)fwd.z - (double)w * (double)right.z - (double)h * (double)down.z) / 2.0;
    st3(m, right); st3(m + 3, down);
    m[6] = (float)ox; m[7] = (float)oy; m[8] = (float)oz;
}

/* MeshTransform::matrix(), src/mesh.hpp:62-69, on top of glm mat3_cast
 * (libraries/glm/glm/gtc/quaternion.inl:41-66).  Columns c0,c1,c2 and translation c3. */
static void transform_matrix(const float* t, v3 c[4]) {
    float w = t[3], x = t[4], y = t[5], z = t[6];
    float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z;
    float qwx = w * x, qwy = w * y, qwz = w * z;
    c[0] = V(1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy));
    c[1] = V(2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx));
    c[2] = V(2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy));
    c[0] = scale3(c[0], t[7]); c[1] = scale3(c[1], t[8]); c[2] = scale3(c[2], t[9]);
    c[3] = V(t[0], t[1], t[2]);
}

/* glm::inverseTranspose(mat3), libraries/glm/glm/gtc/matrix_inverse.inl:43-63.  m[col] = column. */
static void inverse_transpose3(const v3 m[3], v3 out[3]) {
    float m00 = m[0].x, m01 = m[0].y, m02 = m[0].z;
    float m10 = m[1].x, m11 = m[1].y, m12 = m[1].z;
    float m20 = m[2].x, m21 = m[2].y, m22 = m[2].z;
    float det = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
    out[0] = V(+(m11 * m22 - m21 * m12), -(m10 * m22 - m20 * m12), +(m10 * m21 - m20 * m11));
    out[1] = V(-(m01 * m22 - m21 * m02), +(m00 * m22 - m20 * m02), -(m00 * m21 - m20 * m01));
    out[2] = V(+(m01 * m12 - m11 * m02), -(m00 * m12 - m10 * m02), +(m00 * m11 - m10 * m01));
    for (int i = 0; i < 3; ++i) out[i] = V(out[i].x / det, out[i].y / det, out[i].z / det);
}

/* src/bsp.hpp:174-285, restated faithfully: every node rescans all triangles of all instances. */
static bsp_node create_node(orc_scene* s, box3 box, int axis, int depth) {
    bsp_node node; node.box = box; node.index = 0;
    if (depth > 64) { s->buildFailed = 1; return node; }
    struct { uint32_t mesh[32], tri[32]; pretri tris[32]; uint32_t count; } leaf; leaf.count = 0;
    int subdivide = 0;
    for (uint32_t inst = 0; inst < s->modelCount && !subdivide; ++inst) {
        const uint32_t* mtr = s->meshTriRanges + 2 * s->modelMeshes[inst];
        const uint32_t* tris = s->baseTris + 3 * (size_t)mtr[0];
        const v3* pos = s->positions + s->vertexRanges[2 * inst];
        const pretri* pts = s->pretris + s->triRanges[2 * inst];
        for (uint32_t t = 0; t < mtr[1] && !subdivide; ++t) {
            if (tri_intersects_box(pos[tris[3 * t]], pos[tris[3 * t + 1]], pos[tris[3 * t + 2]], &box)) {
                if (leaf.count < 32) {
                    leaf.tris[leaf.count] = pts[t];
                    leaf.mesh[leaf.count] = inst;
                    leaf.tri[leaf.count] = t;
                    ++leaf.count;
                }
                else subdivide = 1;
            }
        }
    }
    if (!subdivide) {
        if (leaf.count == 0) return node;                              /* :213-215 */
        if (s->leafCount == s->leafCap) {
            s->leafCap = s->leafCap ? 2 * s->leafCap : 64;
            s->leaves = (bsp_leaf*)realloc(s->leaves, s->leafCap * sizeof(bsp_leaf));
        }
        if (s->refCount + leaf.count > s->refCap) {
            s->refCap = s->refCap ? 2 * s->refCap : 1024;
            s->refMesh = (uint32_t*)realloc(s->refMesh, s->refCap * sizeof(uint32_t));
            s->refTri = (uint32_t*)realloc(s->refTri, s->refCap * sizeof(uint32_t));
            s->refTris = (pretri*)realloc(s->refTris, s->refCap * sizeof(pretri));
        }
        bsp_leaf rec = {s->refCount, leaf.count, 0};
        for (uint32_t k = 0; k < leaf.count; ++k) {
            s->refMesh[s->refCount] = leaf.mesh[k]; s->refTri[s->refCount] = leaf.tri[k];
            s->refTris[s->refCount] = leaf.tris[k];
            ++s->refCount;
        }
        s->leaves[s->leafCount++] = rec;
        node.index = -(int32_t)s->leafCount;                           /* :249 */
        return node;
    }
    box3 nb = box, pb = box;                                           /* :254-275 */
    if (axis == 0) { float c = (box.min.x + box.max.x) / 2.0f; nb.max.x = c; pb.min.x = c; }
    else if (axis == 1) { float c = (box.min.y + box.max.y) / 2.0f; nb.max.y = c; pb.min.y = c; }
    else { float c = (box.min.z + box.max.z) / 2.0f; nb.max.z = c; pb.min.z = c; }
    int next = (axis + 1) % 3;
    if (s->inodeCount == s->inodeCap) {
        s->inodeCap = s->inodeCap ? 2 * s->inodeCap : 64;
        s->inodes = (bsp_inode*)realloc(s->inodes, s->inodeCap * sizeof(bsp_inode));
    }
    uint32_t idx = s->inodeCount++;                                    /* pre-order, :277-279 */
    memset(&s->inodes[idx], 0, sizeof(bsp_inode));
    s->inodes[idx].axis = (uint8_t)axis;
    bsp_node n = create_node(s, nb, next, depth + 1);
    s->inodes[idx].neg = n;
    bsp_node p = create_node(s, pb, next, depth + 1);
    s->inodes[idx].pos = p;
    node.index = (int32_t)(idx + 1);
    return node;
}

static orc_scene* scene_prepare(const float* positions, const float* normals, uint32_t vertexCount,
                                const uint32_t* tris, uint32_t triCount,
                                const uint32_t* meshVertexRanges, const uint32_t* meshTriRanges, uint32_t meshCount,
                                const float* materials, uint32_t materialCount,
                                const float* transforms, const uint32_t* modelMeshes, const uint32_t* modelMaterials,
                                uint32_t modelCount, const float* camera, uint32_t width, uint32_t height,
                                int inflateMode) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->modelCount = modelCount; s->baseVertexCount = vertexCount; s->baseTriCount = triCount;
    s->meshCount = meshCount; s->materialCount = materialCount; s->width = width; s->height = height;
    s->baseTris = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (size_t)triCount + 4);
    memcpy(s->baseTris, tris, sizeof(uint32_t) * 3 * (size_t)triCount);
    s->meshTriRanges = (uint32_t*)malloc(sizeof(uint32_t) * 2 * meshCount + 4);
    memcpy(s->meshTriRanges, meshTriRanges, sizeof(uint32_t) * 2 * meshCount);
    s->modelMeshes = (uint32_t*)malloc(sizeof(uint32_t) * modelCount + 4);
    memcpy(s->modelMeshes, modelMeshes, sizeof(uint32_t) * modelCount);

    /* src/main.cpp:191-192 + the model->material permutation of src/main.cpp:227 */
    s->modelMaterials = (premat*)malloc(sizeof(premat) * modelCount + 4);
    for (uint32_t i = 0; i < modelCount; ++i)
        s->modelMaterials[i] = preprocess_material(materials + 8 * (size_t)modelMaterials[i]);
    (void)materialCount;

    /* src/camera.hpp:18-28 and :33-46 */
    {
        const float* q = camera + 3;
        v3 fwd = quat_rotate(q, V(0.0f, 0.0f, 1.0f));
        v3 down = quat_rotate(q, V(0.0f, -1.0f, 0.0f));
        v3 right = quat_rotate(q, V(-1.0f, 0.0f, 0.0f));
        pixel_to_ray(fwd, down, right, camera[7], width, height, s->pixelToRay);
        s->cameraPosition = ld3(camera);
    }

    /* instantiateMeshes, src/mesh.hpp:90-138 */
    size_t nv = 0, nt = 0;
    for (uint32_t i = 0; i < modelCount; ++i) {
        nv += meshVertexRanges[2 * modelMeshes[i] + 1];
        nt += meshTriRanges[2 * modelMeshes[i] + 1];
    }
    s->vertexCount = (uint32_t)nv; s->triCount = (uint32_t)nt;
    s->positions = (v3*)malloc(sizeof(v3) * nv + 4);
    s->normals = (v3*)malloc(sizeof(v3) * nv + 4);
    s->vertexRanges = (uint32_t*)malloc(sizeof(uint32_t) * 2 * modelCount + 4);
    s->pretris = (pretri*)malloc(sizeof(pretri) * nt + 4);
    s->triRanges = (uint32_t*)malloc(sizeof(uint32_t) * 2 * modelCount + 4);
    size_t voff = 0;
    for (uint32_t i = 0; i < modelCount; ++i) {
        v3 c[4];
        transform_matrix(transforms + 10 * (size_t)i, c);
        uint32_t vb = meshVertexRanges[2 * modelMeshes[i]], vs = meshVertexRanges[2 * modelMeshes[i] + 1];
        v3 nm[3];
        inverse_transpose3(c, nm);                                     /* src/mesh.hpp:74-76 */
        for (uint32_t k = 0; k < vs; ++k) {
            v3 p = ld3(positions + 3 * (size_t)(vb + k));
            /* mat4x3 * vec4{p,1}, libraries/glm/glm/detail/type_mat4x3.inl:469-478 */
            s->positions[voff + k] = V(c[0].x * p.x + c[1].x * p.y + c[2].x * p.z + c[3].x * 1.0f,
                                       c[0].y * p.x + c[1].y * p.y + c[2].y * p.z + c[3].y * 1.0f,
                                       c[0].z * p.x + c[1].z * p.y + c[2].z * p.z + c[3].z * 1.0f);
            v3 n = ld3(normals + 3 * (size_t)(vb + k));
            /* mat3 * vec3, libraries/glm/glm/detail/type_mat3x3.inl:468-474 */
            v3 tn = V(nm[0].x * n.x + nm[1].x * n.y + nm[2].x * n.z,
                      nm[0].y * n.x + nm[1].y * n.y + nm[2].y * n.z,
                      nm[0].z * n.x + nm[1].z * n.y + nm[2].z * n.z);
            s->normals[voff + k] = glm_normalize(tn);
        }
        s->vertexRanges[2 * i] = (uint32_t)voff; s->vertexRanges[2 * i + 1] = vs;
        voff += vs;
    }
    /* preprocessTris, src/mesh.hpp:149-188 with preprocessTri, src/geometry.hpp:77-82 */
    size_t toff = 0;
    for (uint32_t i = 0; i < modelCount; ++i) {
        uint32_t tb = meshTriRanges[2 * modelMeshes[i]], ts = meshTriRanges[2 * modelMeshes[i] + 1];
        const v3* pos = s->positions + s->vertexRanges[2 * i];
        for (uint32_t k = 0; k < ts; ++k) {
            const uint32_t* t = tris + 3 * (size_t)(tb + k);
            v3 a = pos[t[0]], b = pos[t[1]], c = pos[t[2]];
            pretri p;
            p.v1ToV2 = sub3(b, a);
            p.v1ToV3 = sub3(c, a);
            p.normal = glm_cross(p.v1ToV2, p.v1ToV3);
            p.v1 = a;
            s->pretris[toff + k] = p;
        }
        s->triRanges[2 * i] = (uint32_t)toff; s->triRanges[2 * i + 1] = ts;
        toff += ts;
    }
    /* computeBoundingBox, src/geometry.hpp:85-92 (glm::min(x,y) = y<x ? y : x per component) */
    box3 bb = {V(INFINITY, INFINITY, INFINITY), V(-INFINITY, -INFINITY, -INFINITY)};
    for (size_t i = 0; i < nv; ++i) {
        v3 p = s->positions[i];
        bb.min = V(min2(bb.min.x, p.x), min2(bb.min.y, p.y), min2(bb.min.z, p.z));
        bb.max = V(max2(bb.max.x, p.x), max2(bb.max.y, p.y), max2(bb.max.z, p.z));
    }
    if (inflateMode == 0) {                                            /* src/main.cpp:207-208 */
        bb.min = scale3(bb.min, 1.001f); bb.max = scale3(bb.max, 1.001f);
    }
    else {
        v3 ext = sub3(bb.max, bb.min);
        float pad = 0.001f * max2(ext.x, max2(ext.y, ext.z));
        bb.min = sub3(bb.min, V(pad, pad, pad)); bb.max = add3(bb.max, V(pad, pad, pad));
    }
    s->rootBox = bb;
    return s;
}

/* The 8-wide copies of every leaf's triangles (Leaf::tris, src/bsp.hpp:165; unused lanes are zero). */
static void build_blocks(orc_scene* s) {
    uint32_t total = 0;
    for (uint32_t l = 0; l < s->leafCount; ++l) { s->leaves[l].firstBlock = total; total += (s->leaves[l].count + 7) / 8; }
    s->blocks = (tri_block*)aligned_alloc(32, sizeof(tri_block) * (size_t)(total ? total : 1));
    memset(s->blocks, 0, sizeof(tri_block) * (size_t)(total ? total : 1));
    s->blockCount = total;
    for (uint32_t l = 0; l < s->leafCount; ++l)
        for (uint32_t k = 0; k < s->leaves[l].count; ++k) {
            const float* src = (const float*)&s->refTris[s->leaves[l].first + k];
            tri_block* b = &s->blocks[s->leaves[l].firstBlock + k / 8];
            for (int c = 0; c < 12; ++c) b->c[c][k % 8] = src[c];
        }
}

void orc_scene_set_simd_leaves(orc_scene* s, int enable) { s->simdLeaves = enable ? 1 : 0; }

orc_scene* orc_scene_create(const float* positions, const float* normals, uint32_t vertexCount,
                            const uint32_t* tris, uint32_t triCount,
                            const uint32_t* meshVertexRanges, const uint32_t* meshTriRanges, uint32_t meshCount,
                            const float* materials, uint32_t materialCount,
                            const float* transforms, const uint32_t* modelMeshes, const uint32_t* modelMaterials,
                            uint32_t modelCount, const float* camera, uint32_t width, uint32_t height,
                            int inflateMode) {
    orc_scene* s = scene_prepare(positions, normals, vertexCount, tris, triCount, meshVertexRanges, meshTriRanges,
                                 meshCount, materials, materialCount, transforms, modelMeshes, modelMaterials,
                                 modelCount, camera, width, height, inflateMode);
    s->root = create_node(s, s->rootBox, 0, 0);
    if (s->buildFailed) { orc_scene_destroy(s); return NULL; }
    build_blocks(s);
    return s;
}

orc_scene* orc_scene_create_with_tree(const float* positions, const float* normals, uint32_t vertexCount,
                                      const uint32_t* tris, uint32_t triCount,
                                      const uint32_t* meshVertexRanges, const uint32_t* meshTriRanges, uint32_t meshCount,
                                      const float* materials, uint32_t materialCount,
                                      const float* transforms, const uint32_t* modelMeshes,
                                      const uint32_t* modelMaterials, uint32_t modelCount, const float* camera,
                                      uint32_t width, uint32_t height, int inflateMode,
                                      int32_t rootIndex, const void* inodes60, uint32_t inodeCount,
                                      const uint32_t* leafFirstRef, const uint32_t* leafRefs, uint32_t leafCount) {
    orc_scene* s = scene_prepare(positions, normals, vertexCount, tris, triCount, meshVertexRanges, meshTriRanges,
                                 meshCount, materials, materialCount, transforms, modelMeshes, modelMaterials,
                                 modelCount, camera, width, height, inflateMode);
    s->root.box = s->rootBox;
    s->root.index = rootIndex;
    s->inodes = (bsp_inode*)calloc(inodeCount ? inodeCount : 1, sizeof(bsp_inode));
    s->inodeCount = s->inodeCap = inodeCount;
    for (uint32_t i = 0; i < inodeCount; ++i) {
        const unsigned char* rec = (const unsigned char*)inodes60 + 60 * (size_t)i;   /* src/bsp.hpp:148-159 layout */
        memcpy(&s->inodes[i].neg.box, rec, 24); memcpy(&s->inodes[i].neg.index, rec + 24, 4);
        memcpy(&s->inodes[i].pos.box, rec + 28, 24); memcpy(&s->inodes[i].pos.index, rec + 52, 4);
        s->inodes[i].axis = rec[56];
    }
    s->leaves = (bsp_leaf*)calloc(leafCount ? leafCount : 1, sizeof(bsp_leaf));
    s->leafCount = s->leafCap = leafCount;
    uint32_t refs = leafFirstRef[leafCount];
    s->refCount = s->refCap = refs;
    s->refMesh = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(refs ? refs : 1));
    s->refTri = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(refs ? refs : 1));
    s->refTris = (pretri*)malloc(sizeof(pretri) * (size_t)(refs ? refs : 1));
    for (uint32_t l = 0; l < leafCount; ++l) {
        /* Leaves above Leaf::MAX_TRIS (src/bsp.hpp:162) cannot exist in the reference; the scan below
         * extends to them unchanged (ascending order, strict <). */
        s->leaves[l].first = leafFirstRef[l];
        s->leaves[l].count = leafFirstRef[l + 1] - leafFirstRef[l];
    }
    for (uint32_t r = 0; r < refs; ++r) {
        uint32_t mesh = leafRefs[2 * (size_t)r], tri = leafRefs[2 * (size_t)r + 1];
        s->refMesh[r] = mesh; s->refTri[r] = tri;
        s->refTris[r] = s->pretris[s->triRanges[2 * mesh] + tri];
    }
    build_blocks(s);
    return s;
}

void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    free(s->baseTris); free(s->meshTriRanges); free(s->modelMeshes); free(s->positions); free(s->normals);
    free(s->vertexRanges); free(s->pretris); free(s->triRanges); free(s->modelMaterials);
    free(s->inodes); free(s->leaves); free(s->refMesh); free(s->refTri); free(s->refTris); free(s->blocks);
    free(s->bvhNodes); free(s->bvhTris); free(s->bvhMesh); free(s->bvhTri); free(s->bvhGlobal); free(s);
}

void orc_scene_counts(const orc_scene* s, uint32_t* c) {
    c[0] = s->modelCount; c[1] = s->vertexCount; c[2] = s->triCount; c[3] = s->inodeCount; c[4] = s->leafCount;
    c[5] = s->baseVertexCount; c[6] = s->baseTriCount; c[7] = s->meshCount; c[8] = s->materialCount;
}

void orc_get_preprocessed(const orc_scene* s, float* positions, float* normals, uint32_t* vertexRanges,
                          float* pretris, uint32_t* triRanges, float* modelMaterials, float* cameraPosition,
                          float* pixelToRay, float* rootBox) {
    memcpy(positions, s->positions, sizeof(v3) * (size_t)s->vertexCount);
    memcpy(normals, s->normals, sizeof(v3) * (size_t)s->vertexCount);
    memcpy(vertexRanges, s->vertexRanges, 8 * (size_t)s->modelCount);
    memcpy(pretris, s->pretris, sizeof(pretri) * (size_t)s->triCount);
    memcpy(triRanges, s->triRanges, 8 * (size_t)s->modelCount);
    for (uint32_t i = 0; i < s->modelCount; ++i) {
        const premat* m = &s->modelMaterials[i];
        float v[12] = {m->ndfAlphaSq, m->geometryAlphaSq, m->f0.x, m->f0.y, m->f0.z,
                       m->adjustedColour.x, m->adjustedColour.y, m->adjustedColour.z,
                       m->emission.x, m->emission.y, m->emission.z, 0.0f};
        memcpy(modelMaterials + 12 * (size_t)i, v, sizeof v);
    }
    st3(cameraPosition, s->cameraPosition);
    memcpy(pixelToRay, s->pixelToRay, 36);
    memcpy(rootBox, &s->rootBox, 24);
}

void orc_get_camera(const orc_scene* s, float* cameraPosition, float* pixelToRay) {
    st3(cameraPosition, s->cameraPosition);
    memcpy(pixelToRay, s->pixelToRay, 36);
}

void orc_get_bsp(const orc_scene* s, float* rootBox, int32_t* rootIndex, float* inodeBoxes, int32_t* inodeInts,
                 uint32_t* leafCounts, uint32_t* leafRefs) {
    memcpy(rootBox, &s->root.box, 24);
    *rootIndex = s->root.index;
    for (uint32_t i = 0; i < s->inodeCount; ++i) {
        memcpy(inodeBoxes + 12 * (size_t)i, &s->inodes[i].neg.box, 24);
        memcpy(inodeBoxes + 12 * (size_t)i + 6, &s->inodes[i].pos.box, 24);
        inodeInts[3 * (size_t)i] = s->inodes[i].neg.index;
        inodeInts[3 * (size_t)i + 1] = s->inodes[i].pos.index;
        inodeInts[3 * (size_t)i + 2] = s->inodes[i].axis;
    }
    for (uint32_t i = 0; i < s->leafCount; ++i) {
        const bsp_leaf* l = &s->leaves[i];
        leafCounts[i] = l->count;
        for (uint32_t j = 0; j < 32; ++j) {
            leafRefs[((size_t)i * 32 + j) * 2] = j < l->count ? s->refMesh[l->first + j] : 0xFFFFFFFFu;
            leafRefs[((size_t)i * 32 + j) * 2 + 1] = j < l->count ? s->refTri[l->first + j] : 0xFFFFFFFFu;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Nearest intersection, src/bsp.hpp:50-145                                                     */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float t, u, v; v3 point; uint32_t mesh, tri; } hit_rec;   /* src/bsp.hpp:23-29 */

typedef struct {
    const orc_scene* s; v3 o, d; float tMin; int frontOnly; orc_stats* st;
} traverser;

/* One lane of lineTrisIntersection<Surfaces>, src/geometry.hpp:99-132. */
static inline int line_tri(const pretri* p, v3 o, v3 d, int frontOnly, float* t, float* u, float* v) {
    float negDet = simd_dot(p->normal, d);
    float invDet = -1.0f / negDet;
    v3 AO = sub3(o, p->v1);
    *t = simd_dot(AO, p->normal) * invDet;
    v3 DAO = simd_cross(AO, d);
    *u = simd_dot(p->v1ToV3, DAO) * invDet;
    *v = (0.0f - simd_dot(p->v1ToV2, DAO)) * invDet;     /* unary minus is 0 - x, vectorised.hpp:269-271 */
    int detCheck = frontOnly ? (negDet <= -1e-6f) : (fabsf(negDet) >= 1e-6f);
    return detCheck && (*u >= 0.0f) && (*v >= 0.0f) && (*u + *v <= 1.0f);
}

/* lineTrisIntersection<Surfaces> for one block of 8 triangles, src/geometry.hpp:117-132, with the AVX2/FMA
 * operations of src/utility/vectorised.hpp:405-481: dot = fmadd(z,z, fmadd(y,y, x*x)), cross component =
 * fmsub(a.y,b.z, a.z*b.y), unary minus = 0 - x.  Lane k gets the bits line_tri() gives triangle k. */
static inline __m256 dot8(__m256 ax, __m256 ay, __m256 az, __m256 bx, __m256 by, __m256 bz) {
    return _mm256_fmadd_ps(az, bz, _mm256_fmadd_ps(ay, by, _mm256_mul_ps(ax, bx)));
}
static inline int line_tris8(const tri_block* b, v3 o, v3 d, int frontOnly, float* t, float* u, float* v) {
    __m256 const nx = _mm256_load_ps(b->c[0]), ny = _mm256_load_ps(b->c[1]), nz = _mm256_load_ps(b->c[2]);
    __m256 const dx = _mm256_set1_ps(d.x), dy = _mm256_set1_ps(d.y), dz = _mm256_set1_ps(d.z);
    __m256 const negDet = dot8(nx, ny, nz, dx, dy, dz);
    __m256 const invDet = _mm256_div_ps(_mm256_set1_ps(-1.0f), negDet);
    __m256 const aox = _mm256_sub_ps(_mm256_set1_ps(o.x), _mm256_load_ps(b->c[3]));
    __m256 const aoy = _mm256_sub_ps(_mm256_set1_ps(o.y), _mm256_load_ps(b->c[4]));
    __m256 const aoz = _mm256_sub_ps(_mm256_set1_ps(o.z), _mm256_load_ps(b->c[5]));
    __m256 const tt = _mm256_mul_ps(dot8(aox, aoy, aoz, nx, ny, nz), invDet);
    __m256 const cx = _mm256_fmsub_ps(aoy, dz, _mm256_mul_ps(aoz, dy));
    __m256 const cy = _mm256_fmsub_ps(aoz, dx, _mm256_mul_ps(aox, dz));
    __m256 const cz = _mm256_fmsub_ps(aox, dy, _mm256_mul_ps(aoy, dx));
    __m256 const uu = _mm256_mul_ps(dot8(_mm256_load_ps(b->c[6]), _mm256_load_ps(b->c[7]), _mm256_load_ps(b->c[8]), cx, cy, cz), invDet);
    __m256 const vv = _mm256_mul_ps(_mm256_sub_ps(_mm256_setzero_ps(),
                                                  dot8(_mm256_load_ps(b->c[9]), _mm256_load_ps(b->c[10]), _mm256_load_ps(b->c[11]), cx, cy, cz)),
                                    invDet);
    __m256 const zero = _mm256_setzero_ps();
    __m256 det;
    if (frontOnly) det = _mm256_cmp_ps(negDet, _mm256_set1_ps(-1e-6f), _CMP_LE_OQ);
    else det = _mm256_cmp_ps(_mm256_andnot_ps(_mm256_set1_ps(-0.0f), negDet), _mm256_set1_ps(1e-6f), _CMP_GE_OQ);
    __m256 const ok = _mm256_and_ps(_mm256_and_ps(det, _mm256_cmp_ps(uu, zero, _CMP_GE_OQ)),
                                    _mm256_and_ps(_mm256_cmp_ps(vv, zero, _CMP_GE_OQ),
                                                  _mm256_cmp_ps(_mm256_add_ps(uu, vv), _mm256_set1_ps(1.0f), _CMP_LE_OQ)));
    _mm256_storeu_ps(t, tt); _mm256_storeu_ps(u, uu); _mm256_storeu_ps(v, vv);
    return _mm256_movemask_ps(ok);
}

/* src/bsp.hpp:58-104 */
static int visit_leaf(const traverser* tr, const box3* box, const bsp_leaf* leaf, hit_rec* out) {
    hit_rec best; best.t = INFINITY;
    int has = 0;
    const orc_scene* s = tr->s;
    if (tr->st) { tr->st->leafVisits++; tr->st->trisTested += leaf->count; }
    if (s->simdLeaves) {
        /* :61-69 blocks of 8, then :73-97 the ascending scan over the stored results */
        for (uint32_t base = 0; base < leaf->count; base += 8) {
            float t8[8], u8[8], v8[8];
            int mask = line_tris8(&s->blocks[leaf->firstBlock + base / 8], tr->o, tr->d, tr->frontOnly, t8, u8, v8);
            uint32_t n = leaf->count - base < 8 ? leaf->count - base : 8;
            mask &= (1 << n) - 1;
            while (mask) {
                int k = __builtin_ctz((unsigned)mask);
                mask &= mask - 1;
                float t = t8[k];
                if (t < best.t && t >= tr->tMin) {
                    v3 p = V(tr->o.x + t * tr->d.x, tr->o.y + t * tr->d.y, tr->o.z + t * tr->d.z);
                    int inBox = p.x >= box->min.x && p.x <= box->max.x && p.y >= box->min.y && p.y <= box->max.y
                             && p.z >= box->min.z && p.z <= box->max.z;
                    if (inBox) {
                        best.t = t; best.u = u8[k]; best.v = v8[k]; best.point = p;
                        best.mesh = s->refMesh[leaf->first + base + (uint32_t)k]; best.tri = s->refTri[leaf->first + base + (uint32_t)k];
                        has = 1;
                    }
                }
            }
        }
        if (has) *out = best;
        return has;
    }
    for (uint32_t i = 0; i < leaf->count; ++i) {
        float t, u, v;
        if (line_tri(&s->refTris[leaf->first + i], tr->o, tr->d, tr->frontOnly, &t, &u, &v)) {
            if (t < best.t && t >= tr->tMin) {
                /* Line::operator(), src/geometry.hpp:30-32: separate multiply and add. */
                v3 p = V(tr->o.x + t * tr->d.x, tr->o.y + t * tr->d.y, tr->o.z + t * tr->d.z);
                int inBox = p.x >= box->min.x && p.x <= box->max.x && p.y >= box->min.y && p.y <= box->max.y
                         && p.z >= box->min.z && p.z <= box->max.z;
                if (inBox) {
                    best.t = t; best.u = u; best.v = v; best.point = p;
                    best.mesh = s->refMesh[leaf->first + i]; best.tri = s->refTri[leaf->first + i];
                    has = 1;
                }
            }
        }
    }
    if (has) *out = best;
    return has;
}

static int visit_node(const traverser* tr, const bsp_node* node, hit_rec* out);

/* src/bsp.hpp:119-141 */
static int visit_inode(const traverser* tr, const bsp_inode* in, hit_rec* out) {
    if (tr->st) tr->st->inodeVisits++;
    float planeToOrigin = axis3(tr->o, in->axis) - axis3(in->pos.box.min, in->axis);
    int positiveNear = planeToOrigin >= 0.0f;
    if (visit_node(tr, positiveNear ? &in->pos : &in->neg, out)) return 1;
    return visit_node(tr, positiveNear ? &in->neg : &in->pos, out);
}

/* src/bsp.hpp:106-117 */
static int visit_node(const traverser* tr, const bsp_node* node, hit_rec* out) {
    if (line_intersects_box(tr->o, tr->d, &node->box, tr->st)) {
        if (node->index > 0) return visit_inode(tr, &tr->s->inodes[node->index - 1], out);
        if (node->index < 0) return visit_leaf(tr, &node->box, &tr->s->leaves[-(node->index + 1)], out);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* The product's alternative accelerator.  NOT in the reference: a restatement of the library's */
/* own 4-wide bounding-volume-hierarchy traversal (rt_device.hip, traceStreamBvh), so that the  */
/* GPU traversal can be checked bit for bit.  Only the triangle test inside it is the           */
/* reference's (line_tri above, src/geometry.hpp:99-132).                                        */
/* ------------------------------------------------------------------------------------------ */

/* Node layout (rt_bvh4_node, 16 words): origin[3], exponents (one byte per axis), qlo.x qlo.y qlo.z qhi.x qhi.y qhi.z
 * (one word each, byte k = child k), two unused words, child[4].  Plane = origin + q * 2^(exponent-127). */
int orc_scene_set_bvh4(orc_scene* s, const float* nodes, uint32_t nodeCount, const uint32_t* triOrder, uint32_t triCount) {
    if (!s || !nodes || nodeCount == 0 || triCount != s->triCount) return -1;
    free(s->bvhNodes); free(s->bvhTris); free(s->bvhMesh); free(s->bvhTri); free(s->bvhGlobal);
    s->bvhGlobal = (uint32_t*)malloc(4 * (size_t)(triCount ? triCount : 1));
    s->bvhNodes = (float*)malloc((size_t)nodeCount * 64);
    s->bvhTris = (pretri*)malloc(sizeof(pretri) * (size_t)(triCount ? triCount : 1));
    s->bvhMesh = (uint32_t*)malloc(4 * (size_t)(triCount ? triCount : 1));
    s->bvhTri = (uint32_t*)malloc(4 * (size_t)(triCount ? triCount : 1));
    if (!s->bvhNodes || !s->bvhTris || !s->bvhMesh || !s->bvhTri || !s->bvhGlobal) return -1;
    memcpy(s->bvhNodes, nodes, (size_t)nodeCount * 64);
    s->bvhNodeCount = nodeCount;
    s->bvhTriCount = triCount;
    for (uint32_t k = 0; k < triCount; ++k) {
        uint32_t g = triOrder[k];
        if (g >= triCount) return -1;
        uint32_t m = 0;
        while (m + 1 < s->modelCount && g >= s->triRanges[2 * (m + 1)]) ++m;
        s->bvhTris[k] = s->pretris[g];
        s->bvhMesh[k] = m;
        s->bvhTri[k] = g - s->triRanges[2 * m];
        s->bvhGlobal[k] = g;
    }
    return 0;
}

/* What is left of the leaf's in-box rule (src/bsp.hpp:82-87) without leaves: the hit point (Line::operator(),
 * src/geometry.hpp:30-32) must lie in the root box, bounds included. */
static inline int inside_root(const orc_scene* s, v3 o, v3 d, float t) {
    v3 p = V(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
    const box3* b = &s->root.box;
    return p.x >= b->min.x && p.x <= b->max.x && p.y >= b->min.y && p.y <= b->max.y && p.z >= b->min.z && p.z <= b->max.z;
}

/* Entry parameter of the line into child k's box, +inf when it misses within [lower, upper].  A crossing parameter
 * is evaluated from the stored byte: t = fma(q, step * r, (origin - o) * r). */
static inline float bvh_child_entry(v3 o, v3 r, const float* n, int k, float lower, float upper) {
    uint32_t w[16];
    memcpy(w, n, 64);
    uint32_t e = w[3];
    float sx, sy, sz;
    uint32_t bx_ = (e & 0xFFu) << 23, by_ = ((e >> 8) & 0xFFu) << 23, bz_ = ((e >> 16) & 0xFFu) << 23;
    memcpy(&sx, &bx_, 4); memcpy(&sy, &by_, 4); memcpy(&sz, &bz_, 4);
    float Ax = sx * r.x, Ay = sy * r.y, Az = sz * r.z;
    float Bx = (n[0] - o.x) * r.x, By = (n[1] - o.y) * r.y, Bz = (n[2] - o.z) * r.z;
    int sh = 8 * k;
    /* per axis the sign of r says which plane the line crosses first */
    uint32_t nearX = r.x < 0.0f ? w[7] : w[4], farX = r.x < 0.0f ? w[4] : w[7];
    uint32_t nearY = r.y < 0.0f ? w[8] : w[5], farY = r.y < 0.0f ? w[5] : w[8];
    uint32_t nearZ = r.z < 0.0f ? w[9] : w[6], farZ = r.z < 0.0f ? w[6] : w[9];
    float tn = fmaxf(fmaxf(fmaf((float)((nearX >> sh) & 0xFFu), Ax, Bx), fmaf((float)((nearY >> sh) & 0xFFu), Ay, By)),
                     fmaxf(fmaf((float)((nearZ >> sh) & 0xFFu), Az, Bz), lower));
    float tf = fminf(fminf(fmaf((float)((farX >> sh) & 0xFFu), Ax, Bx), fmaf((float)((farY >> sh) & 0xFFu), Ay, By)),
                     fminf(fmaf((float)((farZ >> sh) & 0xFFu), Az, Bz), upper));
    return tn <= tf * (1.0f + 0x1p-22f) ? tn : INFINITY;
}

static int bvh_nearest_hit(const orc_scene* s, v3 o, v3 d, float tMin, int frontOnly, hit_rec* out, orc_stats* st) {
    const uint32_t NONE = 0xFFFFFFFFu;
    uint32_t stack[256];
    int sp = 0, found = 0;
    uint32_t cur = 0, bestSlot = 0;
    /* reciprocal direction, held within +-2^80 so that an axis-parallel line gets huge finite crossing parameters of
     * the right sign instead of inf - inf */
    v3 r = V(fabsf(d.x) < 0x1p-80f ? copysignf(0x1p80f, d.x) : 1.0f / d.x,
             fabsf(d.y) < 0x1p-80f ? copysignf(0x1p80f, d.y) : 1.0f / d.y,
             fabsf(d.z) < 0x1p-80f ? copysignf(0x1p80f, d.z) : 1.0f / d.z);
    float lower = fminf(tMin, 0.0f);
    float bestT = INFINITY, bestU = 0.0f, bestV = 0.0f;
    while (cur != NONE) {
        if (!(cur & 0x80000000u)) {
            const float* n = s->bvhNodes + 16 * (size_t)cur;
            uint32_t child[4];
            memcpy(child, n + 12, 16);
            float t[4];
            int nHit = 0;
            if (st) st->inodeVisits++;
            for (int k = 0; k < 4; ++k) {
                t[k] = bvh_child_entry(o, r, n, k, lower, bestT);
                if (child[k] == NONE) t[k] = INFINITY;
                else if (st) st->boxTests++;
                if (t[k] < INFINITY) ++nHit;
            }
            if (nHit == 0) { cur = sp > 0 ? stack[--sp] : NONE; continue; }
            /* position in ascending (entry measured from `lower`, slot) order */
            int pos[4];
            for (int k = 0; k < 4; ++k) {
                pos[k] = 0;
                for (int j = 0; j < 4; ++j)
                    if (j != k && ((t[j] - lower) < (t[k] - lower) || ((t[j] - lower) == (t[k] - lower) && j < k))) ++pos[k];
            }
            for (int k = 0; k < 4; ++k) {
                if (pos[k] == 0) cur = child[k];
                else if (pos[k] < nHit) stack[sp + nHit - 1 - pos[k]] = child[k];
            }
            sp += nHit - 1;
        }
        else {
            uint32_t first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 7u) + 1u;
            if (st) { st->leafVisits++; st->trisTested += count; }
            for (uint32_t j = 0; j < count; ++j) {
                float t, u, v;
                if (!line_tri(&s->bvhTris[first + j], o, d, frontOnly, &t, &u, &v) || !(t >= tMin)) continue;
                if (!inside_root(s, o, d, t)) continue;
                /* equal t: the lower scene-wide triangle index wins */
                if (t < bestT || (t == bestT && found && s->bvhGlobal[first + j] < s->bvhGlobal[bestSlot])) {
                    bestT = t; bestU = u; bestV = v; bestSlot = first + j; found = 1;
                }
            }
            cur = sp > 0 ? stack[--sp] : NONE;
        }
    }
    if (found) {
        out->t = bestT; out->u = bestU; out->v = bestV;
        out->point = V(o.x + bestT * d.x, o.y + bestT * d.y, o.z + bestT * d.z);
        out->mesh = s->bvhMesh[bestSlot]; out->tri = s->bvhTri[bestSlot];
    }
    return found;
}

/* frontOnly: bit 0 = Surfaces::FRONT_ONLY, bit 1 = answer from the product's bounding volume hierarchy */
static int nearest_hit(const orc_scene* s, v3 o, v3 d, float tMin, int frontOnly, hit_rec* out, orc_stats* st) {
    if (frontOnly & 2) return s->bvhNodes ? bvh_nearest_hit(s, o, d, tMin, frontOnly & 1, out, st) : 0;
    traverser tr = {s, o, d, tMin, frontOnly & 1, st};
    return visit_node(&tr, &s->root, out);
}

/* Nearest triangle the reference's test accepts (src/geometry.hpp:99-132) with t >= tMin, by scanning every
 * triangle of the scene; the hit point must lie in the root box (inside_root); among equal t the lowest scene-wide
 * index.  No accelerator at all: the yardstick for
 * the alternative accelerator, whose answer must be this one. */
void orc_intersect_brute(const orc_scene* s, const float* origins, const float* directions, size_t n, float tMin,
                         int frontOnly, float* outFloats, uint32_t* outIds) {
#pragma omp parallel for schedule(dynamic, 16)
    for (long long i = 0; i < (long long)n; ++i) {
        v3 o = ld3(origins + 3 * i), d = ld3(directions + 3 * i);
        float bestT = INFINITY, bestU = 0.0f, bestV = 0.0f;
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t g = 0; g < s->triCount; ++g) {
            float t, u, v;
            if (line_tri(&s->pretris[g], o, d, frontOnly & 1, &t, &u, &v) && t >= tMin && t < bestT && inside_root(s, o, d, t)) {
                bestT = t; bestU = u; bestV = v; best = g;
            }
        }
        float* f = outFloats + 6 * i;
        if (best != 0xFFFFFFFFu) {
            uint32_t m = 0;
            while (m + 1 < s->modelCount && best >= s->triRanges[2 * (m + 1)]) ++m;
            f[0] = bestT; f[1] = bestU; f[2] = bestV;
            st3(f + 3, V(o.x + bestT * d.x, o.y + bestT * d.y, o.z + bestT * d.z));
            outIds[2 * i] = m; outIds[2 * i + 1] = best - s->triRanges[2 * m];
        }
        else {
            for (int k = 0; k < 6; ++k) f[k] = 0.0f;
            outIds[2 * i] = outIds[2 * i + 1] = 0xFFFFFFFFu;
        }
    }
}

void orc_intersect(const orc_scene* s, const float* origins, const float* directions, size_t n, float tMin,
                   int frontOnly, float* outFloats, uint32_t* outIds) {
    for (size_t i = 0; i < n; ++i) {
        hit_rec h;
        float* f = outFloats + 6 * i;
        if (nearest_hit(s, ld3(origins + 3 * i), ld3(directions + 3 * i), tMin, frontOnly, &h, NULL)) {
            f[0] = h.t; f[1] = h.u; f[2] = h.v; st3(f + 3, h.point);
            outIds[2 * i] = h.mesh; outIds[2 * i + 1] = h.tri;
        }
        else {
            for (int k = 0; k < 6; ++k) f[k] = 0.0f;
            outIds[2 * i] = outIds[2 * i + 1] = 0xFFFFFFFFu;
        }
    }
}

/* BSPTree::lineTriNearestIntersection's visitLeaf alone (src/bsp.hpp:58-104): for line i, does the scan of leaf
 * leaves[i] (an index into the tree's leaf array) with the leaf box boxes[6 i ..] (min xyz, max xyz: Node::box of the
 * node that refers to the leaf, src/bsp.hpp:148-153) yield a hit?  out[i] = 1 or 0; outT[i] its t.  Used to check,
 * leaf by leaf, the product's rule for leaving a leaf unscanned. */
void orc_leaf_hits(const orc_scene* s, const uint32_t* leaves, const float* boxes, const float* origins, const float* directions,
                   size_t n, float tMin, int frontOnly, uint8_t* out, float* outT) {
    for (size_t i = 0; i < n; ++i) {
        traverser tr = {s, ld3(origins + 3 * i), ld3(directions + 3 * i), tMin, frontOnly & 1, NULL};
        box3 box = {ld3(boxes + 6 * i), ld3(boxes + 6 * i + 3)};
        hit_rec h;
        h.t = 0.0f;
        out[i] = (leaves[i] < s->leafCount && visit_leaf(&tr, &box, &s->leaves[leaves[i]], &h)) ? 1 : 0;
        if (outT) outT[i] = out[i] ? h.t : 0.0f;
    }
}

/* How many triangles of leaf leaves[i] pass the facing test of lineTrisIntersection<FRONT_ONLY> for direction i:
 * negDet = dot(normal, direction) <= -1e-6 (src/geometry.hpp:118,129-131).  Used to check the product's leaf normal cones. */
void orc_leaf_facing_counts(const orc_scene* s, const uint32_t* leaves, const float* directions, size_t n, uint32_t* out) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t count = 0;
        if (leaves[i] < s->leafCount) {
            const bsp_leaf* leaf = &s->leaves[leaves[i]];
            v3 d = ld3(directions + 3 * i);
            for (uint32_t k = 0; k < leaf->count; ++k) count += simd_dot(s->refTris[leaf->first + k].normal, d) <= -1e-6f;
        }
        out[i] = count;
    }
}

/* src/render.hpp:227: normalize(pixelToRay * vec3{sx, sy, 1}) with glm's mat3*vec3
 * (libraries/glm/glm/detail/type_mat3x3.inl:468-474). */
static inline v3 camera_ray(const float* m, float sx, float sy) {
    v3 r = V(m[0] * sx + m[3] * sy + m[6] * 1.0f,
             m[1] * sx + m[4] * sy + m[7] * 1.0f,
             m[2] * sx + m[5] * sy + m[8] * 1.0f);
    return glm_normalize(r);
}

void orc_camera_rays(const orc_scene* s, const float* sampleXY, size_t n, float* directions) {
    for (size_t i = 0; i < n; ++i) st3(directions + 3 * i, camera_ray(s->pixelToRay, sampleXY[2 * i], sampleXY[2 * i + 1]));
}

/* ------------------------------------------------------------------------------------------ */
/* Shading                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* src/utility/math.hpp:38-54 */
static void orthonormal_basis(v3 vec, v3* p1, v3* p2) {
    v3 vec2 = V(0.56863665f, -0.77215318f, 0.28360506f);
    float dot = glm_dot(vec, vec2);
    if (fabsf(1.0f - fabsf(dot)) < 1e-3f) {
        vec2 = V(0.56863665f, 0.77215318f, 0.28360506f);
        dot = glm_dot(vec, vec2);
    }
    *p1 = glm_normalize(sub3(vec2, scale3(vec, dot)));
    *p2 = glm_cross(vec, *p1);
}

/* Interpolated, un-normalised shading normal, src/render.hpp:121-129. */
static v3 shading_normal(const orc_scene* s, const hit_rec* h) {
    const v3* vn = s->normals + s->vertexRanges[2 * h->mesh];
    const uint32_t* mtr = s->meshTriRanges + 2 * s->modelMeshes[h->mesh];
    const uint32_t* tri = s->baseTris + 3 * ((size_t)mtr[0] + h->tri);
    float c2 = h->u, c3 = h->v;
    float c1 = 1.0f - c2 - c3;
    return add3(add3(scale3(vn[tri[0]], c1), scale3(vn[tri[1]], c2)), scale3(vn[tri[2]], c3));
}

/* One bounce's path weight: the lane-b arithmetic of src/render.hpp:58-83 and :186-193. */
static v3 bounce_weight(const premat* m, float nDotO, float nDotI, float nDotH, float hDotO) {
    float pi = 3.14159265358979323846264338327950288f;
    /* fresnel, :77-83: iPow(1-hDotO, 5) then fnma(f0, tmp, f0 + tmp) */
    float omh = 1.0f - hDotO, tmp = 1.0f;
    for (int i = 0; i < 5; ++i) tmp *= omh;
    v3 F = V(fmaf(-m->f0.x, tmp, m->f0.x + tmp), fmaf(-m->f0.y, tmp, m->f0.y + tmp), fmaf(-m->f0.z, tmp, m->f0.z + tmp));
    /* ndf, :58-63 */
    float nDotHSq = nDotH * nDotH;
    float tanThetaSq = 1.0f / nDotHSq - 1.0f;
    float aPlusTan = m->ndfAlphaSq + tanThetaSq;
    float D = m->ndfAlphaSq / (pi * (nDotHSq * nDotHSq) * (aPlusTan * aPlusTan));
    /* geometry, :66-74 */
    float ag = m->geometryAlphaSq;
    float pI = 1.0f + sqrtf(1.0f + ag / (nDotI * nDotI) - ag);
    float pO = 1.0f + sqrtf(1.0f + ag / (nDotO * nDotO) - ag);
    float G = 4.0f / (pI * pO);
    /* :191-193 */
    float dScale = 4.0f * nDotI * hDotO / (D * nDotH);
    float sScale = G * hDotO / (nDotO * nDotH);
    v3 a = m->adjustedColour;
    v3 diffuse = V(fmaf(-F.x, a.x, a.x) * dScale, fmaf(-F.y, a.y, a.y) * dScale, fmaf(-F.z, a.z, a.z) * dScale);
    v3 spec = V(F.x * sScale, F.y * sScale, F.z * sScale);
    if (nDotO > 0.0f) return add3(diffuse, spec);
    return add3(diffuse, V(0.0f, 0.0f, 0.0f));
}

/* Sample source abstraction: reference mode draws from FastRNG and uses libm; device mode is
 * counter based and uses the polynomial sincos below. */
typedef struct {
    int deviceMode;
    fast_rng* rng;
    uint32_t key[2]; uint32_t pixel, sample;
} sampler;

void orc_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
    /* Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11). */
    uint32_t c0 = counter[0], c1 = counter[1], c2 = counter[2], c3 = counter[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Top 24 bits -> [0,1) exactly representable. */
float orc_u32_to_unit_float(uint32_t v) { return (float)(v >> 8) * 0x1p-24f; }

/* sin and cos of 2*pi*u for u in [0,1): quadrant reduction is exact, then degree-7/8 polynomials
 * on [-pi/4, pi/4] evaluated with explicit fmaf so that the GPU kernel can reproduce every bit. */
void orc_sincos_2pi(float u, float* s, float* c) {
    float x = u * 4.0f;
    float k = rintf(x);
    float r = x - k;
    float a = r * 1.57079632679489661923f;
    float a2 = a * a;
    float sn = fmaf(a * a2, fmaf(a2, fmaf(a2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), a);
    float cp = fmaf(a2, fmaf(a2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
    float cn = fmaf(a2 * a2, cp, fmaf(a2, -0.5f, 1.0f));
    int q = (int)k & 3;
    float ss = (q & 1) ? cn : sn;
    float cc = (q & 1) ? sn : cn;
    if (q == 1) { cc = -cc; }
    else if (q == 2) { ss = -ss; cc = -cc; }
    else if (q == 3) { ss = -ss; }
    *s = ss; *c = cc;
}

/* Two uniform numbers for dimension pair `dim` (0 = pixel jitter, b+1 = bounce b). */
static inline void sampler_pair(sampler* sm, uint32_t dim, float* u0, float* u1raw, float* u2, float* u3) {
    if (!sm->deviceMode) { *u0 = rng_unit(sm->rng); *u1raw = rng_unit(sm->rng); *u2 = 0.0f; *u3 = 0.0f; return; }
    uint32_t ctr[4] = {sm->pixel, sm->sample, dim, 0u}, out[4];
    orc_philox4x32_10(ctr, sm->key, out);
    *u0 = orc_u32_to_unit_float(out[0]);
    *u1raw = orc_u32_to_unit_float(out[1]);
    *u2 = orc_u32_to_unit_float(out[2]);     /* third word of the same block: Russian roulette */
    *u3 = orc_u32_to_unit_float(out[3]);     /* fourth word: choice of the lobe (per-lobe sampling) */
}

/* src/render.hpp:53-212.  The reference stores per-bounce terms in 8 AVX lanes and evaluates them
 * after the loop; the per-lane arithmetic is independent, so evaluating bounce b's weight when it is
 * produced and keeping a running throughput gives the same bits:
 *   lightWeights[k] = w[k-1] * lightWeights[k-1], outgoing = fma(lightWeights[k], emission[k], outgoing). */
static v3 ray_trace(const orc_scene* s, v3 o, v3 d, sampler* sm, uint32_t bounceLimit, float tMin, uint32_t flags, orc_stats* st) {
    int const roulette = (flags & ORC_FLAG_RUSSIAN_ROULETTE) != 0;
    int const lobes = (flags & ORC_FLAG_LOBE_SAMPLING) != 0 && sm->deviceMode;
    int const mode = 1 | ((flags & ORC_FLAG_ACCEL_BVH) ? 2 : 0);
    v3 throughput = V(1.0f, 1.0f, 1.0f), light = V(0.0f, 0.0f, 0.0f);
    uint32_t bounce = 0;
    for (;;) {
        hit_rec h;
        if (st) st->segments++;
        if (!nearest_hit(s, o, d, tMin, mode, &h, st)) break;
        const premat* m = &s->modelMaterials[h.mesh];
        light = V(fmaf(throughput.x, m->emission.x, light.x), fmaf(throughput.y, m->emission.y, light.y),
                  fmaf(throughput.z, m->emission.z, light.z));
        if (bounce >= bounceLimit) break;                                      /* :117 */
        if (st) st->shadedHits++;
        v3 normal = shading_normal(s, &h);
        v3 outgoing = neg3(d);
        float nDotO = glm_dot(normal, outgoing);
        if (nDotO < 0.0f) { nDotO = -nDotO; normal = neg3(normal); }
        v3 p1, p2;
        orthonormal_basis(normal, &p1, &p2);
        float thetaParam, u1, u2, u3;
        sampler_pair(sm, bounce + 1, &thetaParam, &u1, &u2, &u3);
        float sinPhi, cosPhi;
        if (sm->deviceMode) orc_sincos_2pi(u1, &sinPhi, &cosPhi);
        else { float phi = 6.28318530717958647692528676655900576f * u1; sinPhi = sinf(phi); cosPhi = cosf(phi); }
        v3 tangent = add3(scale3(p1, cosPhi), scale3(p2, sinPhi));
        v3 incident, w;
        if (!lobes) {
            float cosThetaSq = 1.0f / (1.0f + m->ndfAlphaSq * thetaParam / (1.0f - thetaParam));
            float cosTheta = sqrtf(cosThetaSq);
            float sinTheta = sqrtf(1.0f - cosThetaSq);
            v3 halfway = add3(scale3(normal, cosTheta), scale3(tangent, sinTheta));
            float hDotO = glm_dot(halfway, outgoing);
            incident = sub3(scale3(halfway, 2.0f * hDotO), outgoing);
            float nDotI = glm_dot(normal, incident);
            if (!(nDotI > 0.0f)) break;                                            /* :169-175 */
            w = bounce_weight(m, nDotO, nDotI, cosTheta, hDotO);
        }
        else {
            /* Per-lobe sampling (device mode only; not in the reference, which draws every direction from the GGX
             * lobe, src/render.hpp:145-167): GGX half vector with probability pS, cosine-weighted direction otherwise,
             * weight = BRDF * cos / (pS pdfS + (1 - pS) pdfD). */
            float pi = 3.14159265358979323846264338327950288f;
            float ws = m->f0.x + m->f0.y + m->f0.z;
            float wd = (m->adjustedColour.x + m->adjustedColour.y + m->adjustedColour.z) * pi;
            float pS = 1.0f;
            if (wd > 0.0f) {
                pS = ws / (ws + wd);
                if (pS < 0.1f) pS = 0.1f;
                if (pS > 0.9f) pS = 0.9f;
            }
            float nDotH, hDotO;
            if (u3 < pS) {
                float cosThetaSq = 1.0f / (1.0f + m->ndfAlphaSq * thetaParam / (1.0f - thetaParam));
                nDotH = sqrtf(cosThetaSq);
                float sinTheta = sqrtf(1.0f - cosThetaSq);
                v3 halfway = add3(scale3(normal, nDotH), scale3(tangent, sinTheta));
                hDotO = glm_dot(halfway, outgoing);
                incident = sub3(scale3(halfway, 2.0f * hDotO), outgoing);
            }
            else {
                float sinT = sqrtf(thetaParam), cosT = sqrtf(1.0f - thetaParam);
                incident = add3(scale3(normal, cosT), scale3(tangent, sinT));
                v3 halfway = glm_normalize(add3(outgoing, incident));
                hDotO = glm_dot(halfway, outgoing);
                nDotH = glm_dot(normal, halfway);
            }
            float nDotI = glm_dot(normal, incident);
            if (!(nDotI > 0.0f && hDotO > 0.0f && nDotH > 0.0f)) break;
            float omh = 1.0f - hDotO, tmp = 1.0f;
            for (int i = 0; i < 5; ++i) tmp *= omh;
            v3 F = V(fmaf(-m->f0.x, tmp, m->f0.x + tmp), fmaf(-m->f0.y, tmp, m->f0.y + tmp), fmaf(-m->f0.z, tmp, m->f0.z + tmp));
            float nDotHSq = nDotH * nDotH;
            float tanThetaSq = 1.0f / nDotHSq - 1.0f;
            float aPlusTan = m->ndfAlphaSq + tanThetaSq;
            float D = m->ndfAlphaSq / (pi * (nDotHSq * nDotHSq) * (aPlusTan * aPlusTan));
            float ag = m->geometryAlphaSq;
            float pI = 1.0f + sqrtf(1.0f + ag / (nDotI * nDotI) - ag);
            float pO = 1.0f + sqrtf(1.0f + ag / (nDotO * nDotO) - ag);
            float G = 4.0f / (pI * pO);
            float pdfS = D * nDotH / (4.0f * hDotO);
            float pdfD = nDotI / pi;
            float pdf = pS * pdfS + (1.0f - pS) * pdfD;
            if (!(pdf > 0.0f)) break;
            float scale = nDotI / pdf;
            float specular = nDotO > 0.0f ? D * G / (4.0f * nDotI * nDotO) : 0.0f;
            v3 a = m->adjustedColour;
            w = V((fmaf(-F.x, a.x, a.x) + F.x * specular) * scale, (fmaf(-F.y, a.y, a.y) + F.y * specular) * scale,
                  (fmaf(-F.z, a.z, a.z) + F.z * specular) * scale);
        }
        throughput = V(w.x * throughput.x, w.y * throughput.y, w.z * throughput.z);
        /* Russian roulette (device mode only; the reference has none, its paths end at the bounce limit,
         * src/render.hpp:117): from the third bounce on, survive with probability min(1, largest throughput
         * component) and divide the throughput by it. */
        if (roulette && bounce >= 2u) {
            float big = throughput.x;
            if (throughput.y > big) big = throughput.y;
            if (throughput.z > big) big = throughput.z;
            float survive = big < 1.0f ? big : 1.0f;
            if (u2 >= survive) break;
            throughput = V(throughput.x / survive, throughput.y / survive, throughput.z / survive);
        }
        o = h.point; d = incident;
        ++bounce;
    }
    return light;
}

void orc_ray_trace(const orc_scene* s, const float* origins, const float* directions, const uint64_t* seeds, size_t n,
                   float* out) {
    for (size_t i = 0; i < n; ++i) {
        fast_rng r = {seeds[i]};
        sampler sm; memset(&sm, 0, sizeof sm); sm.rng = &r;
        st3(out + 3 * i, ray_trace(s, ld3(origins + 3 * i), ld3(directions + 3 * i), &sm, 8, 1e-3f, 0, NULL));
    }
}

void orc_render(const orc_scene* s, uint32_t spp, uint64_t seed, float* image) {
    size_t pixels = (size_t)s->width * s->height;
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t index = 0; index < pixels; ++index) {
        size_t px = index % s->width, py = index / s->width;
        fast_rng r = {seed + index};
        sampler sm; memset(&sm, 0, sizeof sm); sm.rng = &r;
        v3 colour = V(0.0f, 0.0f, 0.0f);
        for (uint32_t i = 0; i < spp; ++i) {
            float sx = (float)px + rng_unit(&r);
            float sy = (float)py + rng_unit(&r);
            v3 d = camera_ray(s->pixelToRay, sx, sy);
            colour = add3(colour, ray_trace(s, s->cameraPosition, d, &sm, 8, 1e-3f, 0, NULL));
        }
        float n = (float)spp;
        st3(image + 3 * index, V(colour.x / n, colour.y / n, colour.z / n));
    }
}

void orc_path_radiance_device_mode(const orc_scene* s, const orc_render_params* p, uint32_t pixelX, uint32_t pixelY,
                                   uint32_t sample, float rgb[3], orc_stats* st) {
    sampler sm; memset(&sm, 0, sizeof sm);
    sm.deviceMode = 1;
    sm.key[0] = (uint32_t)p->seed; sm.key[1] = (uint32_t)(p->seed >> 32);
    sm.pixel = pixelY * p->width + pixelX; sm.sample = sample;
    float jx, jy, unused, unused2;
    sampler_pair(&sm, 0, &jx, &jy, &unused, &unused2);
    float sx = (float)pixelX + jx, sy = (float)pixelY + jy;
    v3 d = camera_ray(p->pixelToRay, sx, sy);
    if (st) st->paths++;
    st3(rgb, ray_trace(s, ld3(p->cameraPosition), d, &sm, p->bounceLimit, p->tMin, p->flags, st));
}

void orc_render_device_mode(const orc_scene* s, const orc_render_params* p, float* image, orc_stats* stats, int threads) {
    orc_stats total; memset(&total, 0, sizeof total);
    int nt = threads > 0 ? threads : 1;
#ifndef _OPENMP
    nt = 1;
#endif
    long long pixels = (long long)p->w * p->h;
#pragma omp parallel num_threads(nt)
    {
        orc_stats local; memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 64)
        for (long long i = 0; i < pixels; ++i) {
            uint32_t px = p->x0 + (uint32_t)(i % p->w), py = p->y0 + (uint32_t)(i / p->w);
            v3 colour = V(0.0f, 0.0f, 0.0f);
            for (uint32_t k = 0; k < p->spp; ++k) {
                float rgb[3];
                orc_path_radiance_device_mode(s, p, px, py, k, rgb, stats ? &local : NULL);
                colour = add3(colour, ld3(rgb));
            }
            float n = (float)p->spp;
            st3(image + 3 * i, V(colour.x / n, colour.y / n, colour.z / n));
        }
#pragma omp critical
        {
            total.segments += local.segments; total.shadedHits += local.shadedHits; total.boxTests += local.boxTests;
            total.trisTested += local.trisTested; total.leafVisits += local.leafVisits;
            total.inodeVisits += local.inodeVisits; total.paths += local.paths;
        }
    }
    if (stats) *stats = total;
}

/* ---- post-processing (src/main.cpp:233-240, src/image.hpp:15-109) ------------------------------- */

/* src/image.hpp:20-27; std::pow(float, float) is libm's powf. */
static float linear_to_srgb(float linear) {
    if (linear <= 0.0031308f) return linear * 12.92f;
    return 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
}

/* Per pixel, in the order of src/main.cpp:233-237: Reinhard (image.hpp:15-17), sRGB transfer
 * (:20-31), NaN -> red (:53-60), Inf -> green (:63-70). */
void orc_tone_srgb_flags(const float* hdr, size_t pixels, float* out) {
    for (size_t i = 0; i < pixels; ++i) {
        float c[3];
        for (int k = 0; k < 3; ++k) {
            float const h = hdr[3 * i + k];
            c[k] = linear_to_srgb(h / (1.0f + h));
        }
        if (isnan(c[0]) || isnan(c[1]) || isnan(c[2])) { c[0] = 1.0f; c[1] = 0.0f; c[2] = 0.0f; }
        if (isinf(c[0]) || isinf(c[1]) || isinf(c[2])) { c[0] = 0.0f; c[1] = 1.0f; c[2] = 0.0f; }
        out[3 * i] = c[0]; out[3 * i + 1] = c[1]; out[3 * i + 2] = c[2];
    }
}

static int cmp_float(const void* a, const void* b) {
    float const x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* medianFilter<1>, src/image.hpp:73-109: per channel, the element of rank neighbours/2 among the
 * in-image neighbours of the 3x3 window.  Interior pixels (9 neighbours) are what the reference
 * computes.  On the image border the reference selects among all 9 array slots although only 4 or 6
 * were written for this pixel (:83-85, :103-105: stale values of earlier pixels, uninitialised for the
 * first ones, in an order that depends on the standard library's nth_element); that is not
 * reproducible and not restated: border pixels here use the written values only. */
void orc_median3(const float* image, uint32_t width, uint32_t height, float* out) {
    for (int64_t i = 0; i < (int64_t)height; ++i)
        for (int64_t j = 0; j < (int64_t)width; ++j) {
            float ch[3][9];
            unsigned n = 0;
            for (int64_t di = -1; di <= 1; ++di)
                for (int64_t dj = -1; dj <= 1; ++dj) {
                    int64_t const y = i + di, x = j + dj;
                    if (y >= 0 && x >= 0 && y < (int64_t)height && x < (int64_t)width) {
                        const float* p = image + 3 * (y * (int64_t)width + x);
                        ch[0][n] = p[0]; ch[1][n] = p[1]; ch[2][n] = p[2];
                        ++n;
                    }
                }
            for (int k = 0; k < 3; ++k) {
                qsort(ch[k], n, sizeof(float), cmp_float);
                out[3 * (i * (int64_t)width + j) + k] = ch[k][n / 2];
            }
        }
}

/* floatTo8BitUInt, src/image.hpp:48-50: clamp(255*x, 0, 255) then conversion to uint8 (truncation). */
void orc_to_u8(const float* image, size_t values, uint8_t* out) {
    for (size_t i = 0; i < values; ++i) {
        float v = 255.0f * image[i];
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);     /* glm::clamp = min(max(x, lo), hi) */
        out[i] = (uint8_t)v;
    }
}

void orc_postprocess(const float* hdr, uint32_t width, uint32_t height, uint8_t* rgb8) {
    size_t const pixels = (size_t)width * height;
    float* a = (float*)malloc(pixels * 12 + 12);
    float* b = (float*)malloc(pixels * 12 + 12);
    orc_tone_srgb_flags(hdr, pixels, a);
    orc_median3(a, width, height, b);
    orc_to_u8(b, pixels * 3, rgb8);
    free(a); free(b);
}
