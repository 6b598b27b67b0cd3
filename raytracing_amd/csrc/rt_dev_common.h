// Device data layout and the geometric predicates of the traversal (box test and its conservative filter, leaf normal cones, leaf triangle bounds).
// A fragment of rt_device.hip: included there, inside its anonymous namespace, after the HIP runtime and the
// library's own headers; it is not a translation unit of its own (the kernels are templates that the host code
// of rt_device.hip instantiates and launches).  Must be compiled with -ffp-contract=off, see rt_device.hip.
#ifndef RT_DEV_COMMON_H
#define RT_DEV_COMMON_H

// ------------------------------------------------------------------------------------------------
// Device data layout
// ------------------------------------------------------------------------------------------------
// Node record, 32 bytes = two float4:  q0 = {lo.x, lo.y, lo.z, hi.x}  q1 = {hi.y, hi.z, index, aux}
//   index > 0 : interior; children are nodes[index] (negative side) and nodes[index+1] (positive);
//               aux = division axis (bits 0-1), negative child empty (bit 2), positive child empty (bit 3).  The split plane is the centre of the node's own box on that
//               axis, which is how BSPTree computes positiveChild.box.min[axis] (src/bsp.hpp:256-274).
//   index < 0 : leaf; triangles are leafTris[-index-1 .. -index-1+count); aux = count (bits 0-7) and a cone around the
//               triangles' normals (bits 8-30, see coneAllBackFacing), or bit 31 and the count in a leaf of 256 or more
//   index == 0: empty
// Leaf triangle record, 48 bytes = three float4, stored once per (leaf, triangle) reference so a leaf's
// triangles are contiguous:  {n.x n.y n.z v1.x} {v1.y v1.z e13.x e13.y} {e13.z e12.x e12.y e12.z}.
// What only shading needs sits in a parallel array, leafIds: {mesh, tri, gtri, 0} with gtri the
// scene-wide index of the instantiated triangle; keeping it out of the records makes the traversal's
// working set a quarter smaller.
// Shading records, 48 bytes per triangle record (same index): {model, vn1.xyz} {vn2.xyz, vn3.x} {vn3.yz, 0, 0} -- the model
// (for the material) and the three instantiated vertex normals, so that shading needs one dependent fetch per hit.
// Material, 48 bytes per model: {ndfAlphaSq geomAlphaSq f0.x f0.y} {f0.z adj.x adj.y adj.z} {em.x em.y em.z 0}.
struct DevScene {
    const float4* nodes;
    const float4* leafTris;
    const uint4* leafIds;
    const float4* shadeRecs;        // per triangle record (same index as leafTris): what shading needs, see below
    const float4* materials;
    float boxAbsMax;       // largest |coordinate| of the root box: see filterMargin
    float rootLo[3], rootHi[3];     // the root box (BSPTree::_root.box), used by the hierarchy traversal
    uint32_t refillMin;             // hierarchy traversal, one-step shape: lanes that gather before results are stored and lines fetched
};

struct Counters {
    unsigned long long boxTests, trisTested, leafVisits, inodeVisits, shadedHits, segments, paths;
    unsigned long long boxFallbacks, boxFilterErrors;
};

struct RenderConst {
    float camPos[3];
    float m[9];
    uint32_t width;
    uint32_t x0, y0, w, h;
    uint32_t bounceLimit;
    float tMin;
    uint32_t key0, key1;
    uint32_t roulette;          // RT_FLAG_RUSSIAN_ROULETTE
    uint32_t lobes;             // RT_FLAG_LOBE_SAMPLING
    uint32_t rowGroup, rowSkip; // interleaved row groups: image row = y0 + r + (r / rowGroup) * rowSkip (rowSkip 0: plain rectangle)
};

// ------------------------------------------------------------------------------------------------
// Device math
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// Unfused sum of products, left to right (glm scalar dot).
__device__ __forceinline__ float dotU(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 crossU(V3 a, V3 b) {
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ V3 normalizeU(V3 a) { return a * (1.0f / sqrtf(dotU(a, a))); }
// Fused forms of the reference's AVX wrappers (src/utility/vectorised.hpp:466-481).
__device__ __forceinline__ float dotF(V3 a, V3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 crossF(V3 a, V3 b) {
    return mk(__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x)));
}

// One face of the reference's six-face box test (src/geometry.hpp:136-212): the line meets the face
// plane at t = numer/denom with |denom| >= 1e-6 and t >= 0, and the point origin + t*direction
// (unfused) lies inside the face rectangle, bounds inclusive.  Written without branches: the six faces
// are evaluated and OR-ed, which is the same boolean as the reference's early-out chain.
__device__ __forceinline__ bool faceHit(float denom, float numer, float oa, float da, float loa, float hia, float ob,
                                        float db, float lob, float hib) {
    float const t = numer / denom;
    float const pa = oa + t * da;
    float const pb = ob + t * db;
    return (fabsf(denom) >= 1e-6f) & (t >= 0.0f) & (pa >= loa) & (pa <= hia) & (pb >= lob) & (pb <= hib);
}

__device__ __forceinline__ bool lineBox(V3 o, V3 d, V3 lo, V3 hi) {
    bool const a = faceHit(-d.x, o.x - lo.x, o.y, d.y, lo.y, hi.y, o.z, d.z, lo.z, hi.z);
    bool const b = faceHit(d.x, hi.x - o.x, o.y, d.y, lo.y, hi.y, o.z, d.z, lo.z, hi.z);
    bool const c = faceHit(-d.y, o.y - lo.y, o.x, d.x, lo.x, hi.x, o.z, d.z, lo.z, hi.z);
    bool const e = faceHit(d.y, hi.y - o.y, o.x, d.x, lo.x, hi.x, o.z, d.z, lo.z, hi.z);
    bool const f = faceHit(-d.z, o.z - lo.z, o.x, d.x, lo.x, hi.x, o.y, d.y, lo.y, hi.y);
    bool const g = faceHit(d.z, hi.z - o.z, o.x, d.x, lo.x, hi.x, o.y, d.y, lo.y, hi.y);
    return a | b | c | e | f | g;
}

// ---- conservative filter in front of the six-face test ---------------------------------------------
// The six-face test is ~110 VALU operations; most (line, box) pairs are nowhere near its decision
// boundary.  boxFilter decides those from the slab crossing times T = plane * r - origin * r, r = RN(1/direction),
// each one fused multiply-add on a product rounded once per line (~30 operations), and leaves the rest to the exact
// test, so the result is always the reference's.
//
// Why a sure decision equals the reference's.  The reference evaluates a face of axis a at
// t~ = RN(RN(numerator)/denominator), which differs from the exact crossing time only by a relative
// error below 2^-22, and accepts it when p~_b = RN(o_b + RN(t~ d_b)) lies in [lo_b, hi_b] on the two other
// axes; p~_b differs from the exact position by at most 2^-22 (|p_b| + |o_b|).  With
// m = 2^-20 (largest |box coordinate| + largest |origin coordinate|), a position at least m inside
// (outside) the slab is therefore accepted (rejected), i.e. in time units a crossing time at least
// m |1/d_b| inside (outside) the slab's crossing interval; mu = m max_k |1/d_k| bounds that for every
// axis.  The T's carry a relative error below 2^-22, covered by the 2^-20 |T| terms, plus 2^-24 |origin * r| from the
// rounded product, a sixteenth of mu at most.
//   surely meets:  the exit face (smallest far crossing time tX) has tX > theta, the line is inside the
//     two other slabs at tX by more than theta (tX - tE > theta and second-smallest far time - tX >
//     theta), and no direction component is below 2^-16 (else mu = inf): that face's test is true.
//   (originInsideRule adds a third sure case.)
//   surely misses: tX < -theta (on some axis the box lies behind the origin by a margin: every face
//     either has t < 0 or lies beyond that slab), or tE - tX > 2 mu + 2^-20 (|tE| + |tX|) (every
//     crossing time is clear of the slab of the latest-entered or of the earliest-left axis).
// tools/box_filter_check.py replays this arithmetic on the CPU against the exact test; kernels built
// with COUNT evaluate both on every box and count disagreements (rt_stats.box_filter_errors).
__device__ __forceinline__ float filterMargin(V3 o, V3 d, V3 r, float boxAbsMax) {
    return rtFilterMargin(o.x, o.y, o.z, d.x, d.y, d.z, r.x, r.y, r.z, boxAbsMax);      // rt_leaf_bounds.h
}

// Bit 0: surely meets; bit 1: surely misses; bit 2: the exit face is clear of the other two slabs' far
// planes (used by the origin-inside rule below).
__device__ __forceinline__ int boxFilter(V3 nor, V3 r, float mu, V3 lo, V3 hi) {
    // nor = -(o * r), rounded once per line: a crossing time is one fused multiply-add, plane * r - o * r.  Its error,
    // 2^-24 |o r| from the rounding of nor, is a sixteenth of the margin mu = 2^-20 (|box| + |o|) max|r| it is judged with.
    float const ax = __fmaf_rn(lo.x, r.x, nor.x), bx = __fmaf_rn(hi.x, r.x, nor.x);
    float const ay = __fmaf_rn(lo.y, r.y, nor.y), by = __fmaf_rn(hi.y, r.y, nor.y);
    float const az = __fmaf_rn(lo.z, r.z, nor.z), bz = __fmaf_rn(hi.z, r.z, nor.z);
    float const tE = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float const xx = fmaxf(ax, bx), xy = fmaxf(ay, by), xz = fmaxf(az, bz);
    float const tX = fminf(fminf(xx, xy), xz);
    float const second = __builtin_amdgcn_fmed3f(xx, xy, xz);
    float const theta = 0x1p-20f * fabsf(tX) + mu;
    bool const exitClear = second - tX > theta;
    bool const sureHit = (tX - tE > theta) & exitClear & (tX > theta);
    float const theta2 = 0x1p-20f * (fabsf(tE) + fabsf(tX)) + (mu + mu);
    bool const sureMiss = (tX < -theta) | (tE - tX > theta2);
    return (sureHit ? 1 : 0) | (sureMiss ? 2 : 0) | (exitClear ? 4 : 0);
}

// Origin-inside rules for pairs the filter left open (typically an origin lying exactly on a plane of
// the box: a camera placed on a split plane of the tree, a path vertex on a wall that coincides with a
// box face).  The origin must be inside the closed box (exact comparisons), and either
//   (a) it lies exactly on one of the box's planes: for that face the reference's numerator is an exact
//       zero, so t = 0 (of either sign; t >= 0 holds) and the point it tests is o + 0*d = o itself, which
//       is inside the face rectangle; or
//   (b) the exit face is clear of the other slabs' far planes: the reference's t for that face is a
//       quotient of two numbers of equal sign, hence >= 0; on the other axes the point o_b + t d_b has
//       moved from o_b towards the far plane, so rounding cannot take it back below the near plane
//       (RN(o_b + x) >= o_b for x >= 0), and it stays below the far plane by the margin exitClear provides.
// Both need every |direction component| >= 1e-6 and finite, which mu < inf guarantees (filterMargin).
__device__ __forceinline__ bool originInsideRule(int filterBits, float mu, V3 o, V3 lo, V3 hi) {
    bool const inside = o.x >= lo.x && o.x <= hi.x && o.y >= lo.y && o.y <= hi.y && o.z >= lo.z && o.z <= hi.z;
    bool const onPlane = o.x == lo.x || o.x == hi.x || o.y == lo.y || o.y == hi.y || o.z == lo.z || o.z == hi.z;
    return inside && mu < INFINITY && (onPlane || (filterBits & 4) != 0);
}

// ---- leaf normal cones ---------------------------------------------------------------------------------
// The triangle test rejects a triangle whose normal does not face the line's origin (Surfaces::FRONT_ONLY:
// normal . direction <= -1e-6, src/geometry.hpp:129-131).  44 % of the leaves a path segment visits hold no triangle
// that passes it.  A leaf record therefore carries a cone that contains the normals of all its triangles -- the axis A
// as three 6-bit signed integers (a point near the sphere of radius 31), and an integer q >= |A| sin(spread + 0.002 rad),
// rounded up, spread being the largest angle between A and a normal -- and the leaf is skipped when
// A . direction >= q |direction|: then every normal makes an angle of at most 90 degrees - 0.002 rad with the direction,
// so normal . direction >= 0.002 |normal| |direction|, which no rounding of the fused dot product (relative error below
// 2^-22) brings down to -1e-6: each triangle of the leaf would have been rejected one by one.  A leaf without a cone has
// A = 0, which never passes.  Integer fields decode in two instructions each, which matters: the test runs in the walking
// loop, for the few lanes of a wave that stand on a leaf.  Kernels that count work test every triangle as the reference
// does and count a disagreement with the cone as an error (rt_stats.box_filter_errors).
//   aux of a leaf: count (bits 0-7), q (8-12), A.z (13-18), A.y (19-24), A.x (25-30); bit 31: 256 triangles or more, the
//   count in bits 0-30 and no cone.
__device__ __forceinline__ uint32_t leafTriCount(uint32_t aux) { return (aux & 0x80000000u) ? (aux & 0x7FFFFFFFu) : (aux & 0xFFu); }

__device__ __forceinline__ bool coneAllBackFacing(uint32_t aux, V3 d) { return rtConeAllBackFacing(aux, d.x, d.y, d.z); }      // rt_leaf_bounds.h

// ---- leaf triangle bounds --------------------------------------------------------------------------------
// A leaf's box is a cell of the spatial subdivision; its triangles usually fill a fraction of it (0.4-0.5 of the volume
// on the benchmark scenes), and half of the leaves a line enters (and whose cone does not already exclude them) it
// leaves without coming near a triangle.  The scene therefore keeps, per leaf, the bounds of the triangles AS THE TEST
// SEES THEM (v1, v1 + v1ToV2, v1 + v1ToV3) clipped to the leaf's box, and a leaf whose bounds the line misses by a margin
// that grows as the line approaches the triangles' planes is not scanned.  The rule and the proof that such a leaf holds
// no hit the reference would accept are in rt_leaf_bounds.h (rtLeafBoundsMiss); `rayOk` is "the line's filter margin is
// finite".  Kernels that count work scan every leaf and count an accepted hit in a leaf the rule would have dropped
// as an error (rt_stats.box_filter_errors).
//
// The record of slot s sits 64 (s + 1) bytes in front of nodes[0] ({lo.xyz, hi.x} {hi.yz, cone word, index} {cK, cE, cN, n0}
// {n1..n4}, the n words describing the normals): the
// bounds are stored in front of the node array, mirrored, so that the kernel needs no second base address.  That matters:
// the traversal kernel for deep trees sits at 80 scalar registers, and with two more its time on a 0.9 M-triangle scene
// goes from 32 to 42 ms whatever the extra registers are used for.
// The record carries copies of the leaf's cone word and index (its first triangle), which are handed out: a leaf that
// is kept is scanned from this record alone.  WITH_CONE: the leaf's normal cone (coneAllBackFacing) is tested in the
// same breath.
// AXES: the form of the scene's bounds records, RT_LEAF_AXES_ONE (one cone) or RT_LEAF_AXES_PACKED (up to three axes; scenes
// with leaves that hold several flat faces).
template <bool WITH_CONE, int AXES>
__device__ __forceinline__ bool missesLeafBounds(const float4* __restrict__ nodes, uint32_t slot, V3 o, V3 d, V3 r, bool rayOk, uint32_t& coneWord, int& leafIndex) {
    const float4* bp = nodes - 4 * (static_cast<size_t>(slot) + 1);
    float4 const b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
    coneWord = __float_as_uint(b1.z);
    leafIndex = __float_as_int(b1.w);
    float const dd = __fmaf_rn(d.x, d.x, __fmaf_rn(d.y, d.y, d.z * d.z));
    float const dlen = sqrtf(dd);
    float const gN = AXES == RT_LEAF_AXES_ONE ? rtLeafGCone(b2.w, b3.x, b3.y, b3.z, b3.w, d.x, d.y, d.z, dd)
                                              : rtLeafGPacked(__float_as_uint(b2.w), __float_as_uint(b3.x), __float_as_uint(b3.y),
                                                              __float_as_uint(b3.z), __float_as_uint(b3.w), d.x, d.y, d.z, dlen);
    bool miss = rtLeafBoundsMiss(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b2.x, b2.y, b2.z, gN, dlen, o.x, o.y, o.z, r.x, r.y, r.z, rayOk);
    if (WITH_CONE) miss = miss || coneAllBackFacing(__float_as_uint(b1.z), d);
    return miss;
}
template <bool WITH_CONE, int AXES>
__device__ __forceinline__ bool missesLeafBounds(const float4* __restrict__ nodes, uint32_t slot, V3 o, V3 d, V3 r, bool rayOk) {
    uint32_t coneWord; int leafIndex;
    return missesLeafBounds<WITH_CONE, AXES>(nodes, slot, o, d, r, rayOk, coneWord, leafIndex);
}

// Diagnostic build only (-DRT_DIAG, make diag): per-phase cycle and lane-occupancy tallies of traceStream.
#ifdef RT_DIAG
__device__ unsigned long long g_diag[16];
#define DIAG_T(var) unsigned long long var = clock64()
#define DIAG_ADD(slot, value) diag[slot] += (value)
// once per wave iteration, by the first lane that is here: iterations and the lanes taking part
#define DIAG_LANES(iterSlot, laneSlot) do { unsigned long long const here_ = __ballot(true); \
    if ((threadIdx.x & 63) == static_cast<unsigned>(__ffsll(static_cast<long long>(here_)) - 1)) { \
        atomicAdd(&g_diag[iterSlot], 1ull); atomicAdd(&g_diag[laneSlot], static_cast<unsigned long long>(__popcll(here_))); } } while (0)
// once per wave, whichever lanes are here
#define DIAG_WAVE(slot) do { if ((threadIdx.x & 63) == static_cast<unsigned>(__ffsll(static_cast<long long>(__ballot(true))) - 1)) atomicAdd(&g_diag[slot], 1ull); } while (0)
#else
#define DIAG_WAVE(slot)
#define DIAG_T(var)
#define DIAG_ADD(slot, value)
#define DIAG_LANES(iterSlot, laneSlot)
#endif

#endif  // RT_DEV_COMMON_H
