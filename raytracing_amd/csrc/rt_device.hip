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

struct HitRec {
    float t, u, v;
    uint32_t ref;       // index of the leaf triangle record, 0xFFFFFFFF = miss
};

// Nearest intersection (src/bsp.hpp:50-145) for a stream of n lines, one lane per line.
//
// The recursion of the reference is unrolled onto a per-lane stack of deferred far children.  Visiting
// order, the per-node box test, the first-leaf-with-a-hit termination and the in-leaf-box acceptance
// test are those of the reference.
//
// Shape: persistent "while-while" with refill.  Each lane alternates between (1) walking interior
// nodes until it stands on a leaf whose box its line meets and (2) testing that leaf's triangles; the
// wave runs the two phases in turn so that each runs with most lanes active, and a lane that has found
// its leaf before the others looks ahead for the next ones instead of idling.  A lane whose line is
// finished stores the result and takes the next line of the stream at the top of the loop, so a wave
// is not held hostage by its longest line.  Lines are handed out in chunks of RT_FETCH_CHUNK per wave
// (dealt statically for most of the stream, through one atomic cursor for its tail); inside a chunk
// lanes take lines by their rank among the idle lanes.
// The stack accesses are written out inline on purpose: wrapping them in by-reference lambdas made the
// compiler turn the wave-uniform exit of the outer loop into a divergent one, at a cost of 27 %.
//   load(i, o, d)        reads line i of the stream
//   store(i, hit, o, d)  receives its nearest hit (hit.ref == 0xFFFFFFFF: none)
#define RT_FETCH_CHUNK 64u
// The persistent kernels deal most of the stream out statically, which assumes that every block of the grid is resident.
// The grid is sized from hipOccupancyMaxActiveBlocksPerMultiprocessor, and that answer does not see everything: on this
// system a wave's scalar registers are allocated with 16 more than the kernel's own (trap handler), so a kernel with
// more than 80 gets seven waves per SIMD where the query and the compiler's report both say eight.  The eighth block
// of every CU then runs after the others and the launch takes 30 % longer (32.1 -> 42.0 ms on a 0.9 M-triangle scene
// for two scalar registers more).  Hence the cap, under which the compiler spills to vector lanes instead.
#define RT_PERSISTENT_SGPRS __attribute__((amdgpu_num_sgpr(82)))          // reported as "TotalSGPRs: 80"
// The BSP traversal kernels are also held to 64 vector registers, i.e. eight waves per SIMD (the compiler reorders the
// leaf bounds test to fit instead of taking 66 and seven waves; nothing is spilt in the kernels that render).
#define RT_EIGHT_WAVES __attribute__((amdgpu_waves_per_eu(8, 8)))
template <bool FRONT_ONLY, bool COUNT, bool SPILL, int AXES, class Load, class Store>
__device__ __forceinline__ void traceStream(const DevScene& S, uint32_t n, uint32_t* __restrict__ nextLine, float tMin,
                                            Counters& c, Load load, Store store) {
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t const lane = threadIdx.x & 63u;
    // The stack of deferred far children: the first RT_LDS_STACK_DEPTH entries live in LDS, one column
    // per lane (entry i of lane l at [i][l]: conflict-free); deeper entries, possible only in trees
    // deeper than that (SPILL), go to scratch memory.
    __shared__ uint32_t ldsStack[RT_LDS_STACK_DEPTH * RT_BLOCK];
    uint32_t spillStack[SPILL ? RT_STACK_MAX - RT_LDS_STACK_DEPTH : 1];
    uint32_t* const stack = ldsStack + threadIdx.x;
    // wave-uniform fetch state.  The first three quarters of the stream are dealt out statically, chunk
    // c to wave c mod waves (concurrent waves then work on neighbouring chunks, which keeps the caches
    // warm, and no atomic is needed); the rest is handed out through the cursor so that no wave is left
    // holding a long tail.  An atomic on the one cursor costs several ns because they serialise.
    uint32_t const waves = gridDim.x * (RT_BLOCK / 64u);
    // Workgroups are handed to the eight XCDs in turn (workgroup b runs on XCD b mod 8).  Waves are numbered
    // so that each XCD works on a contiguous eighth of every round of chunks: rays that are neighbours in
    // the stream then share one XCD's L2 instead of being spread over all eight.
    uint32_t const xcdBlocks = gridDim.x / 8u;
    uint32_t const vblock = (gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * xcdBlocks + blockIdx.x / 8u : blockIdx.x;
    uint32_t const waveId = vblock * (RT_BLOCK / 64u) + (threadIdx.x >> 6);
    uint32_t const staticRounds = static_cast<uint32_t>((static_cast<uint64_t>(n) * 3u / 4u) / (static_cast<uint64_t>(waves) * RT_FETCH_CHUNK));
    uint32_t const staticLines = staticRounds * waves * RT_FETCH_CHUNK;
    uint32_t staticRound = 0;
    uint32_t chunkNext = 0, chunkEnd = 0;
    bool drained = false;
    // per-lane line state
    bool active = false;
    uint32_t line = 0, cur = NONE;
    uint32_t pend1 = NONE, pend2 = NONE, pend3 = NONE, pend4 = NONE;     // leaves (node records) found and not yet scanned, oldest first
    int sp = 0;
    V3 o = mk(0.0f, 0.0f, 0.0f), d = o, r = o;
    float mu = INFINITY;
#ifdef RT_DIAG
    unsigned long long diag[16] = {};
#endif

    for (;;) {
        DIAG_T(t0);
        // ---- refill idle lanes ------------------------------------------------------------------
        unsigned long long idle = __ballot(!active);
        while (idle != 0ull && !drained) {
            if (chunkNext == chunkEnd) {
                uint32_t base = 0;
                if (staticRound < staticRounds) {
                    base = (staticRound * waves + waveId) * RT_FETCH_CHUNK;
                    ++staticRound;
                }
                else {
                    if (lane == 0) base = atomicAdd(nextLine, RT_FETCH_CHUNK);
                    base = staticLines + __builtin_amdgcn_readfirstlane(base);
                }
                if (base >= n) { drained = true; break; }
                chunkNext = base;
                chunkEnd = min(base + RT_FETCH_CHUNK, n);
            }
            uint32_t const avail = chunkEnd - chunkNext;
            uint32_t const rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));      // idle lanes below this one
            if (!active && rank < avail) {
                line = chunkNext + rank;
                load(line, o, d);
                r = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                mu = filterMargin(o, d, r, S.boxAbsMax);
                sp = 0; cur = 0;
                pend1 = NONE; pend2 = NONE; pend3 = NONE; pend4 = NONE;
                active = true;
            }
            chunkNext += min(static_cast<uint32_t>(__popcll(idle)), avail);
            idle = __ballot(!active);
        }
        if (__ballot(active) == 0ull) break;
        DIAG_T(t1);
        DIAG_ADD(0, t1 - t0);
        DIAG_ADD(12, 1); DIAG_ADD(13, __popcll(__ballot(active)));
        if (!active) continue;

        // ---- phase 1: interior nodes until a leaf is reached or nothing is left ------------------
        // A lane that has found its leaf would idle until the slowest lane of the wave has found one
        // too; instead it keeps walking and notes the next leaves its line meets (at most three ahead).
        // If the first leaf yields a hit that work is discarded, otherwise it is already done; the
        // sequence of leaves scanned is unchanged.  Kernels that count work do not look ahead, so
        // that they count exactly the visits of the reference.
        V3 const nor = mk(-(o.x * r.x), -(o.y * r.y), -(o.z * r.z));       // for boxFilter; recomputed here so that it is not live across the scan
        while (cur != NONE && pend4 == NONE) {
            if (!COUNT) { if (__ballot(pend1 == NONE) == 0ull) break; }
            DIAG_LANES(6, 7);
            DIAG_WAVE(5);
            float4 const q0 = S.nodes[2 * cur];
            float4 const q1 = S.nodes[2 * cur + 1];
            V3 const nlo = mk(q0.x, q0.y, q0.z), nhi = mk(q0.w, q1.x, q1.y);
            if (COUNT) c.boxTests++;
            int const index = __float_as_int(q1.z);
            uint32_t const aux = __float_as_uint(q1.w);
            int const filterBits = boxFilter(nor, r, mu, nlo, nhi);
            bool meets = (filterBits & 1) != 0;
            bool decided = (filterBits & 3) != 0;
            if (COUNT || !decided) {
                DIAG_WAVE(14);
                if (!decided && originInsideRule(filterBits, mu, o, nlo, nhi)) { meets = true; decided = true; }
                if (COUNT || !decided) {
                    DIAG_WAVE(15);
                    bool const exact = lineBox(o, d, nlo, nhi);
                    if (COUNT) {
                        if (!decided) c.boxFallbacks++;
                        else if (exact != meets) c.boxFilterErrors++;
                    }
                    meets = exact;
                }
            }
            if (meets && index > 0) {
                if (COUNT) c.inodeVisits++;
                uint32_t const axis = aux & 3u;
                float const la = axis == 0 ? nlo.x : (axis == 1 ? nlo.y : nlo.z);
                float const ha = axis == 0 ? nhi.x : (axis == 1 ? nhi.y : nhi.z);
                float const oa = axis == 0 ? o.x : (axis == 1 ? o.y : o.z);
                float const plane = (la + ha) / 2.0f;
                bool const positiveNear = (oa - plane) >= 0.0f;            // src/bsp.hpp:133
                uint32_t const base = static_cast<uint32_t>(index);
                uint32_t const farChild = positiveNear ? base : base + 1;
                uint32_t const nearChild = positiveNear ? base + 1 : base;
                // Visiting an empty child has no effect whatever its box test says (src/bsp.hpp:106-117 returns
                // nullopt for index 0), so the kernels that do not count work never go there: bits 2 and 3 of
                // aux say which children are empty.
                bool const nearEmpty = !COUNT && (aux & (positiveNear ? 8u : 4u)) != 0u;
                bool const farEmpty = !COUNT && (aux & (positiveNear ? 4u : 8u)) != 0u;
                if (nearEmpty) {
                    if (!farEmpty) { cur = farChild; continue; }
                }
                else {
                    if (!farEmpty) {
                        if (!SPILL || sp < RT_LDS_STACK_DEPTH) stack[sp * RT_BLOCK] = farChild;
                        else spillStack[sp - RT_LDS_STACK_DEPTH] = farChild;
                        ++sp;
                    }
                    cur = nearChild;
                    continue;
                }
                // both children empty: fall through to the pop below with nothing found here
            }
            uint32_t const here = cur;
            if (sp > 0) {                                                   // what follows this node
                --sp;
                // Always an LDS read (clamped index), overridden from scratch in the rare deep case: written as one
                // conditional expression the compiler selects between the two ADDRESSES and emits a flat load,
                // which goes through the texture path and drains both wait counters on every pop.
                cur = stack[(SPILL ? min(sp, RT_LDS_STACK_DEPTH - 1) : sp) * RT_BLOCK];
                if (SPILL && sp >= RT_LDS_STACK_DEPTH) cur = spillStack[sp - RT_LDS_STACK_DEPTH];
            }
            else cur = NONE;
            if (meets && index < 0) {
                // a leaf whose triangles all face away from the origin needs no scan (coneAllBackFacing)
                // (kernels for deep trees test the cone later, together with the leaf's triangle bounds: there the test
                // costs more as part of every walking step than the leaves it keeps out of the list are worth, 26.6 ->
                // 24.6 ms on a 0.9 M-triangle scene; on a 4 k-triangle scene, whose waves are coherent, it is the other
                // way round, 33.0 against 34.8 ms)
                if (COUNT || !FRONT_ONLY || SPILL || !coneAllBackFacing(aux, d)) {
                    if (pend1 == NONE) pend1 = here; else if (pend2 == NONE) pend2 = here; else if (pend3 == NONE) pend3 = here; else pend4 = here;
                }
                if (COUNT) break;
            }
        }

        DIAG_T(t2);
        DIAG_ADD(1, t2 - t1);
        uint32_t leafFirst = 0, leafCount = 0, leafSlot = 0;
        bool coneSkips = false;           // counting kernels: the cone says "skip", to be checked against every triangle
        bool boundsSkip = false;          // counting kernels: the triangle bounds say "skip", to be checked against every accepted hit
        V3 lo = mk(0.0f, 0.0f, 0.0f), hi = lo;
        if (!COUNT) {
            // The leaf to scan is the oldest one noted whose triangles the line does not pass clear of
            // (missesLeafBounds).  Tested here, by all lanes together, and not where a leaf is noted: inside the walking
            // loop the test would be issued for the two or three lanes that stand on a leaf, on almost every iteration.
            // Branch-free per lane (a lane without a leaf tests slot 0 and ignores the answer): every level of divergent
            // control flow holds two scalar registers, which this kernel does not have to spare, see missesLeafBounds.
            // The bounds record repeats the leaf's cone word and triangle index, so the leaf's node record is not needed
            // before a candidate hit is checked against the leaf's box: a lane's last test is the one of the leaf it keeps
            // (a test runs whenever some lane dropped a leaf, and a lane that dropped none repeats its own).
            uint32_t word = 0; int leafIndex = -1;
            bool drop = missesLeafBounds<FRONT_ONLY && SPILL, AXES>(S.nodes, pend1 != NONE ? pend1 : 0u, o, d, r, mu < INFINITY, word, leafIndex) && pend1 != NONE;
            pend1 = drop ? pend2 : pend1; pend2 = drop ? pend3 : pend2; pend3 = drop ? pend4 : pend3; pend4 = drop ? NONE : pend4;
            if (__ballot(drop) != 0ull) {
                drop = missesLeafBounds<FRONT_ONLY && SPILL, AXES>(S.nodes, pend1 != NONE ? pend1 : 0u, o, d, r, mu < INFINITY, word, leafIndex) && pend1 != NONE;
                pend1 = drop ? pend2 : pend1; pend2 = drop ? pend3 : pend2; pend3 = drop ? pend4 : pend3; pend4 = drop ? NONE : pend4;
                if (__ballot(drop) != 0ull) {
                    drop = missesLeafBounds<FRONT_ONLY && SPILL, AXES>(S.nodes, pend1 != NONE ? pend1 : 0u, o, d, r, mu < INFINITY, word, leafIndex) && pend1 != NONE;
                    pend1 = drop ? pend2 : pend1; pend2 = drop ? pend3 : pend2; pend3 = drop ? pend4 : pend3; pend4 = drop ? NONE : pend4;
                    if (__ballot(drop) != 0ull) {
                        drop = missesLeafBounds<FRONT_ONLY && SPILL, AXES>(S.nodes, pend1 != NONE ? pend1 : 0u, o, d, r, mu < INFINITY, word, leafIndex) && pend1 != NONE;
                        pend1 = drop ? pend2 : pend1; pend2 = drop ? pend3 : pend2; pend3 = drop ? pend4 : pend3; pend4 = drop ? NONE : pend4;
                    }
                }
            }
            if (pend1 != NONE) {
                leafFirst = static_cast<uint32_t>(-(leafIndex + 1));
                leafCount = leafTriCount(word);
                leafSlot = pend1;
                pend1 = pend2; pend2 = pend3; pend3 = pend4; pend4 = NONE;
            }
        }
        else if (pend1 != NONE) {
            float4 const q0 = S.nodes[2 * pend1];
            float4 const q1 = S.nodes[2 * pend1 + 1];
            lo = mk(q0.x, q0.y, q0.z); hi = mk(q0.w, q1.x, q1.y);
            leafFirst = static_cast<uint32_t>(-(__float_as_int(q1.z) + 1));
            leafCount = leafTriCount(__float_as_uint(q1.w));
            coneSkips = COUNT && FRONT_ONLY && coneAllBackFacing(__float_as_uint(q1.w), d);
            boundsSkip = COUNT && missesLeafBounds<false, AXES>(S.nodes, pend1, o, d, r, mu < INFINITY);
            pend1 = pend2; pend2 = pend3; pend3 = pend4; pend4 = NONE;
        }

        // ---- phase 2: the leaf's triangles, src/bsp.hpp:58-104 -------------------------------------
        // lineTrisIntersection, src/geometry.hpp:117-132 (one lane of the 8-wide form), in two passes.
        // Pass 1 runs every triangle up to the point where the signs of the barycentric numerators
        // are known and notes the few that can still be hits; pass 2 finishes those in ascending order
        // (division, t, in-box test, strict < against the best so far), which is the order and the
        // arithmetic of the reference's scan.  Splitting keeps the long tail of the test out of the loop
        // that all lanes share: it would otherwise be issued for a handful of lanes on almost every
        // triangle.
        // The search ends with the first leaf that yields an accepted hit (src/bsp.hpp:136-138), so the best hit so far is
        // always "none" when a scan begins: it lives from here to the store below, not across the walk.
        HitRec best;
        best.t = INFINITY; best.u = 0.0f; best.v = 0.0f; best.ref = NONE;
        bool has = false;
        DIAG_T(t3);
        DIAG_ADD(2, t3 - t2);
        if (leafCount != 0) {
            if (COUNT) { c.leafVisits++; c.trisTested += leafCount; }
            for (uint32_t window = 0; window < leafCount; window += 32u) {
                uint32_t const windowEnd = min(leafCount, window + 32u);
                uint32_t candidates = 0;
                float4 aNext = S.leafTris[3 * static_cast<size_t>(leafFirst + window)];
                for (uint32_t j = window; j < windowEnd; ++j) {
                    DIAG_LANES(8, 9);
                    float4 const* tp = S.leafTris + 3 * static_cast<size_t>(leafFirst + j);
                    // The facing test needs only the first 16 bytes of the record; lanes that fail it
                    // do not fetch the rest.  Those 16 bytes are fetched one triangle ahead, so that an iteration waits for
                    // memory once and not twice (the array ends with a spare record for the last fetch ahead).
                    float4 a = aNext;
                    asm volatile("" : "+v"(a.w));       // keep the first fetch one 16-byte load
                    aNext = tp[3];
                    V3 const nrm = mk(a.x, a.y, a.z);
                    float const negDet = dotF(nrm, d);
                    bool const detOk = FRONT_ONLY ? (negDet <= -1e-6f) : (fabsf(negDet) >= 1e-6f);
                    if (!detOk) continue;
                    if (COUNT && coneSkips) c.boxFilterErrors++;
                    if (FRONT_ONLY) {
                        float4 const b = tp[1], e = tp[2];
                        V3 const v1 = mk(a.w, b.x, b.y), e13 = mk(b.z, b.w, e.x), e12 = mk(e.y, e.z, e.w);
                        V3 const AO = o - v1;
                        V3 const DAO = crossF(AO, d);
                        float const du = dotF(e13, DAO);
                        float const dv = dotF(e12, DAO);
                        // u = du*invDet and v = (0-dv)*invDet with 0 < invDet = RN(1/|negDet|).  A negative
                        // numerator gives a negative product unless the product underflows to -0; with
                        // |numerator| > 2^-60 and |negDet| < 2^60 it cannot, so the sign decides without
                        // the division.  Likewise u + v, a sum of two non-negative roundings of
                        // numerator/|negDet| with a relative error below 2^-22, exceeds 1 whenever
                        // du - dv exceeds |negDet| by a factor 1 + 2^-20.
                        bool const tame = negDet > -0x1p60f;
                        if ((du < -0x1p-60f || dv > 0x1p-60f) && tame) continue;
                        if (du >= 0.0f && dv <= 0.0f && tame && negDet < -0x1p-60f && (du - dv) > -negDet * (1.0f + 0x1p-20f)) continue;
                    }
                    candidates |= 1u << (j - window);
                }
                while (candidates != 0u) {
                    DIAG_LANES(10, 11);
                    uint32_t const j = window + static_cast<uint32_t>(__ffs(static_cast<int>(candidates)) - 1);
                    candidates &= candidates - 1u;
                    float4 const* tp = S.leafTris + 3 * static_cast<size_t>(leafFirst + j);
                    float4 const a = tp[0], b = tp[1], e = tp[2];
                    if (!COUNT) {
                        float4 const q0 = S.nodes[2 * leafSlot];
                        float4 const q1 = S.nodes[2 * leafSlot + 1];
                        lo = mk(q0.x, q0.y, q0.z); hi = mk(q0.w, q1.x, q1.y);
                    }
                    V3 const nrm = mk(a.x, a.y, a.z);
                    float const negDet = dotF(nrm, d);
                    V3 const v1 = mk(a.w, b.x, b.y), e13 = mk(b.z, b.w, e.x), e12 = mk(e.y, e.z, e.w);
                    V3 const AO = o - v1;
                    V3 const DAO = crossF(AO, d);
                    float const invDet = -1.0f / negDet;
                    float const u = dotF(e13, DAO) * invDet;
                    float const v = (0.0f - dotF(e12, DAO)) * invDet;
                    if (!(u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f)) continue;
                    float const t = dotF(AO, nrm) * invDet;
                    if (!(t < best.t && t >= tMin)) continue;                   // src/bsp.hpp:79
                    V3 const p = mk(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
                    bool const inBox = p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z
                                    && p.z <= hi.z;
                    if (inBox) {
                        best.t = t; best.u = u; best.v = v; best.ref = leafFirst + j; has = true;
                        if (COUNT && boundsSkip) c.boxFilterErrors++;
                    }
                }
            }
        }
        DIAG_T(t4);
        DIAG_ADD(3, t4 - t3);
        // First leaf with an accepted hit ends the search (src/bsp.hpp:136-138); so does running out of nodes.
        if (has || (cur == NONE && pend1 == NONE)) {
            store(line, best, o, d);
            active = false;
        }
        DIAG_T(t5);
        DIAG_ADD(4, t5 - t4);
    }
#ifdef RT_DIAG
    if (lane == 0)
        for (int k = 0; k < 14; ++k) if ((k < 5 || k > 11) && diag[k]) atomicAdd(&g_diag[k], diag[k]);
#endif
}

// ---- alternative accelerator: 4-wide bounding volume hierarchy ----------------------------------------
// Node record (rt_bvh4_node, 64 bytes = four float4):
//   {origin.x origin.y origin.z exponents} {qlo.x[4] qlo.y[4] qlo.z[4] qhi.x[4]} {qhi.y[4] qhi.z[4] - -} {child[4]}
// A child's box plane on axis a is origin[a] + q * 2^(exponent[a]-127), q the child's byte of the plane's word.
// child: 0xFFFFFFFF none; bit 31: leaf, first triangle in bits 0-27, count-1 in bits 28-30; else a node number.
// Triangle records and ids have the layout of the BSP leaves' (48 and 16 bytes), one per triangle.
//
// nodeEntries gives, for each child, the entry parameter of the line into the child's box, or +inf when the line
// misses the box within [lower, upper] or the child does not exist.  A slab crossing parameter is evaluated
// straight from the stored byte, t = fma(q, step * r, (origin - o) * r) with r the (finite) reciprocal direction; on
// each axis the sign of r says which of the two planes is crossed first, so no minimum or maximum per axis is
// needed.  The far end of the interval is widened by one part in 2^22 so that rounding cannot lose a box the exact
// line touches (the boxes are padded as well).  fminf/fmaxf return the other operand when one is NaN, so a NaN
// never rejects a box.
struct NodeEntries { float t0, t1, t2, t3; uint32_t c0, c1, c2, c3; };

// What is left of the BSP leaf's in-box rule (src/bsp.hpp:82-87) when there are no leaves: the hit point, evaluated as
// Line::operator() does (src/geometry.hpp:30-32), must lie in the scene's root box, bounds included.  The reference
// drops hits on geometry that lies exactly on a face of the root box when the point rounds to just outside it (the
// built-in scene's floor at y = 0: 1.5 % of the rays that go down to it); with this test the hierarchy drops them too.
__device__ __forceinline__ bool insideRoot(const DevScene& S, V3 o, V3 d, float t) {
    float const px = o.x + t * d.x, py = o.y + t * d.y, pz = o.z + t * d.z;
    return px >= S.rootLo[0] && px <= S.rootHi[0] && py >= S.rootLo[1] && py <= S.rootHi[1] && pz >= S.rootLo[2] && pz <= S.rootHi[2];
}

__device__ __forceinline__ NodeEntries nodeEntries(const float4* __restrict__ np, V3 o, V3 r, float lower, float upper) {
    float4 const h = np[0], q1 = np[1], q2 = np[2], cf = np[3];
    uint32_t const e = __float_as_uint(h.w);
    float const Ax = __uint_as_float((e & 0xFFu) << 23) * r.x, Ay = __uint_as_float(((e >> 8) & 0xFFu) << 23) * r.y,
                Az = __uint_as_float(((e >> 16) & 0xFFu) << 23) * r.z;
    float const Bx = (h.x - o.x) * r.x, By = (h.y - o.y) * r.y, Bz = (h.z - o.z) * r.z;
    // per axis, the word holding the planes the line crosses first and the word of those it crosses last: for a
    // direction component below zero the upper planes come first
    uint32_t const lx = __float_as_uint(q1.x), ly = __float_as_uint(q1.y), lz = __float_as_uint(q1.z),
                   hx = __float_as_uint(q1.w), hy = __float_as_uint(q2.x), hz = __float_as_uint(q2.y);
    uint32_t const nearX = r.x < 0.0f ? hx : lx, farX = r.x < 0.0f ? lx : hx;
    uint32_t const nearY = r.y < 0.0f ? hy : ly, farY = r.y < 0.0f ? ly : hy;
    uint32_t const nearZ = r.z < 0.0f ? hz : lz, farZ = r.z < 0.0f ? lz : hz;
    NodeEntries n;
    n.c0 = __float_as_uint(cf.x); n.c1 = __float_as_uint(cf.y); n.c2 = __float_as_uint(cf.z); n.c3 = __float_as_uint(cf.w);
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float const tn = fmaxf(fmaxf(__fmaf_rn(static_cast<float>((nearX >> (8 * k)) & 0xFFu), Ax, Bx),
                                     __fmaf_rn(static_cast<float>((nearY >> (8 * k)) & 0xFFu), Ay, By)),
                               fmaxf(__fmaf_rn(static_cast<float>((nearZ >> (8 * k)) & 0xFFu), Az, Bz), lower));
        float const tf = fminf(fminf(__fmaf_rn(static_cast<float>((farX >> (8 * k)) & 0xFFu), Ax, Bx),
                                     __fmaf_rn(static_cast<float>((farY >> (8 * k)) & 0xFFu), Ay, By)),
                               fminf(__fmaf_rn(static_cast<float>((farZ >> (8 * k)) & 0xFFu), Az, Bz), upper));
        t[k] = tn <= tf * (1.0f + 0x1p-22f) ? tn : INFINITY;
    }
    n.t0 = n.c0 == 0xFFFFFFFFu ? INFINITY : t[0];
    n.t1 = n.c1 == 0xFFFFFFFFu ? INFINITY : t[1];
    n.t2 = n.c2 == 0xFFFFFFFFu ? INFINITY : t[2];
    n.t3 = n.c3 == 0xFFFFFFFFu ? INFINITY : t[3];
    return n;
}

// Nearest accepted triangle over the whole scene for a stream of n lines, one lane per line: the query of
// src/bsp.hpp:50-51 answered from the hierarchy.  Each triangle is tested with the reference's arithmetic
// (src/geometry.hpp:99-132) and accepted when t >= tMin, the hit point lies in the root box (insideRoot) and t < the
// best so far; among equal t the triangle
// with the lowest scene-wide index wins (the order of a BSP leaf's scan, src/bsp.hpp:73-97), so the answer
// does not depend on the visiting order.  Children are visited nearest entry first (ties: lower slot first), the
// others are deferred on the per-lane stack farthest first; a child whose entry lies beyond the best hit
// is not visited.  Same persistent shape as traceStream: lanes walk nodes until each stands on a leaf (or
// has finished), then the wave tests leaf triangles together; finished lanes take the next line.
template <bool FRONT_ONLY, bool COUNT, bool SPILL, bool MERGED, class Load, class Store>
__device__ __forceinline__ void traceStreamBvh(const DevScene& S, uint32_t n, uint32_t* __restrict__ nextLine, float tMin,
                                               Counters& c, Load load, Store store) {
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t const lane = threadIdx.x & 63u;
    __shared__ uint32_t ldsStackBvh[RT_BVH_LDS_STACK * RT_BLOCK];
    uint32_t spillStack[SPILL ? RT_STACK_MAX - RT_BVH_LDS_STACK : 1];
    uint32_t* const stack = ldsStackBvh + threadIdx.x;
    uint32_t const waves = gridDim.x * (RT_BLOCK / 64u);
    uint32_t const xcdBlocks = gridDim.x / 8u;
    uint32_t const vblock = (gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * xcdBlocks + blockIdx.x / 8u : blockIdx.x;
    uint32_t const waveId = vblock * (RT_BLOCK / 64u) + (threadIdx.x >> 6);
    uint32_t const staticRounds = static_cast<uint32_t>((static_cast<uint64_t>(n) * 3u / 4u) / (static_cast<uint64_t>(waves) * RT_FETCH_CHUNK));
    uint32_t const staticLines = staticRounds * waves * RT_FETCH_CHUNK;
    uint32_t staticRound = 0;
    uint32_t chunkNext = 0, chunkEnd = 0;
    bool drained = false;
    bool active = false, finished = false;
    uint32_t line = 0, cur = NONE;
    uint32_t triPos = 0, triEnd = 0;          // triangles of the leaf the lane stands on that are still to test
    int sp = 0;
    V3 o = mk(0.0f, 0.0f, 0.0f), d = o, r = o;
    float const lower = fminf(tMin, 0.0f);
    HitRec best;
    best.t = INFINITY; best.u = 0.0f; best.v = 0.0f; best.ref = NONE;

    for (;;) {
        // ---- refill idle lanes (as in traceStream) ------------------------------------------------
        // Lanes whose line is finished keep its result until enough of them have gathered (or no lane is working):
        // storing results and fetching new lines is then done for many lanes at once instead of for the two or
        // three that finish in any one iteration.
        if (MERGED) {
            unsigned long long const waiting = __ballot(!active || finished);
            if (static_cast<uint32_t>(__popcll(waiting)) >= S.refillMin || waiting == ~0ull) {
                if (active && finished) {
                    store(line, best, o, d);
                    active = false;
                    finished = false;
                }
            }
        }
        unsigned long long idle = __ballot(!active);
        while (idle != 0ull && !drained) {
            if (chunkNext == chunkEnd) {
                uint32_t base = 0;
                if (staticRound < staticRounds) {
                    base = (staticRound * waves + waveId) * RT_FETCH_CHUNK;
                    ++staticRound;
                }
                else {
                    if (lane == 0) base = atomicAdd(nextLine, RT_FETCH_CHUNK);
                    base = staticLines + __builtin_amdgcn_readfirstlane(base);
                }
                if (base >= n) { drained = true; break; }
                chunkNext = base;
                chunkEnd = min(base + RT_FETCH_CHUNK, n);
            }
            uint32_t const avail = chunkEnd - chunkNext;
            uint32_t const rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));      // idle lanes below this one
            if (!active && rank < avail) {
                line = chunkNext + rank;
                load(line, o, d);
                // Reciprocal direction, held within +-2^80: an axis-parallel line then gets huge finite crossing
                // parameters of the right sign on that axis (its slab either contains the whole line or none of it)
                // instead of inf - inf, which would leave the axis unconstrained and send the line through every box
                // of its column.
                r = mk(fabsf(d.x) < 0x1p-80f ? copysignf(0x1p80f, d.x) : 1.0f / d.x,
                       fabsf(d.y) < 0x1p-80f ? copysignf(0x1p80f, d.y) : 1.0f / d.y,
                       fabsf(d.z) < 0x1p-80f ? copysignf(0x1p80f, d.z) : 1.0f / d.z);
                sp = 0; cur = 0;
                triPos = 0; triEnd = 0;
                best.t = INFINITY; best.u = 0.0f; best.v = 0.0f; best.ref = NONE;
                active = true;
            }
            chunkNext += min(static_cast<uint32_t>(__popcll(idle)), avail);
            idle = __ballot(!active);
        }
        if (__ballot(active) == 0ull) break;
        if (!active || finished) continue;

        if constexpr (MERGED) {
        // ---- one step per lane and iteration: a node visit, or one triangle of the leaf the lane stands on.
        // The two kinds of step share the iteration (each runs under its own lanes), so a lane that has reached
        // a leaf does not wait for the lanes still walking, and the other way round.  The triangle record is
        // fetched before the node step and used after it: one memory latency covers both.
        bool const inLeaf = triPos < triEnd;
        float4 ta = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tb = ta, te = ta;
        if (inLeaf) {
            float4 const* tp = S.leafTris + 3 * static_cast<size_t>(triPos);
            ta = tp[0]; tb = tp[1]; te = tp[2];
        }
        else {
            NodeEntries const ne = nodeEntries(S.nodes + 4 * static_cast<size_t>(cur), o, r, lower, best.t);
            uint32_t const c0 = ne.c0, c1 = ne.c1, c2 = ne.c2, c3 = ne.c3;
            float const t0 = ne.t0, t1 = ne.t1, t2 = ne.t2, t3 = ne.t3;
            if (COUNT) { c.inodeVisits++; c.boxTests += (c0 != NONE) + (c1 != NONE) + (c2 != NONE) + (c3 != NONE); }
            // Order keys: an entry parameter measured from `lower` is never negative, so its bit pattern (sign cleared: -0)
            // orders like its value and "a < b" is the sign of a - b: no compare-and-select chains.
            uint32_t const b0 = __float_as_uint(t0 - lower) & 0x7FFFFFFFu, b1 = __float_as_uint(t1 - lower) & 0x7FFFFFFFu,
                           b2 = __float_as_uint(t2 - lower) & 0x7FFFFFFFu, b3 = __float_as_uint(t3 - lower) & 0x7FFFFFFFu;
            int const nHit = static_cast<int>(((b0 - 0x7F800000u) >> 31) + ((b1 - 0x7F800000u) >> 31) + ((b2 - 0x7F800000u) >> 31)
                                              + ((b3 - 0x7F800000u) >> 31));
            if (nHit == 0) {
                if (sp > 0) {
                    --sp;
                    cur = stack[(SPILL ? min(sp, RT_BVH_LDS_STACK - 1) : sp) * RT_BLOCK];
                    if (SPILL && sp >= RT_BVH_LDS_STACK) cur = spillStack[sp - RT_BVH_LDS_STACK];
                }
                else cur = NONE;
            }
            else {
                // position of each child in ascending (entry, slot) order; children that are missed (+inf) come last
                uint32_t const l10 = (b1 - b0) >> 31, l20 = (b2 - b0) >> 31, l30 = (b3 - b0) >> 31, l21 = (b2 - b1) >> 31,
                               l31 = (b3 - b1) >> 31, l32 = (b3 - b2) >> 31;          // lij: child i enters before child j
                int const k0 = static_cast<int>(l10 + l20 + l30);
                int const k1 = static_cast<int>(1u - l10 + l21 + l31);
                int const k2 = static_cast<int>(2u - l20 - l21 + l32);
                int const k3 = static_cast<int>(3u - l30 - l31 - l32);
                // the nearest becomes current, the other hit children go on the stack, farthest lowest
                // Unconditional stores when the LDS part of the stack has room: a child that is not deferred (the
                // nearest, or one the line misses) is written to the first free slot above the new top, where it
                // harms nothing.
                int const top = sp + nHit - 1;
                if (!SPILL || top < RT_BVH_LDS_STACK) {
                    stack[((k0 >= 1 && k0 < nHit) ? top - k0 : top) * RT_BLOCK] = c0;
                    stack[((k1 >= 1 && k1 < nHit) ? top - k1 : top) * RT_BLOCK] = c1;
                    stack[((k2 >= 1 && k2 < nHit) ? top - k2 : top) * RT_BLOCK] = c2;
                    stack[((k3 >= 1 && k3 < nHit) ? top - k3 : top) * RT_BLOCK] = c3;
                }
                else {
#define RT_BVH_DEFER(k, child)                                                                            \
                    if ((k) >= 1 && (k) < nHit) {                                                          \
                        int const pos = top - (k);                                                         \
                        if (pos < RT_BVH_LDS_STACK) stack[pos * RT_BLOCK] = (child);                       \
                        else spillStack[pos - RT_BVH_LDS_STACK] = (child);                                 \
                    }
                    RT_BVH_DEFER(k0, c0)
                    RT_BVH_DEFER(k1, c1)
                    RT_BVH_DEFER(k2, c2)
                    RT_BVH_DEFER(k3, c3)
#undef RT_BVH_DEFER
                }
                cur = k0 == 0 ? c0 : (k1 == 0 ? c1 : (k2 == 0 ? c2 : c3));
                sp += nHit - 1;
            }
        }
        if (inLeaf) {
            // one triangle (src/geometry.hpp:99-132, one lane of the 8-wide form)
            uint32_t const slot = triPos;
            ++triPos;
            V3 const nrm = mk(ta.x, ta.y, ta.z);
            float const negDet = dotF(nrm, d);
            bool ok = FRONT_ONLY ? (negDet <= -1e-6f) : (fabsf(negDet) >= 1e-6f);
            V3 const v1 = mk(ta.w, tb.x, tb.y), e13 = mk(tb.z, tb.w, te.x), e12 = mk(te.y, te.z, te.w);
            V3 const AO = o - v1;
            V3 const DAO = crossF(AO, d);
            float const du = dotF(e13, DAO);
            float const dv = dotF(e12, DAO);
            // signs of the barycentric numerators decide most rejections without the division (same argument
            // as in traceStream's first pass)
            if (FRONT_ONLY && (du < -0x1p-60f || dv > 0x1p-60f) && negDet > -0x1p60f) ok = false;
            if (ok) {
                float const invDet = -1.0f / negDet;
                float const u = du * invDet;
                float const v = (0.0f - dv) * invDet;
                float const t = dotF(AO, nrm) * invDet;
                // equal t (coplanar, overlapping triangles): the lower scene-wide triangle index wins, which makes
                // the result independent of the order in which leaves are visited
                if (u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f && t >= tMin
                    && (t < best.t || (t == best.t && best.ref != NONE && S.leafIds[slot].z < S.leafIds[best.ref].z))
                    && insideRoot(S, o, d, t)) {
                    best.t = t; best.u = u; best.v = v; best.ref = slot;
                }
            }
            if (triPos == triEnd) {                     // leaf finished: next deferred child
                if (sp > 0) {
                    --sp;
                    cur = stack[(SPILL ? min(sp, RT_BVH_LDS_STACK - 1) : sp) * RT_BLOCK];
                    if (SPILL && sp >= RT_BVH_LDS_STACK) cur = spillStack[sp - RT_BVH_LDS_STACK];
                }
                else cur = NONE;
            }
        }
        if (triPos == triEnd) {
            if (cur == NONE) finished = true;           // result stored at the next gathering, see above
            else if (cur & 0x80000000u) {               // standing on a leaf: its triangles are the next steps
                triPos = cur & 0x0FFFFFFFu;
                triEnd = triPos + ((cur >> 28) & 7u) + 1u;
                if (COUNT) { c.leafVisits++; c.trisTested += triEnd - triPos; }
            }
        }
        }
        else {
        // ---- phase 1: interior nodes until the lane stands on a leaf or has nothing left ------------
        while ((cur & 0x80000000u) == 0u) {
            NodeEntries const ne = nodeEntries(S.nodes + 4 * static_cast<size_t>(cur), o, r, lower, best.t);
            uint32_t const c0 = ne.c0, c1 = ne.c1, c2 = ne.c2, c3 = ne.c3;
            float const t0 = ne.t0, t1 = ne.t1, t2 = ne.t2, t3 = ne.t3;
            if (COUNT) { c.inodeVisits++; c.boxTests += (c0 != NONE) + (c1 != NONE) + (c2 != NONE) + (c3 != NONE); }
            // Order keys: an entry parameter measured from `lower` is never negative, so its bit pattern (sign cleared: -0)
            // orders like its value and "a < b" is the sign of a - b: no compare-and-select chains.
            uint32_t const b0 = __float_as_uint(t0 - lower) & 0x7FFFFFFFu, b1 = __float_as_uint(t1 - lower) & 0x7FFFFFFFu,
                           b2 = __float_as_uint(t2 - lower) & 0x7FFFFFFFu, b3 = __float_as_uint(t3 - lower) & 0x7FFFFFFFu;
            int const nHit = static_cast<int>(((b0 - 0x7F800000u) >> 31) + ((b1 - 0x7F800000u) >> 31) + ((b2 - 0x7F800000u) >> 31)
                                              + ((b3 - 0x7F800000u) >> 31));
            if (nHit == 0) {
                if (sp > 0) {
                    --sp;
                    cur = stack[(SPILL ? min(sp, RT_BVH_LDS_STACK - 1) : sp) * RT_BLOCK];
                    if (SPILL && sp >= RT_BVH_LDS_STACK) cur = spillStack[sp - RT_BVH_LDS_STACK];
                }
                else cur = NONE;
                continue;
            }
            // position of each child in ascending (entry, slot) order; children that are missed (+inf) come last
            uint32_t const l10 = (b1 - b0) >> 31, l20 = (b2 - b0) >> 31, l30 = (b3 - b0) >> 31, l21 = (b2 - b1) >> 31,
                           l31 = (b3 - b1) >> 31, l32 = (b3 - b2) >> 31;          // lij: child i enters before child j
            int const k0 = static_cast<int>(l10 + l20 + l30);
            int const k1 = static_cast<int>(1u - l10 + l21 + l31);
            int const k2 = static_cast<int>(2u - l20 - l21 + l32);
            int const k3 = static_cast<int>(3u - l30 - l31 - l32);
            // the nearest becomes current, the other hit children go on the stack, farthest lowest
            // Unconditional stores when the LDS part of the stack has room: a child that is not deferred (the
            // nearest, or one the line misses) is written to the first free slot above the new top, where it
            // harms nothing.
            int const top = sp + nHit - 1;
            if (!SPILL || top < RT_BVH_LDS_STACK) {
                stack[((k0 >= 1 && k0 < nHit) ? top - k0 : top) * RT_BLOCK] = c0;
                stack[((k1 >= 1 && k1 < nHit) ? top - k1 : top) * RT_BLOCK] = c1;
                stack[((k2 >= 1 && k2 < nHit) ? top - k2 : top) * RT_BLOCK] = c2;
                stack[((k3 >= 1 && k3 < nHit) ? top - k3 : top) * RT_BLOCK] = c3;
            }
            else {
#define RT_BVH_DEFER(k, child)                                                                            \
                if ((k) >= 1 && (k) < nHit) {                                                          \
                    int const pos = top - (k);                                                         \
                    if (pos < RT_BVH_LDS_STACK) stack[pos * RT_BLOCK] = (child);                       \
                    else spillStack[pos - RT_BVH_LDS_STACK] = (child);                                 \
                }
                RT_BVH_DEFER(k0, c0)
                RT_BVH_DEFER(k1, c1)
                RT_BVH_DEFER(k2, c2)
                RT_BVH_DEFER(k3, c3)
#undef RT_BVH_DEFER
            }
            cur = k0 == 0 ? c0 : (k1 == 0 ? c1 : (k2 == 0 ? c2 : c3));
            sp += nHit - 1;
        }

        // ---- phase 2: the leaf's triangles (src/geometry.hpp:99-132, one lane of the 8-wide form) -------
        if (cur != NONE) {
            uint32_t const first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 7u) + 1u;
            if (COUNT) { c.leafVisits++; c.trisTested += count; }
            for (uint32_t j = 0; j < count; ++j) {
                float4 const* tp = S.leafTris + 3 * static_cast<size_t>(first + j);
                float4 a = tp[0];
                asm volatile("" : "+v"(a.w));       // keep the first fetch one 16-byte load
                V3 const nrm = mk(a.x, a.y, a.z);
                float const negDet = dotF(nrm, d);
                bool const detOk = FRONT_ONLY ? (negDet <= -1e-6f) : (fabsf(negDet) >= 1e-6f);
                if (!detOk) continue;
                float4 const b = tp[1], e = tp[2];
                V3 const v1 = mk(a.w, b.x, b.y), e13 = mk(b.z, b.w, e.x), e12 = mk(e.y, e.z, e.w);
                V3 const AO = o - v1;
                V3 const DAO = crossF(AO, d);
                float const du = dotF(e13, DAO);
                float const dv = dotF(e12, DAO);
                if (FRONT_ONLY) {
                    // signs of the barycentric numerators decide most rejections without the division
                    // (same argument as in traceStream's first pass)
                    bool const tame = negDet > -0x1p60f;
                    if ((du < -0x1p-60f || dv > 0x1p-60f) && tame) continue;
                }
                float const invDet = -1.0f / negDet;
                float const u = du * invDet;
                float const v = (0.0f - dv) * invDet;
                if (!(u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f)) continue;
                float const t = dotF(AO, nrm) * invDet;
                if (!(t >= tMin)) continue;
                // equal t (coplanar, overlapping triangles): the lower scene-wide triangle index wins, which makes the
                // result independent of the order in which leaves are visited
                if ((t < best.t || (t == best.t && best.ref != NONE && S.leafIds[first + j].z < S.leafIds[best.ref].z))
                    && insideRoot(S, o, d, t)) {
                    best.t = t; best.u = u; best.v = v; best.ref = first + j;
                }
            }
            if (sp > 0) {
                --sp;
                cur = stack[(SPILL ? min(sp, RT_BVH_LDS_STACK - 1) : sp) * RT_BLOCK];
                if (SPILL && sp >= RT_BVH_LDS_STACK) cur = spillStack[sp - RT_BVH_LDS_STACK];
            }
            else cur = NONE;
        }
        if (cur == NONE) {
            store(line, best, o, d);
            active = false;
        }
        }
    }
}

// Philox-4x32-10 keyed by the render seed; counter = (pixel index, sample index, dimension pair, 0).
__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1, uint32_t& r0,
                                       uint32_t& r1, uint32_t& r2, uint32_t& r3) {
    uint32_t c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t const h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t const h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t const n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r0 = c0; r1 = c1; r2 = c2; r3 = c3;
}
__device__ __forceinline__ float unitFloat(uint32_t v) { return static_cast<float>(v >> 8) * 0x1p-24f; }

// sin and cos of 2*pi*u, u in [0,1): exact quadrant reduction, then fixed polynomials on
// [-pi/4, pi/4] evaluated with explicit fused multiply-adds (bit-reproducible on any IEEE machine).
__device__ __forceinline__ void sincos2pi(float u, float& s, float& c) {
    float const x = u * 4.0f;
    float const k = rintf(x);
    float const r = x - k;
    float const a = r * 1.57079632679489661923f;
    float const a2 = a * a;
    float const sn = __fmaf_rn(a * a2, __fmaf_rn(a2, __fmaf_rn(a2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), a);
    float const cp = __fmaf_rn(a2, __fmaf_rn(a2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
    float const cn = __fmaf_rn(a2 * a2, cp, __fmaf_rn(a2, -0.5f, 1.0f));
    int const q = static_cast<int>(k) & 3;
    float ss = (q & 1) ? cn : sn;
    float cc = (q & 1) ? sn : cn;
    if (q == 1) cc = -cc;
    else if (q == 2) { ss = -ss; cc = -cc; }
    else if (q == 3) ss = -ss;
    s = ss; c = cc;
}

// ------------------------------------------------------------------------------------------------
// Kernels
// ------------------------------------------------------------------------------------------------

// Wave-aggregated counter flush: one atomic per wave and counter.
__device__ __forceinline__ unsigned long long waveSum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void flushCounters(const Counters& c, Counters* g) {
    unsigned long long const b = waveSum(c.boxTests), t = waveSum(c.trisTested), l = waveSum(c.leafVisits),
                             i = waveSum(c.inodeVisits), f = waveSum(c.boxFallbacks), e = waveSum(c.boxFilterErrors);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g->boxFallbacks, f);
        atomicAdd(&g->boxFilterErrors, e);
        atomicAdd(&g->boxTests, b);
        atomicAdd(&g->trisTested, t);
        atomicAdd(&g->leafVisits, l);
        atomicAdd(&g->inodeVisits, i);
    }
}

// rt_intersect: one lane per line.  COUNT (rt_intersect_check): the traversal that counts work and cross-checks every
// conservative shortcut against the exact evaluation, as k_trace<true, .> does for a render.
template <bool FRONT_ONLY, bool SPILL, bool COUNT, int AXES>
__global__ __launch_bounds__(RT_BLOCK) RT_PERSISTENT_SGPRS RT_EIGHT_WAVES void k_intersect(DevScene S, const float* __restrict__ origins,
                                                        const float* __restrict__ directions, uint32_t n,
                                                        float tMin, rt_hit* __restrict__ out,
                                                        uint32_t* __restrict__ nextLine, Counters* __restrict__ gc) {
    Counters c = {};
    traceStream<FRONT_ONLY, COUNT, SPILL, AXES>(
        S, n, nextLine, tMin, c,
        [&](uint32_t i, V3& o, V3& d) {
            o = mk(origins[3 * static_cast<size_t>(i)], origins[3 * static_cast<size_t>(i) + 1], origins[3 * static_cast<size_t>(i) + 2]);
            d = mk(directions[3 * static_cast<size_t>(i)], directions[3 * static_cast<size_t>(i) + 1], directions[3 * static_cast<size_t>(i) + 2]);
        },
        [&](uint32_t i, HitRec const& h, V3 o, V3 d) {
            rt_hit r;
            if (h.ref != 0xFFFFFFFFu) {
                uint4 const ids = S.leafIds[h.ref];
                r.t = h.t; r.u = h.u; r.v = h.v;
                r.point[0] = o.x + h.t * d.x; r.point[1] = o.y + h.t * d.y; r.point[2] = o.z + h.t * d.z;
                r.mesh = ids.x; r.tri = ids.y;
            }
            else {
                r.t = 0.0f; r.u = 0.0f; r.v = 0.0f;
                r.point[0] = r.point[1] = r.point[2] = 0.0f;
                r.mesh = 0xFFFFFFFFu; r.tri = 0xFFFFFFFFu;
            }
            out[i] = r;
        });
    if (COUNT) flushCounters(c, gc);
}

template <bool FRONT_ONLY, bool SPILL, bool MERGED>
__global__ __launch_bounds__(RT_BLOCK) RT_PERSISTENT_SGPRS void k_intersect_bvh(DevScene S, const float* __restrict__ origins,
                                                            const float* __restrict__ directions, uint32_t n,
                                                            float tMin, rt_hit* __restrict__ out,
                                                            uint32_t* __restrict__ nextLine) {
    Counters c = {};
    traceStreamBvh<FRONT_ONLY, false, SPILL, MERGED>(
        S, n, nextLine, tMin, c,
        [&](uint32_t i, V3& o, V3& d) {
            o = mk(origins[3 * static_cast<size_t>(i)], origins[3 * static_cast<size_t>(i) + 1], origins[3 * static_cast<size_t>(i) + 2]);
            d = mk(directions[3 * static_cast<size_t>(i)], directions[3 * static_cast<size_t>(i) + 1], directions[3 * static_cast<size_t>(i) + 2]);
        },
        [&](uint32_t i, HitRec const& h, V3 o, V3 d) {
            rt_hit r;
            if (h.ref != 0xFFFFFFFFu) {
                uint4 const ids = S.leafIds[h.ref];
                r.t = h.t; r.u = h.u; r.v = h.v;
                r.point[0] = o.x + h.t * d.x; r.point[1] = o.y + h.t * d.y; r.point[2] = o.z + h.t * d.z;
                r.mesh = ids.x; r.tri = ids.y;
            }
            else {
                r.t = 0.0f; r.u = 0.0f; r.v = 0.0f;
                r.point[0] = r.point[1] = r.point[2] = 0.0f;
                r.mesh = 0xFFFFFFFFu; r.tri = 0xFFFFFFFFu;
            }
            out[i] = r;
        });
}

// Path state, structure of arrays over the batch's path slots (slot = sampleInBatch * pixels + pixel):
//   A = {origin.xyz, dir.x}   B = {dir.y, dir.z, throughput.x, throughput.y}
//   Tz = throughput.z   L = {radiance.xyz, 0}, touched only when a path meets an emitter (and by k_accumulate)
struct PathState {
    float4* A;
    float4* B;
    float* Tz;
    float4* L;
    float4* hit;          // {t, u, v, ref}
    uint32_t* queue[2];   // path slots still alive, ping-pong
    uint32_t* counts;     // counts[b] = number of paths entering bounce b
};

// Path slot p of a batch (slot = sampleInBatch * pixels + p) belongs to the pixel at (x, y) of the rendered
// rectangle.  The part of the rectangle that 8x8 tiles cover is numbered tile by tile, so that the 64 lanes
// of a wave start on an 8x8 block of pixels (their rays stay close together in the tree: fewer distinct
// cache lines per load, fewer idle lanes); the strip right of it and the strip below it follow in row order.
// The pixel's own index y*width+x keys the sampler, so the image does not depend on this numbering.
__device__ __forceinline__ void slotPixel(uint32_t p, uint32_t w, uint32_t h, uint32_t& x, uint32_t& y) {
    uint32_t const w8 = w & ~7u, h8 = h & ~7u;
    uint32_t const tiled = w8 * h8;
    if (p < tiled) {
        uint32_t const tile = p >> 6, in = p & 63u, tilesX = w8 >> 3;
        uint32_t const ty = tile / tilesX, tx = tile - ty * tilesX;
        x = (tx << 3) + (in & 7u);
        y = (ty << 3) + (in >> 3);
        return;
    }
    uint32_t q = p - tiled;
    uint32_t const right = (w - w8) * h;
    if (q < right) {
        uint32_t const rw = w - w8;
        y = q / rw;
        x = w8 + (q - y * rw);
        return;
    }
    q -= right;
    y = h8 + q / w8;
    x = q - (q / w8) * w8;
}

// Slot numbering of a batch of `samples` samples of `pixels` pixels.  Pixels (in slotPixel order) are taken
// in groups of `group`; a group's slots are contiguous, sample by sample: slot = groupBase + sample * groupSize
// + pixelInGroup.  Consecutive 64-line chunks of the stream -- which go to neighbouring waves of one CU -- are
// then successive samples of the SAME few pixels rather than neighbouring pixels of one sample, so that the
// waves resident on a CU keep walking the same part of the tree and re-use each other's cache lines.
struct SlotMap {
    uint32_t pixels, samples, group;     // group >= 1
    __device__ __forceinline__ void split(uint32_t slot, uint32_t& p, uint32_t& sample) const {
        uint32_t const fullGroups = pixels / group;
        uint32_t const full = fullGroups * group * samples;            // slots of the complete groups
        uint32_t g, size, rem;
        if (slot < full) { g = slot / (group * samples); size = group; rem = slot - g * group * samples; }
        else { g = fullGroups; size = pixels - fullGroups * group; rem = slot - full; }
        sample = rem / size;
        p = g * group + (rem - sample * size);
    }
    __device__ __forceinline__ uint32_t slot(uint32_t p, uint32_t sample) const {
        uint32_t const fullGroups = pixels / group;
        uint32_t const g = p / group;
        uint32_t const size = g < fullGroups ? group : pixels - fullGroups * group;
        return g * group * samples + sample * size + (p - g * group);
    }
};

// Camera rays, src/render.hpp:225-228: jittered sample position, direction = normalize(M * (sx, sy, 1)).
__global__ __launch_bounds__(RT_BLOCK) void k_raygen(RenderConst R, PathState P, SlotMap M, uint32_t sampleBase,
                                                     uint32_t paths) {
    uint32_t const stride = gridDim.x * blockDim.x;
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < paths; id += stride) {
        uint32_t sLocal, p;
        M.split(id, p, sLocal);
        uint32_t lx, ly;
        slotPixel(p, R.w, R.h, lx, ly);
        uint32_t const px = R.x0 + lx, py = R.y0 + ly + (R.rowSkip ? (ly / R.rowGroup) * R.rowSkip : 0u);
        uint32_t r0, r1, r2, r3;
        philox(py * R.width + px, sampleBase + sLocal, 0u, R.key0, R.key1, r0, r1, r2, r3);
        float const sx = static_cast<float>(px) + unitFloat(r0);
        float const sy = static_cast<float>(py) + unitFloat(r1);
        V3 const v = mk(R.m[0] * sx + R.m[3] * sy + R.m[6] * 1.0f, R.m[1] * sx + R.m[4] * sy + R.m[7] * 1.0f,
                        R.m[2] * sx + R.m[5] * sy + R.m[8] * 1.0f);
        V3 const d = normalizeU(v);
        P.A[id] = make_float4(R.camPos[0], R.camPos[1], R.camPos[2], d.x);
        P.B[id] = make_float4(d.y, d.z, 1.0f, 1.0f);
        P.Tz[id] = 1.0f;
        P.L[id] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) P.counts[0] = paths;
}

// One closest-hit query per live path.  `bounce` selects the input queue; bounce 0 is the identity.
// counts[RT_FETCH_BASE + bounce] is this launch's line-fetch cursor (zeroed with the counts).
#define RT_FETCH_BASE 512u
template <bool COUNT, bool SPILL, int AXES>
__global__ __launch_bounds__(RT_BLOCK) RT_PERSISTENT_SGPRS RT_EIGHT_WAVES void k_trace(DevScene S, PathState P, uint32_t bounce, float tMin,
                                                    Counters* __restrict__ gc) {
    uint32_t const n = P.counts[bounce];
    uint32_t const* __restrict__ q = P.queue[bounce & 1];
    Counters c = {};
    traceStream<true, COUNT, SPILL, AXES>(
        S, n, P.counts + RT_FETCH_BASE + bounce, tMin, c,
        [&](uint32_t i, V3& o, V3& d) {
            uint32_t const slot = bounce == 0 ? i : q[i];
            float4 const A = P.A[slot], B = P.B[slot];
            o = mk(A.x, A.y, A.z);
            d = mk(A.w, B.x, B.y);
        },
        [&](uint32_t i, HitRec const& h, V3, V3) {
            uint32_t const slot = bounce == 0 ? i : q[i];
            P.hit[slot] = make_float4(h.t, h.u, h.v, __uint_as_float(h.ref));
        });
    if (COUNT) flushCounters(c, gc);
}

// The same query answered from the bounding volume hierarchy (S.nodes, S.leafTris, S.leafIds then describe it).
template <bool COUNT, bool SPILL, bool MERGED>
__global__ __launch_bounds__(RT_BLOCK) RT_PERSISTENT_SGPRS void k_trace_bvh(DevScene S, PathState P, uint32_t bounce, float tMin,
                                                        Counters* __restrict__ gc) {
    uint32_t const n = P.counts[bounce];
    uint32_t const* __restrict__ q = P.queue[bounce & 1];
    Counters c = {};
    traceStreamBvh<true, COUNT, SPILL, MERGED>(
        S, n, P.counts + RT_FETCH_BASE + bounce, tMin, c,
        [&](uint32_t i, V3& o, V3& d) {
            uint32_t const slot = bounce == 0 ? i : q[i];
            float4 const A = P.A[slot], B = P.B[slot];
            o = mk(A.x, A.y, A.z);
            d = mk(A.w, B.x, B.y);
        },
        [&](uint32_t i, HitRec const& h, V3, V3) {
            uint32_t const slot = bounce == 0 ? i : q[i];
            P.hit[slot] = make_float4(h.t, h.u, h.v, __uint_as_float(h.ref));
        });
    if (COUNT) flushCounters(c, gc);
}

// Per-bounce shading: emission pick-up, GGX half-vector sampling, Cook-Torrance weight
// (src/render.hpp:106-175 and the per-lane arithmetic of :58-83, :186-207 applied incrementally),
// then compaction of the surviving paths into the next queue with one atomic per wave.
__global__ __launch_bounds__(RT_BLOCK) void k_shade(DevScene S, RenderConst R, PathState P, uint32_t bounce,
                                                    uint32_t sampleBase, SlotMap M,
                                                    Counters* __restrict__ gc) {
    uint32_t const n = P.counts[bounce];
    uint32_t const* __restrict__ qin = P.queue[bounce & 1];
    uint32_t* __restrict__ qout = P.queue[(bounce + 1) & 1];
    uint32_t const lane = threadIdx.x & 63;
    uint32_t const stride = gridDim.x * blockDim.x;
    uint32_t shaded = 0;
    __shared__ uint32_t stageAll[(RT_BLOCK / 64) * RT_STAGE];
    volatile uint32_t* const stage = stageAll + (threadIdx.x >> 6) * RT_STAGE;   // this wave's staging area
    uint32_t staged = 0;                                                          // wave-uniform
    // The loop bound is wave-uniform so that every lane reaches the ballot.
    // A wave takes 256 consecutive entries (four steps of 64) before it moves on by the grid's stride, so that
    // the runs it appends to the next queue come from neighbouring paths: the next bounce's waves then start
    // on lines with nearby origins.
    uint64_t const tileStride = static_cast<uint64_t>(stride) * 4u;
    uint64_t const waveFirst = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + (threadIdx.x & ~63u)) * 4u;
    for (uint64_t step = 0;; ++step) {
        uint64_t const base64 = (step >> 2) * tileStride + waveFirst + (step & 3u) * 64u;
        if ((step & 3u) == 0u && base64 >= n) break;
        if (base64 >= n) continue;
        uint32_t const base = static_cast<uint32_t>(base64);
        uint32_t const i = base + lane;
        bool alive = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = bounce == 0 ? i : qin[i];
            float4 const H = P.hit[slot];
            uint32_t const ref = __float_as_uint(H.w);
            if (ref != 0xFFFFFFFFu) {
                float4 const A = P.A[slot], B = P.B[slot];
                float const Tz = P.Tz[slot];
                V3 const o = mk(A.x, A.y, A.z), d = mk(A.w, B.x, B.y);
                V3 T = mk(B.z, B.w, Tz);
                float4 const n0 = S.shadeRecs[3 * static_cast<size_t>(ref)];
                uint32_t const mesh = __float_as_uint(n0.x);
                float4 const m0 = S.materials[3 * mesh], m1 = S.materials[3 * mesh + 1], m2 = S.materials[3 * mesh + 2];
                float const ndfA = m0.x, geoA = m0.y;
                V3 const f0 = mk(m0.z, m0.w, m1.x), adj = mk(m1.y, m1.z, m1.w), em = mk(m2.x, m2.y, m2.z);
                // outgoingLight = fma(lightWeight, emission, outgoingLight), src/render.hpp:206.  With a zero emission and
                // a finite throughput the sum is the radiance itself, bit for bit, so the radiance is read and written
                // only at emitters (and where the throughput is no longer finite: 0 * inf must still turn it into NaN).
                bool const finiteT = fabsf(T.x) < INFINITY && fabsf(T.y) < INFINITY && fabsf(T.z) < INFINITY;
                if (em.x != 0.0f || em.y != 0.0f || em.z != 0.0f || !finiteT) {
                    float4 const Lr = P.L[slot];
                    P.L[slot] = make_float4(__fmaf_rn(T.x, em.x, Lr.x), __fmaf_rn(T.y, em.y, Lr.y), __fmaf_rn(T.z, em.z, Lr.z), 0.0f);
                }
                if (bounce < R.bounceLimit) {                                    // src/render.hpp:117
                    ++shaded;
                    float4 const n1 = S.shadeRecs[3 * static_cast<size_t>(ref) + 1], n2 = S.shadeRecs[3 * static_cast<size_t>(ref) + 2];
                    V3 const vn1 = mk(n0.y, n0.z, n0.w), vn2 = mk(n1.x, n1.y, n1.z), vn3 = mk(n1.w, n2.x, n2.y);
                    float const c2 = H.y, c3 = H.z;
                    float const c1 = 1.0f - c2 - c3;
                    V3 normal = vn1 * c1 + vn2 * c2 + vn3 * c3;                  // src/render.hpp:128-129
                    V3 const outgoing = neg(d);
                    float nDotO = dotU(normal, outgoing);
                    if (nDotO < 0.0f) { nDotO = -nDotO; normal = neg(normal); }
                    // orthonormalBasis, src/utility/math.hpp:38-54
                    V3 ref2 = mk(0.56863665f, -0.77215318f, 0.28360506f);
                    float dp = dotU(normal, ref2);
                    if (fabsf(1.0f - fabsf(dp)) < 1e-3f) {
                        ref2 = mk(0.56863665f, 0.77215318f, 0.28360506f);
                        dp = dotU(normal, ref2);
                    }
                    V3 const p1 = normalizeU(ref2 - normal * dp);
                    V3 const p2 = crossU(normal, p1);
                    // GGX sampling of the half vector, src/render.hpp:145-153
                    uint32_t sLocal, p;
                    M.split(slot, p, sLocal);
                    uint32_t lx, ly;
                    slotPixel(p, R.w, R.h, lx, ly);
                    uint32_t const px = R.x0 + lx, py = R.y0 + ly + (R.rowSkip ? (ly / R.rowGroup) * R.rowSkip : 0u);
                    uint32_t r0, r1, r2, r3;
                    philox(py * R.width + px, sampleBase + sLocal, bounce + 1u, R.key0, R.key1, r0, r1, r2, r3);
                    float const thetaParam = unitFloat(r0);
                    float sinPhi, cosPhi;
                    sincos2pi(unitFloat(r1), sinPhi, cosPhi);
                    V3 const tangent = p1 * cosPhi + p2 * sinPhi;
                    float const pi = 3.14159265358979323846264338327950288f;
                    V3 incident = mk(0.0f, 0.0f, 0.0f), w = incident;
                    bool sampled = false;
                    if (R.lobes == 0u) {
                        // the reference's estimator: GGX sampling of the half vector, src/render.hpp:145-153
                        float const cosThetaSq = 1.0f / (1.0f + ndfA * thetaParam / (1.0f - thetaParam));
                        float const cosTheta = sqrtf(cosThetaSq);
                        float const sinTheta = sqrtf(1.0f - cosThetaSq);
                        V3 const halfway = normal * cosTheta + tangent * sinTheta;
                        float const hDotO = dotU(halfway, outgoing);
                        incident = halfway * (2.0f * hDotO) - outgoing;
                        float const nDotI = dotU(normal, incident);
                        if (nDotI > 0.0f) {                                          // src/render.hpp:169
                            // Fresnel-Schlick, src/render.hpp:77-83
                            float const omh = 1.0f - hDotO;
                            float const pw = ((((1.0f * omh) * omh) * omh) * omh) * omh;
                            V3 const F = mk(__fmaf_rn(-f0.x, pw, f0.x + pw), __fmaf_rn(-f0.y, pw, f0.y + pw),
                                            __fmaf_rn(-f0.z, pw, f0.z + pw));
                            // GGX distribution, src/render.hpp:58-63 (nDotH is the sampled cosTheta, :166)
                            float const nDotHSq = cosTheta * cosTheta;
                            float const tanSq = 1.0f / nDotHSq - 1.0f;
                            float const aPlus = ndfA + tanSq;
                            float const D = ndfA / (pi * (nDotHSq * nDotHSq) * (aPlus * aPlus));
                            // Smith-GGX geometry, src/render.hpp:66-74
                            float const gI = 1.0f + sqrtf(1.0f + geoA / (nDotI * nDotI) - geoA);
                            float const gO = 1.0f + sqrtf(1.0f + geoA / (nDotO * nDotO) - geoA);
                            float const G = 4.0f / (gI * gO);
                            // src/render.hpp:191-193
                            float const dScale = 4.0f * nDotI * hDotO / (D * cosTheta);
                            float const sScale = G * hDotO / (nDotO * cosTheta);
                            V3 const diffuse = mk(__fmaf_rn(-F.x, adj.x, adj.x) * dScale, __fmaf_rn(-F.y, adj.y, adj.y) * dScale,
                                                  __fmaf_rn(-F.z, adj.z, adj.z) * dScale);
                            V3 const spec = nDotO > 0.0f ? mk(F.x * sScale, F.y * sScale, F.z * sScale) : mk(0.0f, 0.0f, 0.0f);
                            w = diffuse + spec;
                            sampled = true;
                        }
                    }
                    else {
                        // Per-lobe sampling (RT_FLAG_LOBE_SAMPLING; the reference draws every direction from the GGX
                        // lobe, src/render.hpp:145-167, however diffuse the material): with probability pS the GGX half
                        // vector as above, otherwise a cosine-weighted direction; the weight is BRDF * cos / (pS pdfS +
                        // (1 - pS) pdfD), the one-sample mixture estimator of the same integral.
                        float const ws = f0.x + f0.y + f0.z;
                        float const wd = (adj.x + adj.y + adj.z) * pi;
                        float pS = 1.0f;
                        if (wd > 0.0f) {
                            pS = ws / (ws + wd);
                            if (pS < 0.1f) pS = 0.1f;
                            if (pS > 0.9f) pS = 0.9f;
                        }
                        float nDotH, hDotO;
                        if (unitFloat(r3) < pS) {
                            float const cosThetaSq = 1.0f / (1.0f + ndfA * thetaParam / (1.0f - thetaParam));
                            nDotH = sqrtf(cosThetaSq);
                            float const sinTheta = sqrtf(1.0f - cosThetaSq);
                            V3 const halfway = normal * nDotH + tangent * sinTheta;
                            hDotO = dotU(halfway, outgoing);
                            incident = halfway * (2.0f * hDotO) - outgoing;
                        }
                        else {
                            float const sinT = sqrtf(thetaParam), cosT = sqrtf(1.0f - thetaParam);
                            incident = normal * cosT + tangent * sinT;
                            V3 const halfway = normalizeU(outgoing + incident);
                            hDotO = dotU(halfway, outgoing);
                            nDotH = dotU(normal, halfway);
                        }
                        float const nDotI = dotU(normal, incident);
                        if (nDotI > 0.0f && hDotO > 0.0f && nDotH > 0.0f) {
                            float const omh = 1.0f - hDotO;
                            float const pw = ((((1.0f * omh) * omh) * omh) * omh) * omh;
                            V3 const F = mk(__fmaf_rn(-f0.x, pw, f0.x + pw), __fmaf_rn(-f0.y, pw, f0.y + pw),
                                            __fmaf_rn(-f0.z, pw, f0.z + pw));
                            float const nDotHSq = nDotH * nDotH;
                            float const tanSq = 1.0f / nDotHSq - 1.0f;
                            float const aPlus = ndfA + tanSq;
                            float const D = ndfA / (pi * (nDotHSq * nDotHSq) * (aPlus * aPlus));
                            float const gI = 1.0f + sqrtf(1.0f + geoA / (nDotI * nDotI) - geoA);
                            float const gO = 1.0f + sqrtf(1.0f + geoA / (nDotO * nDotO) - geoA);
                            float const G = 4.0f / (gI * gO);
                            float const pdfS = D * nDotH / (4.0f * hDotO);
                            float const pdfD = nDotI / pi;
                            float const pdf = pS * pdfS + (1.0f - pS) * pdfD;
                            float const scale = nDotI / pdf;
                            float const specular = nDotO > 0.0f ? D * G / (4.0f * nDotI * nDotO) : 0.0f;
                            w = mk((__fmaf_rn(-F.x, adj.x, adj.x) + F.x * specular) * scale,
                                   (__fmaf_rn(-F.y, adj.y, adj.y) + F.y * specular) * scale,
                                   (__fmaf_rn(-F.z, adj.z, adj.z) + F.z * specular) * scale);
                            sampled = pdf > 0.0f;
                        }
                    }
                    if (sampled) {
                        T = mk(w.x * T.x, w.y * T.y, w.z * T.z);                 // src/render.hpp:200-202
                        alive = true;
                        // Russian roulette (no counterpart in the reference, whose paths end only at the bounce
                        // limit, src/render.hpp:117): from the third bounce on a path survives with probability
                        // p = min(1, largest throughput component) and its throughput is divided by p, which
                        // leaves the expectation of every later contribution unchanged.
                        if (R.roulette != 0u && bounce >= 2u) {
                            float big = T.x;
                            if (T.y > big) big = T.y;
                            if (T.z > big) big = T.z;
                            float const survive = big < 1.0f ? big : 1.0f;
                            if (unitFloat(r2) >= survive) alive = false;
                            else T = mk(T.x / survive, T.y / survive, T.z / survive);
                        }
                        if (alive) {
                            V3 const point = mk(o.x + H.x * d.x, o.y + H.x * d.y, o.z + H.x * d.z);
                            P.A[slot] = make_float4(point.x, point.y, point.z, incident.x);
                            P.B[slot] = make_float4(incident.y, incident.z, T.x, T.y);
                            P.Tz[slot] = T.z;
                        }
                    }
                }
            }
        }
        // Stream compaction: ballot, prefix count below the lane.  Survivors are staged in LDS per wave
        // and appended to the next queue in runs of up to 256 with one atomic per run: atomics on the
        // single queue counter serialise (a few ns each), one per 64 paths was a quarter of this kernel.
        unsigned long long const mask = __ballot(alive);
        if (alive) {
            uint32_t const rank = static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
            stage[staged + rank] = slot;
        }
        staged += static_cast<uint32_t>(__popcll(mask));
        if (staged > RT_STAGE - 64u) {
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(&P.counts[bounce + 1], staged);
            start = __builtin_amdgcn_readfirstlane(start);
            // lanes read entries other lanes of the wave wrote above: order the LDS writes before the reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t k = lane; k < staged; k += 64) qout[start + k] = stage[k];
            __builtin_amdgcn_wave_barrier();          // the next step's writes must not overtake these reads
            staged = 0;
        }
    }
    if (staged != 0u) {
        uint32_t start = 0;
        if (lane == 0) start = atomicAdd(&P.counts[bounce + 1], staged);
        start = __builtin_amdgcn_readfirstlane(start);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t k = lane; k < staged; k += 64) qout[start + k] = stage[k];
    }
    unsigned long long const s = waveSum(shaded);
    if (lane == 0 && s) atomicAdd(&gc->shadedHits, s);
}

// Adds the batch's radiance to the per-pixel accumulator in sample order (the order of the
// reference's `colour += rayTrace(...)` loop, src/render.hpp:224-230), and on the last batch divides
// by the sample count (src/render.hpp:231) and writes the image.
__global__ __launch_bounds__(RT_BLOCK) void k_accumulate(PathState P, float* __restrict__ accum, SlotMap M,
                                                         uint32_t rectW, uint32_t rectH,
                                                         uint32_t bounceCount, int last,
                                                         float sppF, float* __restrict__ out,
                                                         Counters* __restrict__ gc) {
    uint32_t const p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < M.pixels) {
        float r = accum[3 * static_cast<size_t>(p)], g = accum[3 * static_cast<size_t>(p) + 1],
              b = accum[3 * static_cast<size_t>(p) + 2];
        // this pixel's slots: first one, then a fixed stride (its group's size) from sample to sample
        uint32_t const first = M.slot(p, 0u);
        uint32_t const stride = M.samples > 1u ? M.slot(p, 1u) - first : 0u;
        // four fetches in flight; the additions stay in sample order
        uint32_t s = 0;
        for (; s + 4u <= M.samples; s += 4u) {
            float4 const C0 = P.L[first + s * stride], C1 = P.L[first + (s + 1u) * stride],
                         C2 = P.L[first + (s + 2u) * stride], C3 = P.L[first + (s + 3u) * stride];
            r = r + C0.x; g = g + C0.y; b = b + C0.z;
            r = r + C1.x; g = g + C1.y; b = b + C1.z;
            r = r + C2.x; g = g + C2.y; b = b + C2.z;
            r = r + C3.x; g = g + C3.y; b = b + C3.z;
        }
        for (; s < M.samples; ++s) {
            float4 const C = P.L[first + s * stride];
            r = r + C.x; g = g + C.y; b = b + C.z;
        }
        if (last) {
            uint32_t x, y;
            slotPixel(p, rectW, rectH, x, y);
            size_t const o = 3 * (static_cast<size_t>(y) * rectW + x);       // row-major image of the rectangle
            out[o] = r / sppF;
            out[o + 1] = g / sppF;
            out[o + 2] = b / sppF;
        }
        else {
            accum[3 * static_cast<size_t>(p)] = r;
            accum[3 * static_cast<size_t>(p) + 1] = g;
            accum[3 * static_cast<size_t>(p) + 2] = b;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long seg = 0;
        for (uint32_t k = 0; k < bounceCount; ++k) seg += P.counts[k];
        gc->segments += seg;
        gc->paths += P.counts[0];
    }
}

// How many blocks of the persistent traversal kernels really are resident at once?  The static share of the line stream is
// dealt out over the grid on the assumption that all of it runs concurrently, and the occupancy query that sizes the grid
// does not see everything (see RT_PERSISTENT_SGPRS).  This kernel has the traversal kernels' footprint -- 256 threads, the
// same LDS stack, 64 vector registers, scalar registers in the same allocation bracket -- and does nothing but count:
// every block announces itself, then waits until the whole grid has been seen alive together or a time limit passes,
// so the largest number alive at once is the residency.  c[0]: blocks alive, c[1]: the peak, c[2]: set once the whole grid
// was alive.  Every block leaves when the flag is up or after `timeoutTicks` of the 100 MHz wall clock.
__global__ __launch_bounds__(RT_BLOCK) RT_PERSISTENT_SGPRS RT_EIGHT_WAVES void k_residency_probe(uint32_t* __restrict__ c, uint32_t target,
                                                                                              unsigned long long timeoutTicks, int touchExtraLds) {
    __shared__ uint32_t lds[RT_LDS_STACK_DEPTH * RT_BLOCK];
    extern __shared__ uint32_t extraLds[];            // measurements: RT_AMD_PROBE_LDS bytes more per block lower the residency
    lds[threadIdx.x] = target;
    asm volatile("" ::: "v63", "s70");                 // the register footprint of the traversal kernels
    if (threadIdx.x == 0) {
        if (touchExtraLds) extraLds[0] = target;
        uint32_t alive = atomicAdd(&c[0], 1u) + 1u;
        atomicMax(&c[1], alive);
        unsigned long long const t0 = wall_clock64();
        for (;;) {
            if (alive >= target) { atomicExch(&c[2], 1u); break; }
            if (atomicAdd(&c[2], 0u) != 0u) break;
            if (wall_clock64() - t0 > timeoutTicks) break;
            __builtin_amdgcn_s_sleep(32);
            alive = atomicAdd(&c[0], 0u);
            atomicMax(&c[1], alive);
        }
        atomicSub(&c[0], 1u);
    }
    __syncthreads();
    if (lds[threadIdx.x] != target) c[3] = 1u;          // keeps the LDS array alive
}

// Device-side evaluation of the scalar building blocks, so tests can pin their bits.
__global__ void k_selftest(int op, const float* __restrict__ a, const float* __restrict__ b, uint32_t n,
                           float* __restrict__ out) {
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op) {
    case 0: out[i] = a[i] / b[i]; break;
    case 1: out[i] = sqrtf(a[i]); break;
    case 2: { float s, c; sincos2pi(a[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; break; }
    case 3: {
        uint32_t r0, r1, r2, r3;
        philox(__float_as_uint(a[i]), __float_as_uint(b[i]), 7u, 0x5EEDu, 0x1234u, r0, r1, r2, r3);
        out[2 * i] = __uint_as_float(r0); out[2 * i + 1] = __uint_as_float(r1);
        break;
    }
    case 4: out[i] = __fmaf_rn(a[i], b[i], a[i]); break;
    case 5: out[i] = a[i] * b[i] + a[i]; break;          // must NOT be fused
    default: out[i] = 0.0f;
    }
}

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
template <class Body>
int guarded(const char* what, Body&& body) {
    try { return body(); }
    catch (std::bad_alloc const&) { return fail(RT_ERR_NOMEM, std::string(what) + ": out of host memory"); }
    catch (std::exception const& e) { return fail(RT_ERR_INVALID, std::string(what) + ": " + e.what()); }
    catch (...) { return fail(RT_ERR_INVALID, std::string(what) + ": unexpected exception"); }
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

int rt_intersect_device(rt_scene* s, const float* d_origins, const float* d_directions, uint64_t n, float t_min,
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
        rc = rt_intersect_device(s, dO, dD, n, t_min, front_only, dH, nullptr);
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

int rt_render(rt_scene* s, const rt_render_params* p, float* out_rgb, rt_stats* stats) {
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
}

}  // extern "C"
