// Nearest intersection through the alternative accelerator, a 4-wide bounding volume hierarchy: traceStreamBvh.
// A fragment of rt_device.hip: included there, inside its anonymous namespace, after the HIP runtime and the
// library's own headers; it is not a translation unit of its own (the kernels are templates that the host code
// of rt_device.hip instantiates and launches).  Must be compiled with -ffp-contract=off, see rt_device.hip.
#ifndef RT_DEV_TRACE_BVH_H
#define RT_DEV_TRACE_BVH_H

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

#endif  // RT_DEV_TRACE_BVH_H
