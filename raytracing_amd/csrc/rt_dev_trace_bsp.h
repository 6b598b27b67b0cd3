// Nearest intersection through the reference's BSP tree (src/bsp.hpp:50-145): traceStream.
// A fragment of rt_device.hip: included there, inside its anonymous namespace, after the HIP runtime and the
// library's own headers; it is not a translation unit of its own (the kernels are templates that the host code
// of rt_device.hip instantiates and launches).  Must be compiled with -ffp-contract=off, see rt_device.hip.
#ifndef RT_DEV_TRACE_BSP_H
#define RT_DEV_TRACE_BSP_H

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

#endif  // RT_DEV_TRACE_BSP_H
