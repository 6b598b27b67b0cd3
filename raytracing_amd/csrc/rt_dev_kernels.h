// Sampler, path state and the kernels: k_intersect, k_raygen, k_trace, k_shade, k_accumulate, the residency probe and the arithmetic self-test.
// A fragment of rt_device.hip: included there, inside its anonymous namespace, after the HIP runtime and the
// library's own headers; it is not a translation unit of its own (the kernels are templates that the host code
// of rt_device.hip instantiates and launches).  Must be compiled with -ffp-contract=off, see rt_device.hip.
#ifndef RT_DEV_KERNELS_H
#define RT_DEV_KERNELS_H

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

#endif  // RT_DEV_KERNELS_H
