// Leaf triangle bounds: when may a BSP leaf be left unscanned because the line passes clear of its triangles?
//
// One definition for the traversal kernels (device) and for the host replay behind rt_debug_leaf_drop, so that
// the rule can be checked against the reference's arithmetic without a GPU.  Every operation is IEEE binary32,
// written so that host and device produce the same bits (explicit fused multiply-adds, correctly rounded square
// root, no reciprocal approximations); both sides must be compiled with -ffp-contract=off.
//
// The reference scans every leaf its line enters and accepts a triangle on the floating-point values
//     negDet = n.d,  u = (e13 . (AO x d)) / -negDet,  v = -(e12 . (AO x d)) / -negDet,  AO = o - v1
// (src/geometry.hpp:117-132) when |negDet| >= 1e-6, u >= 0, v >= 0, u + v <= 1, and then when the point
// o + t d lies in the leaf's box (src/bsp.hpp:79-87).  The cross product AO x d carries an absolute rounding
// error of about eps |AO| |d| per component whatever the triangle's orientation, and the division by n.d turns it
// into an error of the hit's position in the triangle's plane of about
//     eps |AO| kappa / g,      g = |n.d| / (|n| |d|)  (sine of the angle between line and plane),
//                              kappa = (longest of |e12|, |e13|)^2 / |n|  (1 for a right isosceles corner),
// which has no bound in terms of the triangle's size: the facing test only demands |n.d| >= 1e-6 absolutely, so a
// line that grazes the plane can be accepted although it passes the triangle at a distance of many edge lengths.
// A fixed padding of the triangles' bounds therefore cannot be right; the padding has to grow as the line
// approaches the plane.
//
// Rule.  A leaf's bounds record holds B = (box around its triangles as the test sees them: v1, v1 + e12,
// v1 + e13) cut to the leaf's box, the cone around its triangles' normals (axis A, threshold q, see
// coneAllBackFacing in rt_device.hip), and two coefficients computed at scene creation,
//     cK = sqrt(3) eps (32 kappaMax + 8)        (dimensionless; eps = 2^-24)
//     cE = 2 eMax cK                            (a length; eMax the leaf's longest e12 or e13),
// kappaMax and eMax taken over the leaf's triangles.  For a line (o, d) the leaf is dropped when
//     the cone proves  g >= gLb > 0  for every triangle of the leaf,
//     and the line misses B widened on every side by  P = (cK Dinf + cE) / gLb,
// Dinf being the largest coordinate distance from o to a point of B.
//
// Why a dropped leaf holds no accepted hit (rounding model: every operation exact up to a relative eps).
//  * a = fl(o - v1) is the exact AO of an origin moved by at most eps |a|.
//  * w = fl(a x d) = a x d + dw, |dw_i| <= 2 eps |a| |d|; the fused dot products add 3 eps |e| |w|:
//      fl(e13 . w) = e13.(a x d) + h,  |h| <= 6.5 eps |e13| |a| |d|       (same for e12).
//  * n is the stored normal, N = e12 x e13 the exact one: |n - N| <= 3.5 eps |e12| |e13|, fl(n.d) = n.d + 3 eps |n||d|,
//    so the exact denominator N.d differs from the computed one by at most 6.5 eps |e12||e13||d|, a relative
//    6.5 eps kappa / g, which the rule keeps below 2^-7 (it refuses to drop otherwise).
//  * u >= 0 in floating point therefore implies for the exact barycentric coordinate of the exact line
//      uTrue >= -6.5 eps |e13| |a| / (g |n|) (1 + 2^-6),
//    i.e. the exact plane point lies within 6.5 eps kappa |a| / g of the edge; the same for v; and
//    fl(u + v) <= 1 implies uTrue + vTrue - 1 <= (13 eps kappa |a| + 10 eps kappa eMax) / (g eMax), i.e. within
//    13 eps kappa (|a| + eMax) / g of the third edge.  Together: the exact line passes within
//      R = 26 eps kappa (|a| + eMax) / g
//    of the triangle, hence of the box around it.
//  * The accepted point fl(o + fl(t d)) lies in the leaf's box; t carries the same two errors, which moves the
//    point along the line by at most 13 eps kappa (|a| + eMax) / g + 2 eps (|a| + eMax), and the two roundings of
//    the point add eps (|a| + eMax) + eps |coordinate|.  So the exact plane point lies within R of the leaf's box
//    as well, up to the terms without 1/g.
//  * Terms without 1/g: origin shift eps |a|, point roundings 3 eps (|a| + eMax), the subtraction fl(plane - o) of
//    the slab test below eps Dinf: at most 8 eps (|a| + eMax) with room to spare; eps |coordinate| is covered by
//    padding the stored bounds by 2^-22 of the largest box coordinate.
//  * |a| + eMax <= sqrt(3) (Dinf + 2 eMax): every triangle of the leaf meets B, so v1 lies within eMax of it.
//  Hence the exact plane point of an accepted hit lies inside B widened by sqrt(3) eps (Dinf + 2 eMax)(26 kappa + 8)/g
//  <= P (the coefficients use 32 for 26: second-order terms, the rounding of P and of gLb themselves), and it lies
//  on the exact line: a line that misses the widened B has no accepted hit in the leaf.
//
// The cone gives gLb: with c = |A.d| / (|A||d|) and s = q / |A| >= sin(spread + 0.002), every normal makes an angle of
// at most acos(c) + asin(s) with +-d, whose cosine is at least c (1 - s^2) - s >= c - s - s^2.  |A| lies in [30, 32]
// (a point within sqrt(3)/2 of the sphere of radius 31), so  32 |d| gLb = G = |A.d| - (32/30) q (1 + q/30) |d|  is a
// valid (slightly smaller) bound; the rule needs G >= 2^-5 |d| (gLb >= 2^-10), which also keeps the rounding of G itself
// (a few eps |A.d|) below one part in a thousand.  A leaf without a cone (A = 0) is never dropped.
//
// The slab test is evaluated on crossing times scaled by G, which avoids the division: a plane of the widened box
// is crossed at ((plane -+ P) - o) r, and G times that is fma(plane - o, G, -+M) r with M = P G = (cK Dinf + cE) 32 |d|.
// Each scaled time carries a relative error below 4 eps; the interval ends are widened by 2^-20.  The caller passes
// rayOk = every |direction component| >= 2^-16 and finite and the origin finite (filterMargin's condition), so r is
// finite and no 0 * inf arises.  Any NaN or overflow makes a comparison false, i.e. no drop.
#ifndef RT_LEAF_BOUNDS_H
#define RT_LEAF_BOUNDS_H

#include <math.h>
#include <stdint.h>

#if defined(__HIP__)
#define RT_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define RT_HD inline
#endif

// The margin of the conservative slab filter in front of the six-face box test (see boxFilter in rt_device.hip), in
// crossing-time units: m = 2^-20 (largest |box coordinate| + largest |origin coordinate|) times the largest |1/d|,
// or +inf when a direction component is below 2^-16 in magnitude or not finite (NaN included).  A finite margin is
// also the condition under which the leaf bounds rule below may be applied to the line.
RT_HD float rtFilterMargin(float ox, float oy, float oz, float dx, float dy, float dz, float rx, float ry, float rz,
                           float boxAbsMax) {
    float const m = 0x1p-20f * (boxAbsMax + fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)));
    float const dMin = fminf(fminf(fabsf(dx), fabsf(dy)), fabsf(dz));
    float const dMax = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    float const rMax = fmaxf(fmaxf(fabsf(rx), fabsf(ry)), fabsf(rz));
    return (dMin >= 0x1p-16f && dMax < INFINITY) ? m * rMax : INFINITY;      // false for NaN components too
}

// cone word layout: count (bits 0-7), q (8-12), A.z (13-18), A.y (19-24), A.x (25-30); bit 31: no cone.
// Returns true when the leaf may be left unscanned.
RT_HD bool rtLeafBoundsMiss(float lox, float loy, float loz, float hix, float hiy, float hiz, uint32_t coneWord, float cK,
                            float cE, float ox, float oy, float oz, float dx, float dy, float dz, float rx, float ry,
                            float rz, bool rayOk) {
    float const ax = static_cast<float>(static_cast<int32_t>(coneWord << 1) >> 26);
    float const ay = static_cast<float>(static_cast<int32_t>(coneWord << 7) >> 26);
    float const az = static_cast<float>(static_cast<int32_t>(coneWord << 13) >> 26);
    float const q = static_cast<float>((coneWord >> 8) & 0x1Fu);
    float const ad = fmaf(ax, dx, fmaf(ay, dy, az * dz));
    float const dlen = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
    float const qf = q * fmaf(q, 32.0f / 900.0f, 32.0f / 30.0f);          // (32/30) q (1 + q/30)
    float const G = fabsf(ad) - qf * dlen;                                 // 32 |d| gLb
    float const lx = lox - ox, ly = loy - oy, lz = loz - oz, hx = hix - ox, hy = hiy - oy, hz = hiz - oz;
    float const dinf = fmaxf(fmaxf(fmaxf(fabsf(lx), fabsf(hx)), fmaxf(fabsf(ly), fabsf(hy))), fmaxf(fabsf(lz), fabsf(hz)));
    float const scale = 32.0f * dlen;
    float const M = fmaf(cK, dinf, cE) * scale;                            // P G
    bool const valid = rayOk && static_cast<int32_t>(coneWord) >= 0 && G >= 0x1p-5f * dlen && cK * scale <= 0x1p-7f * G
                    && dinf < 0x1p100f && M < INFINITY;
    float const tax = fmaf(lx, G, -M) * rx, tbx = fmaf(hx, G, M) * rx;
    float const tay = fmaf(ly, G, -M) * ry, tby = fmaf(hy, G, M) * ry;
    float const taz = fmaf(lz, G, -M) * rz, tbz = fmaf(hz, G, M) * rz;
    float const tn = fmaxf(fmaxf(fminf(tax, tbx), fminf(tay, tby)), fminf(taz, tbz));
    float const tf = fminf(fminf(fmaxf(tax, tbx), fmaxf(tay, tby)), fmaxf(taz, tbz));
    return valid && (tn - fabsf(tn) * 0x1p-20f > tf + fabsf(tf) * 0x1p-20f);
}

#endif
