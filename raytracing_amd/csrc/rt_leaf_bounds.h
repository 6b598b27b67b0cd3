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
// v1 + e13) cut to the leaf's box, a description of its triangles' normals (one cone, or up to three axes, below),
// and three coefficients computed at scene creation,
//     cK = sqrt(3) eps (32 kappaMax + 8)        (dimensionless; eps = 2^-24)
//     cE = 2 eMax cK                            (a length; eMax the leaf's longest e12 or e13)
//     cN = 0.999e-6 / nMax                      (nMax the leaf's longest normal),
// kappaMax, eMax and nMax taken over the leaf's triangles.  For a line (o, d) the leaf is dropped when
//     a lower bound gLb > 0 on g holds for every triangle of the leaf that passes the facing test,
//     and the line misses B widened on every side by  P = (cK Dinf + cE) / gLb,
// Dinf being the largest coordinate distance from o to a point of B.
//
// Why a dropped leaf holds no accepted hit (rounding model: every operation exact up to a relative eps).
//  * a = fl(o - v1) is the exact AO of an origin moved by at most eps |a|.
//  * w = fl(a x d) = a x d + dw, |dw_i| <= 2 eps |a| |d|; the fused dot products add 3 eps |e| |w|:
//      fl(e13 . w) = e13.(a x d) + h,  |h| <= 6.5 eps |e13| |a| |d|       (same for e12).
//  * n is the stored normal, N = e12 x e13 the exact one: |n - N| <= 3.5 eps |e12| |e13|, fl(n.d) = n.d + 3 eps |n||d|,
//    so the exact denominator N.d differs from the computed one by at most 6.5 eps |e12||e13||d|, a relative
//    6.5 eps kappa / g, which the rule keeps below 2^-10 (it refuses to drop when cK / gLb > 2^-7).
//  * u >= 0 in floating point therefore implies for the exact barycentric coordinate of the exact line
//      uTrue >= -6.5 eps |e13| |a| / (g |n|) (1 + 2^-9),
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
// Three lower bounds on g, the largest is used:
//  * from one cone (smooth surfaces): every normal lies within b of +-A1, the direction makes the angle alpha with
//    +-A1, cos(alpha) = |A1.d| / |d|, so every normal makes at most alpha + b with +-d and  g >= cos(alpha + b) =
//    cos b cos(alpha) - sin b sin(alpha).  The record stores cos b rounded down and sin b rounded up (a part in 2^18 to
//    spare each, which covers the unit axis' own rounding); |d| sin(alpha) = sqrt(d.d - (A1.d)^2) is evaluated with a
//    part in 2^10 and 2^-18 d.d added under the root, which covers the rounding of A1.d.
//  * from up to three axes (boxes, rooms: a few flat faces in a leaf; the form of ALL records of a scene in which enough
//    leaves are of that kind, chosen at scene creation): every normal lies within b2 of one of +-A1, +-A2, +-A3, so
//    g >= min_k cos(alpha_k + b2) >= min_k |Ak.d| / |d| - 2 sin b2  (cos(x + y) >= cos x - (1 - cos y) - sin y and
//    1 - cos y <= sin y).  A leaf with fewer axes repeats the last one; a leaf described by one cone stores its axis
//    three times and sin b (the bound is valid for it too, and weaker than the angle sum).
//    A leaf whose normals fit neither description has sin b = 1 (and cos b = 0).
//  * from the facing test itself: an accepted triangle has |fl(n.d)| >= 1e-6 (src/geometry.hpp:126), so
//    g >= 1e-6 / (|n| |d|) - 3 eps: small triangles (short normals) cannot be hit at a grazing angle at all.
//    |d| gLb = cN - 2^-22 |d|.
// The rule applies when gLb >= 2^-10 and cK / gLb <= 2^-7.  (Normals enter only through |A.d|: a normal and its
// opposite are the same axis, which is what the triangle test sees too, up to the sign it selects by.)
//
// The slab test is evaluated on crossing times scaled by G = |d| gLb, which avoids the division: a plane of the widened
// box is crossed at ((plane -+ P) - o) r, and G times that is fma(plane - o, G, -+M) r with M = P G = (cK Dinf + cE) |d|.
// Each scaled time carries a relative error below 4 eps; the interval ends are widened by 2^-20.  The caller passes
// rayOk = every |direction component| >= 2^-16 and finite and the origin finite (filterMargin's condition), so r is
// finite and no 0 * inf arises.  An overflow (coordinates beyond 2^100 or so) ends in inf - inf = NaN in the final
// comparison, and a NaN anywhere makes it false: no drop.
#ifndef RT_LEAF_BOUNDS_H
#define RT_LEAF_BOUNDS_H

#include <math.h>
#include <stdint.h>
#include <string.h>

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

// Leaf normal cone (a leaf's aux word: count in bits 0-7, q in 8-12, A.z 13-18, A.y 19-24, A.x 25-30; bit 31: no cone): true
// when every triangle of the leaf fails the facing test for direction d (Surfaces::FRONT_ONLY, src/geometry.hpp:129-131),
// i.e. when A . d >= q |d|; see the comment above coneAllBackFacing in rt_device.hip for the argument.
RT_HD bool rtConeAllBackFacing(uint32_t aux, float dx, float dy, float dz) {
    float const ax = static_cast<float>(static_cast<int32_t>(aux << 1) >> 26);
    float const ay = static_cast<float>(static_cast<int32_t>(aux << 7) >> 26);
    float const az = static_cast<float>(static_cast<int32_t>(aux << 13) >> 26);
    float const q = static_cast<float>((aux >> 8) & 0x1Fu);
    float const ad = fmaf(ax, dx, fmaf(ay, dy, az * dz));
    float const dd = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    // a direction so long that the squares overflow, or so short that they underflow, decides nothing
    float const lhs = ad * ad, rhs = (q * q) * dd;
    return static_cast<int32_t>(aux) >= 0 && ad > 0.0f && lhs >= rhs && lhs < INFINITY && rhs >= 0x1p-100f;
}

// |d| gLb from a record's description of the normals.  One-cone form (RT_LEAF_AXES_ONE records): sinB, cosB, axis A1.
RT_HD float rtLeafGCone(float sinB, float cosB, float a1x, float a1y, float a1z, float dx, float dy, float dz, float dd) {
    float const ad1 = fabsf(fmaf(a1x, dx, fmaf(a1y, dy, a1z * dz)));         // |d| cos(alpha_1)
    float const cr = sqrtf(fmaf(fmaxf(fmaf(-ad1, ad1, dd), 0.0f), 1.0f + 0x1p-10f, 0x1p-18f * dd));      // >= |d| sin(alpha_1)
    return fmaf(cosB, ad1, -(sinB * cr));
}
// Axes form (RT_LEAF_AXES_PACKED records, same length): sin b2 as an unsigned 16-bit fraction (unit = 65536, rounded up) and
// three axes as signed 16-bit components (unit length = 32767), ten halves in the five words that hold {sinB, cosB, A1}
// in the cone form:
//   word 0: sin b2 | A1.x << 16   word 1: A1.y | A1.z << 16   word 2: A2.x | A2.y << 16   word 3: A2.z | A3.x << 16
//   word 4: A3.y | A3.z << 16
// The quantisation moves an axis by less than 2^-15 rad, which the stored sin b2 allows for; dividing by 32768 instead of
// the quantised axis' own length (at most 32767 + 1) only lowers the bound.
#define RT_LEAF_AXES_ONE 1
#define RT_LEAF_AXES_PACKED 3
RT_HD float rtLeafSnorm16(uint32_t word, int high) {
    return static_cast<float>(static_cast<int32_t>(high ? word : word << 16) >> 16);
}
RT_HD float rtLeafGPacked(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, float dx, float dy, float dz, float dlen) {
    float const sin2 = static_cast<float>(w0 & 0xFFFFu) * 0x1p-15f;          // 2 sin b2
    float const ad1 = fabsf(fmaf(rtLeafSnorm16(w0, 1), dx, fmaf(rtLeafSnorm16(w1, 0), dy, rtLeafSnorm16(w1, 1) * dz)));
    float const ad2 = fabsf(fmaf(rtLeafSnorm16(w2, 0), dx, fmaf(rtLeafSnorm16(w2, 1), dy, rtLeafSnorm16(w3, 0) * dz)));
    float const ad3 = fabsf(fmaf(rtLeafSnorm16(w3, 1), dx, fmaf(rtLeafSnorm16(w4, 0), dy, rtLeafSnorm16(w4, 1) * dz)));
    return fmaf(-sin2, dlen, fminf(fminf(ad1, ad2), ad3) * 0x1p-15f);
}

// The rule, given gN = |d| gLb from the normals' description.  Record fields: box lo/hi; cK, cE, cN as above.  Returns
// true when the leaf may be left unscanned.
RT_HD bool rtLeafBoundsMiss(float lox, float loy, float loz, float hix, float hiy, float hiz, float cK, float cE, float cN,
                            float gN, float dlen, float ox, float oy, float oz, float rx, float ry, float rz, bool rayOk) {
    float const G = fmaxf(gN, fmaf(-0x1p-22f, dlen, cN));                   // |d| gLb
    float const lx = lox - ox, ly = loy - oy, lz = loz - oz, hx = hix - ox, hy = hiy - oy, hz = hiz - oz;
    float const dinf = fmaxf(fmaxf(fmaxf(fabsf(lx), fabsf(hx)), fmaxf(fabsf(ly), fabsf(hy))), fmaxf(fabsf(lz), fabsf(hz)));
    float const M = fmaf(cK, dinf, cE) * dlen;                              // P G
    bool const valid = rayOk && G >= fmaxf(0x1p-10f, 0x1p7f * cK) * dlen;
    float const tax = fmaf(lx, G, -M) * rx, tbx = fmaf(hx, G, M) * rx;
    float const tay = fmaf(ly, G, -M) * ry, tby = fmaf(hy, G, M) * ry;
    float const taz = fmaf(lz, G, -M) * rz, tbz = fmaf(hz, G, M) * rz;
    float const tn = fmaxf(fmaxf(fminf(tax, tbx), fminf(tay, tby)), fminf(taz, tbz));
    float const tf = fminf(fminf(fmaxf(tax, tbx), fmaxf(tay, tby)), fmaxf(taz, tbz));
    return valid && (tn - fabsf(tn) * 0x1p-20f > tf + fabsf(tf) * 0x1p-20f);
}

// A bounds record as stored, 16 floats (words 6, 7 and, in the axes form, 11-15 hold integers):
//   {lo.xyz, hi.x} {hi.yz, cone word, leaf index} {cK, cE, cN, sinB | packed} {cosB, A1.xyz | packed}
RT_HD bool rtLeafRecordMiss(const float* rec, int axes, float ox, float oy, float oz, float dx, float dy, float dz, float rx,
                            float ry, float rz, bool rayOk) {
    float const dd = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    float const dlen = sqrtf(dd);
    float gN;
    if (axes == RT_LEAF_AXES_ONE) gN = rtLeafGCone(rec[11], rec[12], rec[13], rec[14], rec[15], dx, dy, dz, dd);
    else {
        uint32_t w[5];
        memcpy(w, rec + 11, 20);
        gN = rtLeafGPacked(w[0], w[1], w[2], w[3], w[4], dx, dy, dz, dlen);
    }
    return rtLeafBoundsMiss(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[8], rec[9], rec[10], gN, dlen, ox, oy, oz, rx, ry, rz,
                            rayOk);
}

#endif
