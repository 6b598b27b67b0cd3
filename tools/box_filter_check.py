#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a checker: it compares the library with the oracle; the library never uses it).
CPU check of the conservative ray/box filter used in front of the exact six-face test.

For many (line, box) pairs, including lines aimed a few ulps around box corners, edges and faces, the
filter's "surely meets" / "surely misses" decisions are compared with the exact test (the oracle's
restatement of src/geometry.hpp:135-214).  Prints the number of wrong sure decisions (must be 0) and the
share of undecided pairs (the cost of the fallback)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import binding

f32 = np.float32


def filter_decisions(o, d, lo, hi, mbox):
    """Vectorised float32 restatement of the device filter.  Returns (sure_hit, sure_miss)."""
    with np.errstate(all="ignore"):
        r = (f32(1.0) / d).astype(f32)
        m = f32(2.0 ** -20) * (f32(mbox) + np.abs(o).max(axis=1)).astype(f32)
        rmax = np.abs(r).max(axis=1)
        ok = (np.abs(d).min(axis=1) >= f32(2.0 ** -16)) & (np.abs(d).max(axis=1) < f32(np.inf))
        mu = np.where(ok, (m * rmax).astype(f32), f32(np.inf)).astype(f32)
        # crossing times as the kernel forms them: fma(plane, r, nor) with nor = -(o * r) rounded once.  (The fused
        # multiply-add is emulated in float64, where the product of two float32 values is exact.)
        nor = (-(o * r)).astype(f32)
        a = (lo.astype(np.float64) * r + nor).astype(f32)
        b = (hi.astype(np.float64) * r + nor).astype(f32)
        te = np.minimum(a, b); tx = np.maximum(a, b)
        tE = te.max(axis=1); tX = tx.min(axis=1)
        med = np.sort(tx, axis=1)[:, 1]
        theta = (f32(2.0 ** -20) * np.abs(tX) + mu).astype(f32)
        gap = (tX - tE).astype(f32)
        exit_clear = (med - tX).astype(f32) > theta
        sure_hit = (gap > theta) & exit_clear & (tX > theta)
        theta2 = (f32(2.0 ** -20) * (np.abs(tE) + np.abs(tX)).astype(f32) + (mu + mu).astype(f32)).astype(f32)
        sure_miss = (tX < -theta) | ((tE - tX).astype(f32) > theta2)
        # origin-inside rule, applied only to pairs still open
        inside = ((o >= lo) & (o <= hi)).all(axis=1)
        on_plane = ((o == lo) | (o == hi)).any(axis=1)
        sure_hit = sure_hit | (~sure_hit & ~sure_miss & inside & np.isfinite(mu) & (exit_clear | on_plane))
    return sure_hit, sure_miss


def nudge(x, k):
    """x moved by k ulps (k integer array)."""
    i = x.view(np.int32).astype(np.int64)
    i = np.where(i < 0, -(i & 0x7FFFFFFF), i) + k
    i = np.where(i < 0, (-i) | 0x80000000, i)
    return i.astype(np.uint32).view(np.float32)


def main():
    O = binding.oracle()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    total = wrong = undecided = 0
    for trial in range(40):
        n = 500_000
        scale = f32(10.0 ** rng.uniform(-2, 3))
        c = rng.uniform(-1, 1, (n, 3)) * scale
        ext = rng.uniform(0, 1, (n, 3)) ** 3 * scale
        lo = (c - ext).astype(f32); hi = (c + ext).astype(f32)
        mbox = max(np.abs(lo).max(), np.abs(hi).max())
        o = (rng.uniform(-2, 2, (n, 3)) * scale).astype(f32)
        # a quarter of the origins inside the box or exactly on its planes
        w = rng.uniform(0, 1, (n, 3))
        w = np.where(rng.random((n, 3)) < 0.4, np.round(w), w)
        o = np.where((rng.random(n) < 0.25)[:, None], (lo + (hi - lo) * w).astype(f32), o).astype(f32)
        o = np.where((rng.random(n) < 0.1)[:, None] & (rng.random((n, 3)) < 0.5), np.where(rng.random((n, 3)) < 0.5, lo, hi), o).astype(f32)
        kind = rng.integers(0, 5, n)
        # target point: random / on a face / on an edge / at a corner of the box
        u = rng.uniform(0, 1, (n, 3))
        snap = np.zeros((n, 3), bool)
        for k, cnt in ((1, 1), (2, 2), (3, 3)):
            rows = np.nonzero(kind == k)[0]
            for row_axes in [rng.permuted(np.tile(np.arange(3), (len(rows), 1)), axis=1)[:, :cnt]]:
                for j in range(cnt):
                    snap[rows, row_axes[:, j]] = True
        u = np.where(snap, np.round(u), u)
        target = (lo + (hi - lo) * u).astype(f32)
        target = np.where(kind[:, None] == 4, (c + rng.uniform(-3, 3, (n, 3)) * scale), target).astype(f32)
        target = nudge(target, rng.integers(-3, 4, (n, 3)))
        v = (target.astype(np.float64) - o) * rng.choice([-1.0, 1.0], (n, 1), p=[0.15, 0.85])
        d = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)
        axis_aligned = rng.random(n) < 0.02
        d = np.where(axis_aligned[:, None] & (rng.random((n, 3)) < 0.5), f32(0), d).astype(f32)
        d[(d == 0).all(axis=1)] = f32(1)
        d = nudge(d, rng.integers(-2, 3, (n, 3)))
        truth = O.line_intersects_box(o, d, np.concatenate([lo, hi], axis=1)).astype(bool)
        sh, sm = filter_decisions(o, d, lo, hi, mbox)
        bad = (sh & ~truth) | (sm & truth) | (sh & sm)
        total += n; wrong += int(bad.sum()); undecided += int((~sh & ~sm).sum())
        if bad.any():
            i = np.nonzero(bad)[0][0]
            print("WRONG", o[i], d[i], lo[i], hi[i], truth[i], sh[i], sm[i])
    print(f"pairs {total}  wrong sure decisions {wrong}  undecided {undecided} ({100.0 * undecided / total:.2f} %)")
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main())
