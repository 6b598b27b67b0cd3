#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a checker: it compares the library with the oracle; the library never uses it).
CPU check of the rule by which the traversal leaves a BSP leaf unscanned (rtLeafBoundsMiss, rt_leaf_bounds.h).

For adversarial scenes and lines (tests/grazing.py: lines that graze the planes of rotated triangles of every size, in
scenes from millimetres to kilometres) every (line, leaf) pair is evaluated twice: the oracle's restatement of the
reference's leaf scan (src/bsp.hpp:58-104) says whether the leaf yields a hit, and the library's own host replay of
the rule (rt_debug_leaf_drop, the kernels' arithmetic) says whether the kernels would have dropped the leaf.  A
dropped leaf with a hit is a wrong decision (must be 0).  Also printed: the share of hit-free pairs the rule drops
(what the rule is worth), and how many wrong decisions the fixed padding the library used before (triangle corners
padded by 2^-18 of the largest box coordinate) makes on the same pairs, as evidence that the lines bite.

Usage: python3 tools/leaf_bounds_check.py [seed] [scenes] [lines per triangle]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding
import grazing
import helpers
from raytracing_amd.render import FlatTree, RayTraceData

f32 = np.float32


def fixed_padding_drop(flat, slots, tris_by_slot, o, d):
    """The rule the library used before: slab test against the triangles' corner box cut to the leaf box and padded by
    2^-18 of the largest box coordinate, interval ends widened by 2^-16 (float32 throughout)."""
    nodes = flat.nodes[slots]
    pad = f32(flat.box_abs_max * 2.0 ** -18)
    lo = np.maximum(tris_by_slot[0], nodes["lo"]) - pad
    hi = np.minimum(tris_by_slot[1], nodes["hi"]) + pad
    with np.errstate(all="ignore"):
        r = (f32(1) / d).astype(f32)
        a = ((lo - o).astype(f32) * r).astype(f32)
        b = ((hi - o).astype(f32) * r).astype(f32)
        tn = np.fmax(np.fmax(np.fmin(a[:, 0], b[:, 0]), np.fmin(a[:, 1], b[:, 1])), np.fmin(a[:, 2], b[:, 2]))
        tf = np.fmin(np.fmin(np.fmax(a[:, 0], b[:, 0]), np.fmax(a[:, 1], b[:, 1])), np.fmax(a[:, 2], b[:, 2]))
        return (tn - np.abs(tn) * f32(2.0 ** -16)).astype(f32) > (tf + np.abs(tf) * f32(2.0 ** -16)).astype(f32)


SHRINK = (1.0, 1 / 16, 1 / 256, 1 / 4096, 1 / 65536)      # the rule's coefficients scaled down: where do wrong drops begin?


def check_scene(O, rng, scale, lines_per_triangle, targets, max_leaf_tris, verbose=False):
    scene, tris = grazing.make_scene(rng, scale, targets=targets)
    scene.preprocess(max_leaf_tris=max_leaf_tris)
    rtd = RayTraceData(scene)
    flat = FlatTree(rtd)
    osc = O.scene_create_with_tree(helpers.base_from_scene(scene), 64, 64, scene.bsp_tree)
    slots, leaves, boxes = grazing.leaf_table(flat, scene.bsp_tree)
    # corner boxes of each leaf's triangles (for the old rule)
    pre = np.asarray(rtd.pretris, f32)
    first = np.asarray(scene.bsp_tree.leaf_first_ref, np.int64)
    refs = scene.bsp_tree.leaf_refs
    begin = np.asarray(rtd.pretri_begin, np.int64)
    clo, chi = np.zeros((len(slots), 3), f32), np.zeros((len(slots), 3), f32)
    for k, leaf in enumerate(leaves):
        r = refs[first[leaf]:first[leaf + 1]]
        p = pre[begin[r["mesh"]] + r["tri"]]
        corners = np.stack([p[:, 3:6], p[:, 3:6] + p[:, 9:12], p[:, 3:6] + p[:, 6:9]])
        clo[k], chi[k] = corners.min((0, 1)), corners.max((0, 1))

    od = [grazing.grazing_lines(rng, t, lines_per_triangle, scale) for t in tris]
    o = np.concatenate([x[0] for x in od]); d = np.concatenate([x[1] for x in od])
    res = dict(pairs=0, hits=0, wrong=0, wrong_fixed=0, dropped=0, dropped_fixed=0, free=0, cones=int(flat.described()[slots].sum()), leaves=len(slots))
    for f in SHRINK[1:]:
        res[f"wrong_at_{f}"] = 0
    shrunk = []
    for f in SHRINK[1:]:
        g = FlatTree(rtd)
        g.bounds["cK"] *= f32(f); g.bounds["cE"] *= f32(f)
        shrunk.append(g)
    for k in range(len(slots)):
        sl = np.full(len(o), slots[k], np.uint32)
        drop = flat.leaf_drop(sl, o, d)
        weaker = [g.leaf_drop(sl, o, d) for g in shrunk]
        old = fixed_padding_drop(flat, sl, (clo[k], chi[k]), o, d)
        for front in (True, False):
            hit, _ = osc.leaf_hits(np.full(len(o), leaves[k], np.uint32), np.tile(boxes[k], (len(o), 1)), o, d, t_min=-1e30,
                                   front_only=front)
            res["pairs"] += len(o); res["hits"] += int(hit.sum())
            res["wrong"] += int((drop & hit).sum()); res["wrong_fixed"] += int((old & hit).sum())
            for f, w in zip(SHRINK[1:], weaker):
                res[f"wrong_at_{f}"] += int((w & hit).sum())
            res["free"] += int((~hit).sum()); res["dropped"] += int((drop & ~hit).sum()); res["dropped_fixed"] += int((old & ~hit).sum())
            if verbose and (drop & hit).any():
                i = np.nonzero(drop & hit)[0][0]
                print("WRONG DROP: slot", slots[k], "o", o[i].tolist(), "d", d[i].tolist(), "front" if front else "all")
    osc.close()
    return res


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    O = binding.oracle()
    rng = np.random.default_rng(seed)
    total = {}
    for s in range(scenes):
        scale = 10.0 ** rng.uniform(-2, 3)
        single = s % 2 == 0
        r = check_scene(O, rng, scale, per * (8 if single else 1), 1 if single else 8, 32 if single else 1, verbose=True)
        print(f"scene {s}: scale {scale:9.3g}  leaves {r['leaves']} ({r['cones']} with a cone)  pairs {r['pairs']}  with a hit {r['hits']}  wrong drops {r['wrong']}  "
              f"(fixed padding: {r['wrong_fixed']})  hit-free pairs dropped {100.0 * r['dropped'] / max(r['free'], 1):.1f} % "
              f"(fixed padding {100.0 * r['dropped_fixed'] / max(r['free'], 1):.1f} %)", flush=True)
        for k, v in r.items():
            total[k] = total.get(k, 0) + v
    print(f"total: pairs {total['pairs']}  with a hit {total['hits']}  wrong drops {total['wrong']}  (fixed padding: {total['wrong_fixed']})")
    print("wrong drops with the rule's coefficients scaled by " + ", ".join(f"{f:g}: {total[f'wrong_at_{f}']}" for f in SHRINK[1:]))
    return 1 if total["wrong"] else 0


if __name__ == "__main__":
    sys.exit(main())
