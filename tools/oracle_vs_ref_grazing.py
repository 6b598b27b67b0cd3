#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, development container only (needs oracle/_ref/libref.so, i.e. /root/reference): the oracle's
restatement against the compiled reference on fresh adversarial scenes -- lines that graze the planes of rotated triangles
(tests/grazing.py), the set on which the reference's floating-point accept decision is least like the geometric one.
Each side builds its own tree (the reference's constructor, the library's builder through the oracle binding); nearest
intersections are compared bit for bit in both surface modes.  Prints one JSON object with the totals.
Usage: python3 tools/oracle_vs_ref_grazing.py [scenes] [lines per scene] [first seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import grazing
import helpers
from oracle import binding

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per_scene = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 31000
R = binding.reference()
assert R is not None, "build oracle/_ref/libref.so first (oracle/build_ref.sh)"
O = binding.oracle()
tot = dict(scenes=0, lines=0, reference_hits=0, hit_records_differing=0, trees_differing=0)
t0 = time.time()
for k in range(scenes):
    rng = np.random.default_rng(seed0 + k)
    scale = 10.0 ** rng.uniform(-3, 3)
    targets = int(rng.choice([1, 1, 4, 8, 24]))
    edges = [(1e-3, 10.0), (1.0, 10.0), (0.02, 2.0), (3e-3, 0.3)][int(rng.integers(0, 4))]
    scene, tris = grazing.make_scene(rng, scale, targets=targets, edge_range=edges)
    scene.preprocess(max_leaf_tris=32)
    base = helpers.base_from_scene(scene)
    od = [grazing.grazing_lines(rng, t, per_scene // targets, scale) for t in tris]
    o = np.concatenate([x[0] for x in od]); d = np.concatenate([x[1] for x in od])
    rs = R.scene_create(base, 64, 64)
    osc = O.scene_create(base, 64, 64)
    try:
        rt, ot = rs.bsp(), osc.bsp()
        if not all(np.array_equal(rt[key], ot[key]) for key in rt):
            tot["trees_differing"] += 1
        for front in (True, False):
            rf, ri = rs.intersect(o, d, 1e-3, front)
            of, oi = osc.intersect(o, d, 1e-3, front)
            bad = (ri != oi).any(1) | (np.asarray(rf, np.float32).view(np.uint32) != np.asarray(of, np.float32).view(np.uint32)).any(1)
            tot["hit_records_differing"] += int(bad.sum())
            tot["reference_hits"] += int((ri[:, 0] != 0xFFFFFFFF).sum())
        tot["lines"] += len(o)
        tot["scenes"] += 1
    finally:
        rs.close(); osc.close()
    print(f"scene {k}: scale {scale:.3g}, {targets} targets: {tot['hit_records_differing']} differing so far", file=sys.stderr, flush=True)
tot["seconds"] = time.time() - t0
tot["first_seed"] = seed0
print(json.dumps(tot, indent=1))
