#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (a checker: it compares the library with the oracle; the library never uses it).
A larger run of what tests/test_grazing.py does with fixed seeds: many fresh adversarial scenes (rotated triangles of
every size and shape, alone in their leaves or sharing them; scenes from millimetres to kilometres) and lines that graze
the triangles' planes a little outside the triangles, traced on the GPU by rt_intersect and by rt_intersect_check (the
kernels that cross-check every conservative shortcut against the exact evaluation) and compared with the oracle's
restatement of the reference on the same tree, both surface modes, with the leaf bounds records in either form.
Prints one JSON object with the totals; any difference or any counted disagreement is a bug.
Usage (GPU box): python3 tools/grazing_campaign.py [scenes] [lines per scene] [first seed] > profiles/r03_grazing_campaign.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import grazing
from helpers import base_from_scene
from oracle import binding
from raytracing_amd.render import DeviceScene, RayTraceData

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per_scene = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
tot = dict(scenes=0, lines=0, queries=0, reference_hits=0, hit_records_differing=0, counted_disagreements=0, leaf_visits=0,
           box_tests=0)
t0 = time.time()
for k in range(scenes):
    rng = np.random.default_rng(seed0 + k)
    scale = 10.0 ** rng.uniform(-3, 3)
    targets = int(rng.choice([1, 1, 4, 8, 24]))
    cap = int(rng.choice([1, 2, 32]))
    edges = [(1e-3, 10.0), (1.0, 10.0), (0.02, 2.0), (3e-3, 0.3)][int(rng.integers(0, 4))]
    form = "1" if k % 2 == 0 else "3"
    os.environ["RT_AMD_LEAF_AXES"] = form
    scene, tris = grazing.make_scene(rng, scale, targets=targets, edge_range=edges)
    scene.preprocess(max_leaf_tris=cap)
    od = [grazing.grazing_lines(rng, t, per_scene // targets, scale) for t in tris]
    o = np.concatenate([x[0] for x in od]); d = np.concatenate([x[1] for x in od])
    osc = binding.oracle().scene_create_with_tree(base_from_scene(scene), 64, 64, scene.bsp_tree)
    ds = DeviceScene(RayTraceData(scene))
    try:
        for front in (True, False):
            ef, eids = osc.intersect(o, d, 1e-3, front)
            for hits, st in ((ds.intersect(o, d, 1e-3, front), None), ds.intersect_check(o, d, 1e-3, front)):
                got_ids = np.stack([hits["mesh"], hits["tri"]], 1)
                got = np.concatenate([np.stack([hits["t"], hits["u"], hits["v"]], 1), hits["point"]], 1).astype(np.float32)
                bad = (got_ids != eids).any(1) | (got.view(np.uint32) != np.asarray(ef, np.float32).view(np.uint32)).any(1)
                tot["hit_records_differing"] += int(bad.sum())
                tot["queries"] += len(o)
                if st is not None:
                    tot["counted_disagreements"] += int(st["box_filter_errors"])
                    tot["leaf_visits"] += int(st["leaf_visits"])
                    tot["box_tests"] += int(st["box_tests"])
            tot["reference_hits"] += int((eids[:, 0] != 0xFFFFFFFF).sum())
        tot["lines"] += len(o)
        tot["scenes"] += 1
    finally:
        ds.close()
        osc.close()
    print(f"scene {k}: scale {scale:.3g}, {targets} targets, leaves of {cap}, records form {form}: "
          f"{tot['hit_records_differing']} differing, {tot['counted_disagreements']} disagreements so far", file=sys.stderr, flush=True)
tot["seconds"] = time.time() - t0
tot["first_seed"] = seed0
tot["note"] = ("lines: distinct (origin, direction) pairs; queries: lines x 2 surface modes x 2 kernels (production, counting); "
               "reference_hits: lines on which the oracle's restatement of the reference reports an intersection, summed over "
               "the two surface modes")
print(json.dumps(tot, indent=1))
