"""The scene-graph builder oracle (oracle/graph_build_ref.py) against fixtures produced by the reference's own
``compute_rel`` / ``SuncgDataset.__getitem__`` / ``suncg_collate_fn`` (oracle/gen_golden_graph.py)."""
import json
import random

import numpy as np

from conftest import load_golden
from oracle import graph_build_ref as G


def _fixture():
    g = load_golden("graph_build")
    meta = json.loads(bytes(g["meta"]).decode())
    rooms, names, sd, sd30 = G.synth_rooms(meta["n_rooms"], meta["seed"])
    assert names == meta["names"] and sd == meta["size_data"] and sd30 == meta["size_data_30"]
    return g, meta, rooms, names, sd, sd30


def test_compute_rel_matches_reference_on_quantised_pairs():
    g = load_golden("graph_build")
    pairs, want = g["rel_pairs"], g["rel_expected"]
    got = np.array([G.PRED[G.compute_rel(p[0], p[1])] for p in pairs], np.int32)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert len(set(want.tolist())) >= 10          # every reachable predicate occurs


def test_getitem_and_collate_match_reference():
    g, meta, rooms, names, sd, sd30 = _fixture()
    for tag, use30 in (("a", False), ("b", True)):
        table = G.RoomTable(rooms, names, sd, sd30, use_attr_30=use30)
        batch = []
        for idx, room in enumerate(rooms):
            random.seed(meta["getitem_seed_base"] + idx)
            draws = G.draw_room(len(room["objs"]), room["objs"], table)
            o, b, t, a, at = G.build_room(room, table, draws)
            key = "%s_room%02d_" % (tag, idx)
            assert np.array_equal(o, g[key + "objs"]) and np.array_equal(a, g[key + "angles"])
            assert np.array_equal(b, g[key + "boxes"]), idx
            assert np.array_equal(t, g[key + "triples"]), (idx, t, g[key + "triples"])
            assert np.array_equal(at, g[key + "attrs"]), (idx, at, g[key + "attrs"])
            batch.append((100 + idx, o, b, t, a, at))
        col = G.collate(batch)
        for k, v in zip(("ids", "objs", "boxes", "triples", "angles", "attrs", "obj_to_img", "triple_to_img"), col):
            want = g["%s_collate_%s" % (tag, k)]
            assert v.dtype == want.dtype and np.array_equal(v, want), k


def test_angle_sectors_by_comparison_equal_atan2():
    """the device kernel classifies the direction with comparisons instead of atan2; same decision on a lattice with ties"""
    vals = np.arange(-4, 5).astype(np.float32) * np.float32(0.25)
    for dx in vals:
        for dz in vals:
            assert G.sector_by_compare(dx, dz) == G.sector_by_atan2(dx, dz), (dx, dz)
    rng = np.random.default_rng(0)
    for dx, dz in rng.standard_normal((4000, 2)).astype(np.float32):
        assert G.sector_by_compare(dx, dz) == G.sector_by_atan2(dx, dz)
