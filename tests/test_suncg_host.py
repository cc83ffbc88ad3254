"""Host-side logic of host/suncg_dataset.py that needs no GPU: the python-``random`` draw order (against the oracle's
``draw_room``, which the reference fixture pins) and ``suncg_collate_fn`` (against the reference's collated fixture)."""
import json
import random

import numpy as np
import torch

from conftest import load_golden, pkg
from oracle import graph_build_ref as G


def _fixture():
    g = load_golden("graph_build")
    meta = json.loads(bytes(g["meta"]).decode())
    rooms, names, sd, sd30 = G.synth_rooms(meta["n_rooms"], meta["seed"])
    return g, meta, rooms, names, sd, sd30


def test_draw_consumes_python_random_like_the_reference():
    g, meta, rooms, names, sd, sd30 = _fixture()
    D = pkg("host.suncg_dataset")
    for use30 in (False, True):
        ds = D.SuncgDataset.from_tables(rooms, names, sd, sd30, use_attr_30=use30, device="cpu")
        table = G.RoomTable(rooms, names, sd, sd30, use_attr_30=use30)
        random.seed(7)
        other, swap, mode = ds.draw(list(range(len(rooms))))
        state_after = random.getstate()
        random.seed(7)
        pos = 0
        for room in rooms:
            n = len(room["objs"])
            o, s, u1, u2 = G.draw_room(n, room["objs"], table)
            assert np.array_equal(other[pos:pos + n], o) and np.array_equal(swap[pos:pos + n].astype(bool), s)
            want = np.where((u1 > 0.5) | np.isnan(u2), 0, np.where(u2 > 0.5, 1, 2))
            assert np.array_equal(mode[pos:pos + n], want)
            pos += n
        assert pos == other.shape[0] and random.getstate() == state_after       # same number of draws consumed


def test_collate_fn_matches_reference_fixture():
    g, meta, rooms, names, sd, sd30 = _fixture()
    D = pkg("host.suncg_dataset")
    batch = [(100 + i,) + tuple(torch.from_numpy(g["a_room%02d_%s" % (i, k)]) for k in ("objs", "boxes", "triples", "angles", "attrs"))
             for i in range(len(rooms))]
    col = D.suncg_collate_fn(batch)
    for k, v in zip(("ids", "objs", "boxes", "triples", "angles", "attrs", "obj_to_img", "triple_to_img"), col):
        want = g["a_collate_" + k]
        assert v.numpy().dtype == want.dtype and np.array_equal(v.numpy(), want), k
    assert batch[3][3][0, 0] == int(g["a_room03_triples"][0, 0])                 # the inputs are not modified (triples are cloned)
