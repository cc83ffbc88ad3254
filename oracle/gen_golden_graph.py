"""Generate tests/golden/graph_build.npz by running the REFERENCE's dataset code on CPU.

Runs ONLY in the build container (needs /root/reference, read-only).  Synthetic rooms (oracle/graph_build_ref.py::
synth_rooms) are written as json into a scratch directory in the layout the reference's ``SuncgDataset.__init__``
reads (data json + metadata/valid_types.json + metadata/size_info_many.json + metadata/30_size_info_many.json,
suncg_dataset.py:9-90); its real ``__init__``, ``__getitem__`` (under ``random.seed(1000 + index)``) and
``suncg_collate_fn`` produce the expected outputs.  ``compute_rel`` is additionally sampled on box pairs with
quantised coordinates.  Only data is written; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_graph.py
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

N_ROOMS, SEED, N_PAIRS = 48, 5, 6000


def rel_pairs(rng, n):
    """box pairs on a coarse lattice: equal coordinates, containment, overlap and touching all occur often"""
    lo = rng.integers(0, 6, size=(n, 2, 3)).astype(np.float32) * 0.25
    size = rng.integers(1, 5, size=(n, 2, 3)).astype(np.float32) * 0.25
    fine = rng.random(n) < 0.35                                        # a third with off-lattice jitter around the 0.05 'on' band
    lo[fine] += rng.uniform(-0.06, 0.06, size=(int(fine.sum()), 2, 3)).astype(np.float32)
    return np.concatenate([lo, lo + size], axis=2)                     # [n, 2, 6]


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    from oracle import graph_build_ref as G
    sys.path.insert(0, REF)
    import utils as ref_utils
    import data.suncg_dataset as ref_ds

    rooms, names, size_data, size_data_30 = G.synth_rooms(N_ROOMS, SEED)
    out = {}
    # ---- compute_rel on raw pairs (float32 tensors, as __getitem__ passes them) -----------------
    pairs = rel_pairs(np.random.default_rng(11), N_PAIRS)
    rel = np.zeros(N_PAIRS, np.int32)
    for i in range(N_PAIRS):
        p = ref_utils.compute_rel(torch.from_numpy(pairs[i, 0]), torch.from_numpy(pairs[i, 1]), None, None)
        rel[i] = G.PRED[p]
    out["rel_pairs"], out["rel_expected"] = pairs, rel

    # ---- the dataset: real __init__ over scratch json, __getitem__, collate ---------------------
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "metadata"))
        data = {}
        for r, room in enumerate(rooms):
            objs = [dict(type=names[c], new_bbox=[[float(x) for x in b[:3]], [float(x) for x in b[3:]]], rotation=int(a))
                    for c, b, a in zip(room["objs"], room["boxes"], room["rot"])]
            data[str(100 + r)] = dict(valid_objects=objs, bbox=[float(x) for x in room["bbox"]])
        json.dump(data, open(os.path.join(tmp, "rooms.json"), "w"))
        json.dump(names[1:], open(os.path.join(tmp, "metadata", "valid_types.json"), "w"))
        json.dump(size_data, open(os.path.join(tmp, "metadata", "size_info_many.json"), "w"))
        json.dump(size_data_30, open(os.path.join(tmp, "metadata", "30_size_info_many.json"), "w"))
        os.chdir(tmp)
        try:
            for tag, use30 in (("a", False), ("b", True)):
                with contextlib.redirect_stdout(io.StringIO()):
                    ds = ref_ds.SuncgDataset("rooms.json", True, use_attr_30=use30)
                assert ds.vocab["object_idx_to_name"] == names and ds.vocab["pred_idx_to_name"] == G.PRED_NAMES
                assert ds.vocab["attrib_idx_to_name"] == G.ATTR_NAMES
                batch = []
                for idx in range(len(ds)):
                    random.seed(1000 + idx)
                    batch.append(ds[idx])
                for idx, (rid, o, b, t, a, at) in enumerate(batch):
                    out["%s_room%02d_objs" % (tag, idx)] = o.numpy()
                    out["%s_room%02d_boxes" % (tag, idx)] = b.numpy()
                    out["%s_room%02d_triples" % (tag, idx)] = t.numpy().reshape(-1, 3)
                    out["%s_room%02d_angles" % (tag, idx)] = a.numpy()
                    out["%s_room%02d_attrs" % (tag, idx)] = at.numpy()
                col = ref_ds.suncg_collate_fn(batch)
                for k, v in zip(("ids", "objs", "boxes", "triples", "angles", "attrs", "obj_to_img", "triple_to_img"), col):
                    out["%s_collate_%s" % (tag, k)] = v.numpy()
        finally:
            os.chdir(cwd)
    out["meta"] = np.frombuffer(json.dumps(dict(n_rooms=N_ROOMS, seed=SEED, names=names, size_data=size_data,
                                                size_data_30=size_data_30, getitem_seed_base=1000)).encode(), dtype=np.uint8)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "graph_build.npz"), **out)
    n_on = sum(int((out["a_room%02d_triples" % i][:, 1] == G.PRED["on"]).sum()) for i in range(N_ROOMS))
    print("wrote graph_build.npz: %d rooms, %d collated triples (%d 'on'), rel histogram %s" %
          (N_ROOMS, out["a_collate_triples"].shape[0], n_on, np.bincount(rel, minlength=16).tolist()))


if __name__ == "__main__":
    main()
