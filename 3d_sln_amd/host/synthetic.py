"""Synthetic scene-graph batches with the tuple layout of the reference's ``suncg_collate_fn``
(data/suncg_dataset.py:295-337): objects of one room are contiguous, the room node (class 0) is
last, every object has an ``__in_room__`` (predicate 0) triple to it (:207-212), and triples carry
global row ids.  Used by bench.py / train.py because the SUNCG metadata is not distributable."""
import numpy as np
import torch


def scene_graph_batch(n_graphs, objs_per_graph=32, triples_per_graph=64, seed=0, num_objs=32, num_preds=16,
                      num_attrs=5, n_angle=24, box_dim=6, device="cpu"):
    rng = np.random.default_rng(seed)
    n, tt = objs_per_graph, triples_per_graph
    n_rand = tt - (n - 1)
    if n < 2 or n_rand < 0:
        raise ValueError("need >= 2 objects and >= objs-1 triples per graph")
    objs = rng.integers(1, num_objs, size=(n_graphs, n)); objs[:, -1] = 0
    s = rng.integers(0, n - 1, size=(n_graphs, n_rand))
    o = (s + rng.integers(1, max(n - 1, 2), size=(n_graphs, n_rand))) % (n - 1) if n > 2 else s.copy()
    p = rng.integers(1, num_preds, size=(n_graphs, n_rand))
    off = (np.arange(n_graphs) * n)[:, None]
    rand = np.stack([s + off, p, o + off], -1)
    room = np.stack([np.arange(n - 1)[None] + off, np.zeros((n_graphs, n - 1), np.int64),
                     np.full((n_graphs, n - 1), n - 1) + off], -1)
    triples = np.concatenate([rand, room], 1).reshape(-1, 3)
    lo = rng.uniform(0.0, 0.7, size=(n_graphs, n, 3)); hi = lo + rng.uniform(0.05, 0.3, size=(n_graphs, n, 3))
    boxes = np.concatenate([lo, hi], -1); boxes[:, -1] = [0, 0, 0, 1, 1, 1]
    if box_dim == 4:
        boxes = boxes[..., [0, 2, 3, 5]]
    angles = rng.integers(0, n_angle, size=(n_graphs, n))
    attrs = rng.integers(0, num_attrs, size=(n_graphs, n))
    o2i = np.repeat(np.arange(n_graphs), n)
    t2i = np.repeat(np.arange(n_graphs), tt)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(device)
    return dict(objs=t(objs.reshape(-1), np.int64), triples=t(triples, np.int64),
                boxes=t(boxes.reshape(-1, box_dim), np.float32), angles=t(angles.reshape(-1), np.int64),
                attributes=t(attrs.reshape(-1), np.int64), obj_to_img=t(o2i, np.int64), triple_to_img=t(t2i, np.int64))


def default_vocab(num_objs=32, num_preds=16, num_attrs=5):
    return {"object_idx_to_name": ["__room__"] + ["type%02d" % i for i in range(1, num_objs)],
            "pred_idx_to_name": ["pred%02d" % i for i in range(num_preds)],
            "attrib_idx_to_name": ["attr%d" % i for i in range(num_attrs)]}
