"""CPU oracle for the scene-graph builder (the caller that feeds path A).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  numpy/float32 restatement of

  * ``compute_rel``                         /root/reference/utils.py:36-80
  * ``SuncgDataset.__getitem__``            /root/reference/data/suncg_dataset.py:110-307 (train_3d=True)
  * ``suncg_collate_fn``                    /root/reference/data/suncg_dataset.py:310-353

Parity status: PINNED - ``oracle/gen_golden_graph.py`` runs the reference's own ``SuncgDataset`` (its real
``__init__`` over synthetic json written to a scratch directory), ``__getitem__`` under ``random.seed(k)`` and
``suncg_collate_fn``, plus ``compute_rel`` on 6 k box pairs with quantised coordinates (ties on every comparison),
and writes ``tests/golden/graph_build.npz``; ``tests/test_oracle_graph_build.py`` holds this file to it.

Arithmetic notes (all verified by the fixture): boxes are float32 (``torch.FloatTensor``), every difference /
product / quotient of box coordinates is a float32 operation evaluated left to right, python constants compared with
them are first rounded to float32 (torch/numpy weak-scalar promotion), only ``atan2`` runs in double.

The random draws of ``__getitem__`` come from python's global ``random`` in this order: per real object one
``random.choice`` + one ``random.random`` (subject/object swap); then per non-room object one ``random.random``
(attribute at all?) and, when that one is <= 0.5 and the class has size statistics, a second one (height or volume).
``draw_room`` reproduces the stream so that the device path can be fed the same decisions.
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Sequence

import numpy as np

PRED_NAMES = ['__in_room__', 'left of', 'right of', 'behind', 'in front of', 'inside', 'surrounding', 'left touching',
              'right touching', 'front touching', 'behind touching', 'front left', 'front right', 'back left', 'back right', 'on']
PRED = {n: i for i, n in enumerate(PRED_NAMES)}
ATTR_NAMES = ['none', 'tall', 'short', 'large', 'small']
ATTR = {n: i for i, n in enumerate(ATTR_NAMES)}
F = np.float32


def compute_rel(box1, box2, name2=None) -> str:
    """utils.py:36-80 on float32 boxes [x0,y0,z0,x1,y1,z1]."""
    if name2 == "__room__":
        return "__in_room__"
    b1 = np.asarray(box1, dtype=F); b2 = np.asarray(box2, dtype=F)
    two = F(2)
    c1 = [(b1[0] + b1[3]) / two, (b1[1] + b1[4]) / two, (b1[2] + b1[5]) / two]
    c2 = [(b2[0] + b2[3]) / two, (b2[1] + b2[4]) / two, (b2[2] + b2[5]) / two]
    if c1[0] >= b2[0] and c1[0] <= b2[3] and c1[2] >= b2[2] and c1[2] <= b2[5]:
        delta1 = c1[1] - c2[1]
        delta2 = (b1[4] - b1[1] + b2[4] - b2[1]) / two
        if abs(delta1 - delta2) < F(0.05):
            return "on"
    sx0, sy0, sz0, sx1, sy1, sz1 = b1
    ox0, oy0, oz0, ox1, oy1, oz1 = b2
    dx, dz = c1[0] - c2[0], c1[2] - c2[2]
    theta = math.atan2(float(dz), float(dx))
    area_s = (sx1 - sx0) * (sz1 - sz0)
    area_o = (ox1 - ox0) * (oz1 - oz0)
    ix0, ix1 = max(sx0, ox0), min(sx1, ox1)
    iz0, iz1 = max(sz0, oz0), min(sz1, oz1)
    area_i = max(F(0), ix1 - ix0) * max(F(0), iz1 - iz0)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = area_i / (area_s + area_o - area_i)
    touching = bool(F(0.0001) < iou) and bool(iou < F(0.5))
    if sx0 < ox0 and sx1 > ox1 and sz0 < oz0 and sz1 > oz1:
        return "surrounding"
    if sx0 > ox0 and sx1 < ox1 and sz0 > oz0 and sz1 < oz1:
        return "inside"
    if theta >= 3 * math.pi / 4 or theta <= -3 * math.pi / 4:
        return "right touching" if touching else "left of"
    if -3 * math.pi / 4 <= theta < -math.pi / 4:
        return "behind touching" if touching else "behind"
    if -math.pi / 4 <= theta < math.pi / 4:
        return "left touching" if touching else "right of"
    if math.pi / 4 <= theta < 3 * math.pi / 4:
        return "front touching" if touching else "in front of"
    return None


def sector_by_atan2(dx, dz) -> int:
    """0 left, 1 behind, 2 right, 3 front - the four direction branches of compute_rel (utils.py:70-77)."""
    theta = math.atan2(float(dz), float(dx))
    if theta >= 3 * math.pi / 4 or theta <= -3 * math.pi / 4:
        return 0
    if -3 * math.pi / 4 <= theta < -math.pi / 4:
        return 1
    if -math.pi / 4 <= theta < math.pi / 4:
        return 2
    return 3


def sector_by_compare(dx, dz) -> int:
    """The same decision without atan2 (what the HIP kernel evaluates): float32 inputs differ from a sector boundary by
    at least 2^-24 relative, far above atan2's rounding, so the comparisons agree with the double-precision angles."""
    dx, dz = F(dx), F(dz)
    if dx < 0 and abs(dz) <= -dx:
        return 0
    if dz < 0 and abs(dx) < -dz:
        return 1
    if (dx > 0 and -dx <= dz < dx) or (dx == 0 and dz == 0):
        return 2
    return 3


def rel_matrix(boxes: np.ndarray) -> np.ndarray:
    """pred id of compute_rel(boxes[i], boxes[j]) for every ordered pair of REAL objects (diagonal = -1)."""
    n = boxes.shape[0]
    out = -np.ones((n, n), dtype=np.int32)
    for i in range(n):
        for j in range(n):
            if i != j:
                out[i, j] = PRED[compute_rel(boxes[i], boxes[j])]
    return out


class RoomTable:
    """The dataset-side state ``__getitem__`` reads: rooms (class, raw bbox, rotation, room bbox), vocabulary and the
    size statistics.  ``rooms``: list of dicts {"objs": [n] int, "boxes": [n,6] f32 raw, "rot": [n] int, "bbox": [3] f32}."""

    def __init__(self, rooms, object_idx_to_name: Sequence[str], size_data: Dict, size_data_30: Dict, use_attr_30=False):
        self.rooms = rooms
        self.object_idx_to_name = list(object_idx_to_name)
        self.size_data, self.size_data_30, self.use_attr_30 = size_data, size_data_30, use_attr_30

    def has_size(self, cls: int) -> bool:
        name = self.object_idx_to_name[cls]
        return name in (self.size_data_30 if self.use_attr_30 else self.size_data)


def draw_room(n_real: int, classes: Sequence[int], table: RoomTable, rng=random):
    """Consume python's random stream exactly as ``__getitem__`` does for a room with ``n_real`` non-room objects
    (needs n_real >= 2 for the relationship part, like the reference).  Returns (other[n], swap[n], u1[n], u2[n]);
    ``swap`` True means ``s, o = cur, other`` (random() > 0.5); ``u2`` is NaN where the reference draws nothing."""
    other = np.zeros(n_real, np.int32); swap = np.zeros(n_real, np.bool_)
    for cur in range(n_real):
        choices = [k for k in range(n_real) if k != cur]
        other[cur] = rng.choice(choices)
        swap[cur] = rng.random() > 0.5
    u1 = np.zeros(n_real, np.float64); u2 = np.full(n_real, np.nan, np.float64)
    for i in range(n_real):
        u1[i] = rng.random()
        if not (u1[i] > 0.5 or not table.has_size(int(classes[i]))):
            u2[i] = rng.random()
    return other, swap, u1, u2


def build_room(room: dict, table: RoomTable, draws):
    """__getitem__ (suncg_dataset.py:110-307) for one room given the random decisions ``draws`` (see draw_room).
    Returns (objs i64[n+1], boxes f32[n+1,6] normalised, triples i64[T,3], angles i64[n+1], attributes i64[n+1])."""
    other, swap, u1, u2 = draws
    n = len(room["objs"])
    objs = np.concatenate([np.asarray(room["objs"], np.int64), np.zeros(1, np.int64)])          # '__room__' = 0
    rb = np.asarray(room["bbox"], dtype=F)
    boxes = np.concatenate([np.asarray(room["boxes"], dtype=F).reshape(n, 6), np.array([[0, 0, 0, rb[0], rb[1], rb[2]]], dtype=F)])
    angles = np.concatenate([np.asarray(room["rot"], np.int64), np.zeros(1, np.int64)])
    triples: List[List[int]] = []
    if n + 1 > 1:
        for cur in range(n):
            for oth in range(n):
                if oth != cur and compute_rel(boxes[cur], boxes[oth]) == "on":
                    triples.append([cur, PRED["on"], oth])
        for cur in range(n):
            if n < 2:
                break
            s, o = (cur, int(other[cur])) if swap[cur] else (int(other[cur]), cur)
            # The reference means to skip pairs already linked by 'on' (suncg_dataset.py:202), but ``on_rels`` is keyed by
            # 0-dim tensors (hashed by identity), so the lookup never hits: nothing is skipped and an 'on' pair drawn here
            # is emitted a second time.  The fixture pins that behaviour; it is kept.
            triples.append([s, PRED[compute_rel(boxes[s], boxes[o])], o])
        for i in range(n):
            triples.append([i, PRED["__in_room__"], n])
    triples = np.asarray(triples, np.int64).reshape(-1, 3)
    for i in range(n):
        boxes[i, 0] /= boxes[n, 3]; boxes[i, 3] /= boxes[n, 3]
        boxes[i, 1] /= boxes[n, 4]; boxes[i, 4] /= boxes[n, 4]
        boxes[i, 2] /= boxes[n, 5]; boxes[i, 5] /= boxes[n, 5]
    attrs = []
    for i in range(n):
        name = table.object_idx_to_name[int(objs[i])]
        if u1[i] > 0.5 or not table.has_size(int(objs[i])):
            attrs.append("none"); continue
        height = boxes[i, 4] - boxes[i, 1]
        volume = (boxes[i, 3] - boxes[i, 0]) * (boxes[i, 4] - boxes[i, 1]) * (boxes[i, 5] - boxes[i, 2])
        if not table.use_attr_30:
            sd = table.size_data[name]
            if u2[i] > 0.5:
                attrs.append("tall" if height > F(sd[0][1]) else "short")
            else:
                attrs.append("large" if volume > F(sd[1]) else "small")
        else:
            sd = table.size_data_30[name]
            if u2[i] > 0.5:
                attrs.append("tall" if height > F(sd["height_7"]) else ("short" if height < F(sd["height_3"]) else "none"))
            else:
                attrs.append("large" if volume > F(sd["volume_7"]) else ("small" if volume < F(sd["volume_3"]) else "none"))
    attrs.append("none")
    return objs, boxes, triples, angles, np.asarray([ATTR[a] for a in attrs], np.int64)


def collate(batch):
    """suncg_collate_fn (suncg_dataset.py:310-353) over [(room_id, objs, boxes, triples, angles, attributes), ...]."""
    ids, objs, boxes, trip, ang, att, o2r, t2r = [], [], [], [], [], [], [], []
    off = 0
    for i, (rid, o, b, t, a, at) in enumerate(batch):
        O, T = o.shape[0], t.shape[0]
        ids.append(rid); objs.append(o); boxes.append(b); ang.append(a); att.append(at)
        t = t.copy(); t[:, 0] += off; t[:, 2] += off
        trip.append(t)
        o2r.append(np.full(O, i, np.int64)); t2r.append(np.full(T, i, np.int64))
        off += O
    cat = np.concatenate
    return (np.asarray(ids, np.int64), cat(objs), cat(boxes), cat(trip), cat(ang), cat(att), cat(o2r), cat(t2r))


# ----------------------------------------------------------------------------------------------
def synth_rooms(n_rooms: int, seed: int = 0, n_classes: int = 12, max_objs: int = 14, quantum: float = 0.25):
    """Synthetic rooms with the structure of data_rot_*.json (suncg_dataset.py:84-90): floor-standing boxes, some stacked
    exactly on another one ('on'), some nested in plan view ('inside'/'surrounding'), coordinates snapped to ``quantum``
    so that the comparisons of compute_rel meet ties.  Returns (rooms, object_idx_to_name, size_data, size_data_30)."""
    rng = np.random.default_rng(seed)
    names = ["__room__"] + ["type%02d" % i for i in range(n_classes)]
    q = lambda v: np.round(np.asarray(v) / quantum) * quantum
    rooms = []
    for r in range(n_rooms):
        room = q(rng.uniform([3, 2.5, 3], [7, 3.25, 8]))
        n = int(rng.integers(2, max_objs + 1))
        boxes, cls, rot = [], [], []
        for i in range(n):
            kind = rng.random()
            size = np.maximum(q(rng.uniform([0.25, 0.25, 0.25], [2.0, 1.5, 2.0])), quantum)
            if i > 0 and kind < 0.2:                                  # stacked on a previous object
                b = boxes[int(rng.integers(0, i))]
                cx, cz = (b[0] + b[3]) / 2, (b[2] + b[5]) / 2
                size = np.minimum(size, [b[3] - b[0], 1.0, b[5] - b[2]]); size = np.maximum(size, quantum)
                lo = np.array([cx - size[0] / 2, b[4] + (0.0 if rng.random() < 0.7 else 0.03), cz - size[2] / 2])
            elif i > 0 and kind < 0.35:                               # strictly inside a previous object's footprint
                b = boxes[int(rng.integers(0, i))]
                size = np.array([(b[3] - b[0]) / 2, size[1], (b[5] - b[2]) / 2])
                lo = np.array([b[0] + (b[3] - b[0]) / 4, 0.0, b[2] + (b[5] - b[2]) / 4])
            else:
                lo = q(rng.uniform([0, 0, 0], np.maximum(room - size, 0.0))); lo[1] = 0.0 if rng.random() < 0.8 else lo[1]
            boxes.append(np.concatenate([lo, lo + size]).astype(np.float32))
            cls.append(int(rng.integers(1, n_classes + 1))); rot.append(int(rng.integers(0, 24)))
        rooms.append(dict(objs=cls, boxes=np.stack(boxes), rot=rot, bbox=room.astype(np.float32)))
    with_size = names[1:1 + (2 * n_classes) // 3]                      # a third of the classes has no size statistics
    size_data = {nm: [[0.1, float(q(rng.uniform(0.1, 0.4)))], float(rng.uniform(0.001, 0.02))] for nm in with_size}
    size_data_30 = {nm: dict(height_7=float(q(rng.uniform(0.25, 0.5))), height_3=float(q(rng.uniform(0.0, 0.25))),
                             volume_7=float(rng.uniform(0.01, 0.03)), volume_3=float(rng.uniform(0.0005, 0.01))) for nm in with_size}
    return rooms, names, size_data, size_data_30
