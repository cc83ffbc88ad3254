"""Host-side tensor helpers of host/sampling.py that need no GPU: the heat-map accumulation against a numpy
restatement of the reference's loop (testing/test_heatmap.py:80-99)."""
import numpy as np
import torch

from conftest import pkg


def _reference_loop(boxes, size, clip):
    n, O, _ = boxes.shape
    out = np.zeros((O - 1, size, size))
    for obj in range(O - 1):
        for trial in range(n):
            bb, nz = boxes[trial][obj], boxes[trial][-1]
            bb = np.array(bb) * np.concatenate([nz[3:] - nz[:3], nz[3:] - nz[:3]])
            ct = (bb[:3] + bb[3:]) * 0.5
            if clip:
                ct = np.clip(ct, 0.0, 1.0)
            elif not (np.all(ct > 0.0) and np.all(ct < 1.0)):
                continue
            rd = np.floor(ct * (size - 1)).astype("int")
            out[obj, rd[2], rd[0]] += 1.0
        out[obj] = out[obj] / max(np.sum(out[obj]), 1.0)
    return out


def test_layout_heatmap_equals_reference_loop():
    S = pkg("host.sampling")
    rng = np.random.default_rng(0)
    boxes = rng.uniform(-0.2, 1.2, size=(300, 5, 6)).astype(np.float32)
    boxes[:, -1] = [0, 0, 0, 1, 1, 1]
    for clip in (True, False):
        got = S.layout_heatmap(torch.from_numpy(boxes), 50, clip).numpy()
        want = _reference_loop(boxes, 50, clip)
        assert np.allclose(got, want, atol=1e-7), clip
        assert np.allclose(got.sum((1, 2))[want.sum((1, 2)) > 0], 1.0)


def test_scene_graph_from_words_layout():
    """testing/test_utils.py:43-90 on the example of test_heatmap.py:41-44"""
    S = pkg("host.sampling")
    objs5 = ["bed", "desk", "cabinet", "chair", "lamp"]
    rels5 = [("bed", "behind", "desk"), ("cabinet", "left of", "bed"), ("chair", "left of", "desk"), ("lamp", "on", "desk")]
    objs, triples, attrs = S.scene_graph_from_words(objs5, rels5)
    assert objs.tolist() == [30, 11, 18, 9, 13, 0] and attrs.tolist() == [0] * 6
    assert triples.tolist() == [[0, 3, 1], [2, 1, 0], [3, 1, 1], [4, 15, 1]] + [[i, 0, 5] for i in range(5)]
    o2, t2, _ = S.scene_graph_from_words(["chair:0", "chair:1"], [("chair:0", "left of", "chair:1")])
    assert o2.tolist() == [9, 9, 0] and t2.tolist() == [[0, 1, 1], [0, 0, 2], [1, 0, 2]]
