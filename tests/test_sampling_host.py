"""Host-side tensor helpers of host/sampling.py that need no GPU, against fixtures made by executing the reference's own source
(oracle/gen_golden_sampling.py): the heat-map accumulation (testing/test_heatmap.py:74-99) and get_sg_from_words."""
import numpy as np
import torch

from conftest import load_golden, pkg


def test_layout_heatmap_equals_the_reference_loop():
    """host/sampling.py::layout_heatmap against the maps the reference's own accumulation loop produced (testing/test_heatmap.py:74-99,
    executed from its source by oracle/gen_golden_sampling.py with matplotlib replaced by a recorder): clipped and rejecting variants,
    room rows that are not the unit box."""
    S = pkg("host.sampling")
    g = load_golden("sampling_helpers")
    for tag, clip in (("clip", True), ("reject", False)):
        boxes, want, size = g["heat_%s:boxes" % tag], g["heat_%s:maps" % tag], int(g["heat_%s:size" % tag])
        got = S.layout_heatmap(torch.from_numpy(boxes), size, clip).numpy()
        assert got.shape == want.shape
        assert np.allclose(got, want, atol=1e-7), tag
        assert np.allclose(got.sum((1, 2))[want.sum((1, 2)) > 0], 1.0)


def test_scene_graph_from_words_equals_the_reference_function():
    """... and scene_graph_from_words against get_sg_from_words (testing/test_utils.py:43-90, executed from its source)"""
    S = pkg("host.sampling")
    from oracle.gen_golden_sampling import SCENES
    g = load_golden("sampling_helpers")
    for i, (objs, rels) in enumerate(SCENES):
        o, t, a = S.scene_graph_from_words(objs, rels)
        assert np.array_equal(o.numpy(), g["sg%d:objs" % i]) and np.array_equal(t.numpy(), g["sg%d:triples" % i])
        assert np.array_equal(a.numpy(), g["sg%d:attributes" % i])
        assert o.dtype == torch.int64 and t.dtype == torch.int64 and a.dtype == torch.int64


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
