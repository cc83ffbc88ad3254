"""Golden fixture for the posterior-sampling / heat-map helpers, produced by EXECUTING THE REFERENCE'S SOURCE
(build container only): `get_sg_from_words` (testing/test_utils.py:43-90; the module imports shapely, so the function node is taken
from the source text with `ast`, as oracle/gen_golden_refine.py does) and the accumulation loop of `plot_heatmap`
(testing/test_heatmap.py:74-99: the `for obj_type ...` statement, executed with matplotlib replaced by a recorder - every
`plt.imshow(container, ...)` call hands over one object's finished map).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_sampling.py

Writes tests/golden/sampling_helpers.npz (numeric arrays only; the word lists are in tests/test_sampling_host.py).
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle.gen_golden_refine import GOLD, REF, _run, _top_level, _quiet      # noqa: E402

# the scenes of tests/test_sampling_host.py (first: testing/test_heatmap.py:41-44)
SCENES = [(["bed", "desk", "cabinet", "chair", "lamp"], [("bed", "behind", "desk"), ("cabinet", "left of", "bed"), ("chair", "left of", "desk"), ("lamp", "on", "desk")]),
          (["chair:0", "chair:1"], [("chair:0", "left of", "chair:1")]),
          (["sofa", "television", "table", "lamp:0", "lamp:1", "bookshelf"], [("television", "in front of", "sofa"), ("table", "surrounding", "lamp:0"), ("lamp:1", "on", "bookshelf"),
                                                                                ("bookshelf", "back right", "sofa"), ("table", "front touching", "sofa")])]


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    out = {}
    tu = os.path.join(REF, "testing", "test_utils.py")
    ns = dict(torch=torch, np=np)
    _run(_top_level(tu, ["get_sg_from_words"]).values(), ns, tu)
    for i, (objs, rels) in enumerate(SCENES):
        o, t, a = ns["get_sg_from_words"](objs, rels)
        out["sg%d:objs" % i], out["sg%d:triples" % i], out["sg%d:attributes" % i] = o.numpy(), t.numpy(), a.numpy()
    # the accumulation loop of plot_heatmap: the `for obj_type in range(len(heat_pkl[2][0]) - 1)` statement
    th = os.path.join(REF, "testing", "test_heatmap.py")
    fn = _top_level(th, ["plot_heatmap"])["plot_heatmap"]
    loops = [n for n in ast.walk(fn) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == "obj_type"]
    assert len(loops) == 1
    rng = np.random.default_rng(5)
    for tag, clip, size, trials, O in (("clip", True, 100, 400, 6), ("reject", False, 50, 300, 5)):
        boxes = rng.uniform(-0.25, 1.25, size=(trials, O, 6)).astype(np.float32)
        lo = rng.uniform(0.0, 0.1, size=(trials, 3)); hi = rng.uniform(0.9, 1.1, size=(trials, 3))
        boxes[:, -1] = np.concatenate([lo, hi], 1)                        # the room row normalises the others (test_heatmap.py:84-85)
        maps = []
        plt = types.SimpleNamespace(imshow=lambda c, **k: maps.append(np.array(c, np.float64)), tight_layout=_quiet, savefig=_quiet, show=_quiet, close=_quiet,
                                    gca=lambda: types.SimpleNamespace(axes=types.SimpleNamespace(get_yaxis=lambda: types.SimpleNamespace(set_visible=_quiet),
                                                                                                 get_xaxis=lambda: types.SimpleNamespace(set_visible=_quiet))))
        env = dict(np=np, os=os, plt=plt, print=_quiet, heat_pkl=[None, None, [list(b) for b in boxes], []], container_size=size, clip_coor=clip,
                   visualize=False, save_dir="/tmp", heat_pkl_idx="0000")
        exec(compile(ast.fix_missing_locations(ast.Module(body=[loops[0]], type_ignores=[])), th, "exec"), env)
        assert len(maps) == O - 1
        out["heat_%s:boxes" % tag], out["heat_%s:maps" % tag], out["heat_%s:size" % tag] = boxes, np.stack(maps), np.int64(size)
    np.savez_compressed(os.path.join(GOLD, "sampling_helpers.npz"), **out)
    print("wrote sampling_helpers.npz: %d arrays" % len(out))


if __name__ == "__main__":
    main()
