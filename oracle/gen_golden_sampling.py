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


def gen_spade_input():
    """`colorize_with_spade` (testing/test_SPADE_shade.py:30-79) and `save_color` (:16-27) executed from their source: imageio serves the
    procedural scene of oracle/spade_input_ref.py::synth_scene at the 1 024^2 the function hard-codes, the generator is a recorder (it
    receives `total`, the [1,41,256,256] tensor under test), skimage's resize - not installed here, PARITY UNPINNED - is the scipy
    restatement of oracle/spade_input_ref.py."""
    import tempfile
    from oracle import spade_input_ref as sir
    from oracle.gen_golden_refine import _neutralise
    _neutralise()
    th = os.path.join(REF, "testing", "test_SPADE_shade.py")
    depth, masks = sir.synth_scene(1024, seed=2)
    d = tempfile.mkdtemp(prefix="sln_spade_input_")
    files = {"room_0000_depth.exr": np.stack([depth] * 3, -1)}
    for name, m in masks.items():
        files["room_0000_mask_%s.png" % name] = np.stack([m] * 3, -1)
    for f in files:
        open(os.path.join(d, f), "w").close()
    seen, written = [], []

    class Writer:
        def append_data(self, a): written.append(np.array(a))
        def close(self): pass
    fake = torch.from_numpy(np.random.default_rng(9).uniform(-1, 1, size=(1, 3, 16, 16)).astype(np.float32))
    ns = dict(np=np, torch=torch, os=os, print=_quiet, resize=lambda a, shape, **k: sir.resize_skimage(a, shape),
              imageio=types.SimpleNamespace(imread=lambda p: files[os.path.basename(p)], get_writer=lambda *a, **k: Writer()),
              colorization_model=lambda total, z: (seen.append(total.clone()), fake)[1])
    _run(_top_level(th, ["save_color", "colorize_with_spade"]).values(), ns, th)
    torch.manual_seed(0)
    ns["colorize_with_spade"](2, d, tempfile.mkdtemp(prefix="sln_spade_out_"))
    assert len(seen) == 2 and torch.equal(seen[0], seen[1]) and seen[0].shape == (1, 41, 256, 256) and len(written) == 2
    total = seen[0][0].numpy()
    live = np.nonzero(np.abs(total).reshape(41, -1).max(1) > 1e-6)[0]          # (the spline prefilter along the channel axis leaves ~1e-17 in empty channels)
    out = {"total_channels": live.astype(np.int64), "total_half": total[live][:, ::2, ::2].astype(np.float32),
           "total_sums": total.astype(np.float64).reshape(41, -1).sum(1), "total_abs_sums": np.abs(total.astype(np.float64)).reshape(41, -1).sum(1),
           "save_color_in": fake.numpy(), "save_color_out": written[0]}
    np.savez_compressed(os.path.join(GOLD, "spade_input.npz"), **out)
    print("wrote spade_input.npz: live channels", live.tolist())


if __name__ == "__main__":
    main()
    gen_spade_input()
