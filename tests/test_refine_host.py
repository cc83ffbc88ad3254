"""Host logic of the refinement path that needs no GPU: the room-shell placement of `MeshBank.from_arrays(..., shell=)` against the
oracle's restatement (which reproduces the reference's own `mesh_render_func` bit for bit, tests/test_oracle_refine_golden.py), the
shell's buffer order, the camera, and the argument checks of the mesh seam."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from oracle import raster_ref, refine_ref as rf


def _bank(g):
    R = pkg("host.refine")
    t = rf.load_tables(g)
    meshes = {k: (m["v"], m["f"], m["bbox_min"], m["bbox_max"]) for k, m in t["models"].items()}
    return R.MeshBank.from_arrays(meshes, "cpu", vocab=t["vocab"], shell=t["shell"]), t


@pytest.mark.parametrize("ext", [(4.0, 2.7, 5.0), (6.5, 3.1, 3.2), (2.9, 2.4, 8.0)])
def test_shell_placement_equals_the_oracle_restatement(ext):
    R = pkg("host.refine")
    bank, t = _bank(load_golden("refine_scene"))
    got = R.place_shell(bank, ext)
    topo = R.shell_topology(bank)
    want = rf.place_shell(torch.tensor(ext), t["shell"])
    assert [n for n, _, _ in topo] == ["wall"] * len(t["shell"]["wall_f"]) + ["floor", "ceiling"]       # the reference's buffer order
    kept = {id(v) for _, v, _ in want}
    at, k = 0, 0
    dropped = 0
    for (name, nv, f), src_f in zip(topo, t["shell"]["wall_f"] + [t["shell"]["floor_f"], t["shell"]["ceil_f"]]):
        part = got[at:at + nv]; at += nv
        assert np.array_equal(f, np.asarray(src_f))
        if k < len(want) and want[k][0] == name and want[k][1].shape[0] == nv and torch.equal(want[k][2], torch.from_numpy(np.asarray(src_f)).long()):
            assert torch.equal(part, want[k][1]), "%s sub-mesh differs from the oracle's placement" % name
            k += 1
        else:                                           # a wall the bad-wall rule drops (diff_render.py:203-213): collapsed to a point here
            assert name == "wall" and float(part.abs().max()) == 0.0
            dropped += 1
    assert k == len(want) and at == got.shape[0]
    assert dropped == len(topo) - len(want)


def test_the_wall_in_front_of_the_camera_is_dropped_and_only_that_one():
    R = pkg("host.refine")
    bank, t = _bank(load_golden("refine_scene"))
    got = R.place_shell(bank, (4.0, 2.7, 5.0))
    nv = t["shell"]["wall_v"].shape[0]
    zero = [float(got[i * nv:(i + 1) * nv].abs().max()) == 0.0 for i in range(len(t["shell"]["wall_f"]))]
    assert zero == [False, False, False, True]          # back, left, right kept; the wall at z = Z, centred in x, goes


def test_camera_on_the_host_matches_the_reference_function():
    DR = pkg("host.diff_render")
    g = load_golden("refine_helpers")
    for i, room in enumerate(g["cam:rooms"]):
        K, R_, t = DR.get_cam_mat([np.zeros(6), room], "cpu")
        assert np.allclose(K[0].numpy(), g["cam:K"][i], rtol=1e-7, atol=0) and np.allclose(R_[0].numpy(), g["cam:R"][i], rtol=1e-7, atol=1e-8)
        assert np.allclose(t[0].numpy(), g["cam:t"][i], rtol=1e-6, atol=1e-7)
        K2, R2, t2 = raster_ref.get_cam_mat(torch.from_numpy(room))
        assert torch.equal(K2, K) and torch.allclose(R2, R_) and torch.allclose(t2, t)


def test_mesh_seam_rejects_bad_arrays():
    R = pkg("host.refine")
    v = np.zeros((4, 3), np.float32); f = np.array([[0, 1, 2]], np.int32)
    with pytest.raises(ValueError):
        R.MeshBank.from_arrays({"bed": (np.zeros((0, 3), np.float32), f)}, "cpu")
    with pytest.raises(IndexError):
        R.MeshBank.from_arrays({"bed": (v, np.array([[0, 1, 9]], np.int32))}, "cpu")
    shell = dict(wall_v=v, wall_f=[np.array([[0, 1, 7]])], wall_bbox=np.zeros((2, 3)), floor_v=v, floor_f=f, floor_bbox=np.zeros((2, 3)), ceil_v=v, ceil_f=f)
    with pytest.raises(IndexError):
        R.MeshBank.from_arrays({"bed": (v, f)}, "cpu", shell=shell)
    # table bounding boxes are kept as given (diff_render.py:106-115 scales by the TABLE's box, not the vertices' own)
    b = R.MeshBank.from_arrays({"bed": (v + 1.0, f, [0, 0, 0], [2, 3, 4])}, "cpu")
    assert b.models["bed"]["bbox_max"].tolist() == [2.0, 3.0, 4.0] and b.models["bed"]["bbox_min"].tolist() == [0.0, 0.0, 0.0]
