"""Device scene-graph builder (csrc/graph_build.hip, host/suncg_dataset.py) against the fixture produced by the
reference's own SuncgDataset / suncg_collate_fn and against the oracle restatement.
Integer outputs and the normalised boxes are compared bit-exactly."""
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import load_golden, pkg

pytestmark = pytest.mark.gpu

from oracle import graph_build_ref as G     # noqa: E402


def _fixture():
    g = load_golden("graph_build")
    meta = json.loads(bytes(g["meta"]).decode())
    rooms, names, sd, sd30 = G.synth_rooms(meta["n_rooms"], meta["seed"])
    return g, meta, rooms, names, sd, sd30


def _dataset(rooms, names, sd, sd30, use30):
    D = pkg("host.suncg_dataset")
    return D, D.SuncgDataset.from_tables(rooms, names, sd, sd30, use_attr_30=use30, room_ids=[100 + i for i in range(len(rooms))])


@pytest.mark.parametrize("tag,use30", [("a", False), ("b", True)])
def test_getitem_matches_reference_fixture(tag, use30):
    g, meta, rooms, names, sd, sd30 = _fixture()
    D, ds = _dataset(rooms, names, sd, sd30, use30)
    assert len(ds) == len(rooms) and ds.total_objects() == sum(len(r["objs"]) for r in rooms)
    assert ds.vocab["pred_idx_to_name"] == G.PRED_NAMES and ds.vocab["object_idx_to_name"] == names
    batch = []
    for idx in range(len(ds)):
        random.seed(meta["getitem_seed_base"] + idx)
        rid, o, b, t, a, at = ds[idx]
        key = "%s_room%02d_" % (tag, idx)
        assert rid == 100 + idx
        for got, k in ((o, "objs"), (b, "boxes"), (t, "triples"), (a, "angles"), (at, "attrs")):
            want = g[key + k]
            assert got.numpy().dtype == want.dtype and np.array_equal(got.numpy(), want), (idx, k)
        batch.append((rid, o, b, t, a, at))
    col = D.suncg_collate_fn(batch)
    for k, v in zip(("ids", "objs", "boxes", "triples", "angles", "attrs", "obj_to_img", "triple_to_img"), col):
        assert np.array_equal(v.numpy(), g["%s_collate_%s" % (tag, k)]), k


@pytest.mark.parametrize("tag,use30", [("a", False), ("b", True)])
def test_build_batch_equals_reference_collate(tag, use30):
    g, meta, rooms, names, sd, sd30 = _fixture()
    D, ds = _dataset(rooms, names, sd, sd30, use30)
    parts = []
    for idx in range(len(ds)):
        random.seed(meta["getitem_seed_base"] + idx)
        parts.append(ds.draw([idx]))
    draws = tuple(np.concatenate([p[k] for p in parts]) for k in range(3))
    out = ds.build_batch(list(range(len(ds))), draws=draws)
    for k, v in zip(("ids", "objs", "boxes", "triples", "angles", "attrs", "obj_to_img", "triple_to_img"), out):
        want = g["%s_collate_%s" % (tag, k)]
        assert v.is_cuda and v.cpu().numpy().dtype == want.dtype and np.array_equal(v.cpu().numpy(), want), k
    # a permuted, repeating subset: rooms are independent
    sel = [5, 0, 17, 5, 33]
    sub = ds.build_batch(sel, draws=tuple(np.concatenate([parts[i][k] for i in sel]) for k in range(3)))
    batch = [(100 + i,) + tuple(g["%s_room%02d_%s" % (tag, i, k)] for k in ("objs", "boxes", "triples", "angles", "attrs")) for i in sel]
    want = G.collate(batch)
    for v, w in zip(sub, want):
        assert np.array_equal(v.cpu().numpy(), w)


def test_json_constructor_reads_the_reference_layout(tmp_path):
    g, meta, rooms, names, sd, sd30 = _fixture()
    os.makedirs(tmp_path / "metadata")
    data = {}
    for r, room in enumerate(rooms):
        data[str(100 + r)] = dict(valid_objects=[dict(type=names[c], new_bbox=[[float(x) for x in b[:3]], [float(x) for x in b[3:]]], rotation=int(a))
                                                 for c, b, a in zip(room["objs"], room["boxes"], room["rot"])], bbox=[float(x) for x in room["bbox"]])
    json.dump(data, open(tmp_path / "rooms.json", "w")); json.dump(names[1:], open(tmp_path / "metadata" / "valid_types.json", "w"))
    json.dump(sd, open(tmp_path / "metadata" / "size_info_many.json", "w")); json.dump(sd30, open(tmp_path / "metadata" / "30_size_info_many.json", "w"))
    D = pkg("host.suncg_dataset")
    ds = D.SuncgDataset(str(tmp_path / "rooms.json"), True, metadata_dir=str(tmp_path / "metadata"))
    assert ds.return_room_ids() == [100 + i for i in range(len(rooms))]
    random.seed(meta["getitem_seed_base"] + 7)
    rid, o, b, t, a, at = ds.get_by_room_id(107)
    assert rid == 107 and np.array_equal(t.numpy(), g["a_room07_triples"]) and np.array_equal(b.numpy(), g["a_room07_boxes"])
    assert np.array_equal(at.numpy(), g["a_room07_attrs"])


def test_device_draws_large_batch_properties_and_oracle():
    rooms, names, sd, sd30 = G.synth_rooms(700, seed=21, max_objs=40)
    D = pkg("host.suncg_dataset")
    ds = D.SuncgDataset.from_tables(rooms, names, sd, sd30)
    gen = torch.Generator(device="cuda").manual_seed(3)
    idx = torch.randint(0, len(rooms), (2048,), generator=torch.Generator().manual_seed(1))
    ids, objs, boxes, triples, angles, attrs, o2i, t2i = (t.cpu().numpy() for t in ds.build_batch(idx, generator=gen))
    again = ds.build_batch(idx, generator=torch.Generator(device="cuda").manual_seed(3))
    assert np.array_equal(again[3].cpu().numpy(), triples) and np.array_equal(again[5].cpu().numpy(), attrs)
    # host-side indices are planned on the host from per-room counts taken once (no read-back per batch); indices that live on
    # the device go through sln_graph_plan + a read-back: same batch either way
    on_dev = ds.build_batch(idx.cuda(), generator=torch.Generator(device="cuda").manual_seed(3))
    for a, b in zip(on_dev, (ids, objs, boxes, triples, angles, attrs, o2i, t2i)):
        assert np.array_equal(a.cpu().numpy(), b)
    n = np.array([len(rooms[i]["objs"]) for i in idx.tolist()])
    row0 = np.concatenate([[0], np.cumsum(n + 1)])
    assert objs.shape[0] == row0[-1] and np.array_equal(ids, idx.numpy())
    assert np.array_equal(o2i, np.repeat(np.arange(len(n)), n + 1)) and np.all(np.diff(t2i) >= 0)
    assert np.all(objs[row0[1:] - 1] == 0) and np.all(attrs[row0[1:] - 1] == 0) and attrs.min() >= 0 and attrs.max() <= 4
    assert np.all((triples[:, 0] >= row0[t2i]) & (triples[:, 0] < row0[t2i + 1]) & (triples[:, 2] >= row0[t2i]) & (triples[:, 2] < row0[t2i + 1]))
    assert np.all(triples[:, 0] != triples[:, 2])
    table = G.RoomTable(rooms, names, sd, sd30)
    for b in list(range(48)) + [511, 2047]:                         # full oracle comparison, draws recovered from the output
        room = rooms[int(idx[b])]
        k = len(room["objs"])
        tr = triples[t2i == b].copy(); tr[:, 0] -= row0[b]; tr[:, 2] -= row0[b]
        n_on = tr.shape[0] - 2 * k
        drawn = tr[n_on:n_on + k]
        other = np.where(drawn[:, 0] == np.arange(k), drawn[:, 2], drawn[:, 0]); swap = drawn[:, 0] == np.arange(k)
        at = attrs[row0[b]:row0[b + 1]]
        u1 = np.where(at[:k] == 0, 0.9, 0.1); u2 = np.where((at[:k] == 1) | (at[:k] == 2), 0.9, 0.1)
        o, bx, t, a, att = G.build_room(room, table, (other, swap, u1, u2))
        assert np.array_equal(t, tr) and np.array_equal(bx, boxes[row0[b]:row0[b + 1]]) and np.array_equal(a, angles[row0[b]:row0[b + 1]])
        assert np.array_equal(o, objs[row0[b]:row0[b + 1]])
        known = np.array([table.has_size(c) for c in room["objs"]])
        assert np.all(att[:k][at[:k] != 0] == at[:k][at[:k] != 0]) and np.all(at[:k][~known] == 0)
    # the builder feeds the training step directly
    M = pkg("host.Sg2ScVAE_model")
    from oracle import vae_ref
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, num_objs=len(names))
    model = M.Sg2ScVAEModel(**cfg.model_kwargs()); model.load_state_dict(vae_ref.init_state(cfg, seed=0)); model = model.cuda().train()
    out = ds.build_batch(idx[:64], generator=gen)
    losses = model.train_step(out[1], out[3], out[2], out[4], out[5], kl_weight=0.1, lr=1e-4, use_graph=False)
    assert torch.isfinite(losses).all() and float(losses[3]) > 0


def test_errors_mirror_the_reference():
    rooms, names, sd, sd30 = G.synth_rooms(4, seed=2)
    rooms[2] = dict(objs=rooms[2]["objs"][:1], boxes=rooms[2]["boxes"][:1], rot=rooms[2]["rot"][:1], bbox=rooms[2]["bbox"])
    D = pkg("host.suncg_dataset")
    ds = D.SuncgDataset.from_tables(rooms, names, sd, sd30)
    with pytest.raises(IndexError):
        ds[2]                                   # random.choice([]) in the reference
    with pytest.raises(IndexError):
        ds.build_batch([0, 2])
    with pytest.raises(IndexError):
        ds.build_batch([0, 9])
    with pytest.raises(ValueError):
        ds.build_batch([0], draws=(np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(1, np.uint8)))


def test_train_loop_reads_rooms_through_the_device_builder(tmp_path, capsys):
    """host/train.py --suncg_train_dir: json rooms -> device scene-graph builder -> fused train step (eager: sizes vary)."""
    g, meta, rooms, names, sd, sd30 = _fixture()
    os.makedirs(tmp_path / "metadata")
    data = {str(100 + r): dict(valid_objects=[dict(type=names[c], new_bbox=[[float(x) for x in b[:3]], [float(x) for x in b[3:]]], rotation=int(a))
                                              for c, b, a in zip(room["objs"], room["boxes"], room["rot"])], bbox=[float(x) for x in room["bbox"]])
            for r, room in enumerate(rooms)}
    json.dump(data, open(tmp_path / "rooms.json", "w")); json.dump(names[1:], open(tmp_path / "metadata" / "valid_types.json", "w"))
    json.dump(sd, open(tmp_path / "metadata" / "size_info_many.json", "w")); json.dump(sd30, open(tmp_path / "metadata" / "30_size_info_many.json", "w"))
    T = pkg("host.train")
    T.main(["--suncg_train_dir", str(tmp_path / "rooms.json"), "--metadata_dir", str(tmp_path / "metadata"), "--batch_size", "16",
            "--num_iterations", "12", "--print_every", "4", "--checkpoint_every", "1000", "--embedding_dim", "32",
            "--gconv_num_layers", "2", "--output_dir", str(tmp_path / "ck")])
    logs = capsys.readouterr().out.splitlines()
    totals = [float(l.split(":")[1]) for l in logs if "[total_loss]" in l]
    assert any("Training dataset has 48 scenes" in l for l in logs)
    assert len(totals) == 3 and all(np.isfinite(totals)) and totals[-1] < totals[0]


def test_train_script_checkpoints_and_resumes(tmp_path, capsys):
    """train.py:16-31,92-98: the checkpoint holds the model, the optimizer state (torch.optim.Adam's layout) and t; a run with
    --restore_from_checkpoint continues from there (the reference restores '<name>_with_model.pt', it saves 'latest_<name>_…')."""
    T = pkg("host.train")
    common = ["--batch_size", "8", "--objs_per_graph", "6", "--triples_per_graph", "9", "--embedding_dim", "32", "--gconv_num_layers", "2",
              "--print_every", "2", "--output_dir", str(tmp_path), "--checkpoint_name", "run"]
    T.main(common + ["--num_iterations", "4", "--checkpoint_every", "4"])
    saved = tmp_path / "latest_run_with_model.pt"
    assert saved.is_file()
    ck = torch.load(saved, map_location="cpu", weights_only=False)
    assert ck["counters"]["t"] == 4 and float(ck["optim_state"]["state"][0]["step"]) == 4.0
    assert set(ck["optim_state"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and ck["losses_ts"] == [2, 4]
    os.rename(saved, tmp_path / "run_with_model.pt")
    capsys.readouterr()
    T.main(common + ["--num_iterations", "6", "--checkpoint_every", "6", "--restore_from_checkpoint", "1"])
    logs = capsys.readouterr().out.splitlines()
    assert any("Restoring from checkpoint" in l for l in logs)
    assert [l for l in logs if l.startswith("On batch")] == ["On batch 6 out of 6"]        # t resumed at 4: only iterations 5 and 6 ran
    ck2 = torch.load(tmp_path / "latest_run_with_model.pt", map_location="cpu", weights_only=False)
    assert ck2["counters"]["t"] == 6 and float(ck2["optim_state"]["state"][0]["step"]) == 6.0 and ck2["losses_ts"] == [2, 4, 6]
    moved = max(float((ck2["model_state"][k] - ck["model_state"][k]).abs().max()) for k in ck["model_state"] if k.endswith(".weight"))
    assert 0 < moved < 1e-2


def test_device_draw_kernel_has_the_reference_distributions():
    """sln_graph_draw: the partner is uniform over the n - 1 OTHER objects of the room (random.choice, suncg_dataset.py:189-196), the
    order flips with probability 1/2, the size attribute is 'none' with probability 1/2 (always for classes without statistics)
    and height / volume with 1/4 each (:236-282); same generator state -> same draws, next call -> new draws."""
    rooms, names, sd, sd30 = G.synth_rooms(300, seed=5, max_objs=12)
    D = pkg("host.suncg_dataset")
    ds = D.SuncgDataset.from_tables(rooms, names, sd, sd30)
    idx = torch.randint(0, len(rooms), (8192,), generator=torch.Generator().manual_seed(2))
    n = np.array([len(rooms[i]["objs"]) for i in idx.tolist()])
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(n + 1)]).astype(np.int32)).cuda()
    idx_t = idx.int().cuda()
    O = int(off[-1])
    gen = torch.Generator(device="cuda").manual_seed(11)
    other, swap, mode = (t.cpu().numpy() for t in ds.device_draws(idx_t, off, O, gen))
    o2, s2, m2 = (t.cpu().numpy() for t in ds.device_draws(idx_t, off, O, torch.Generator(device="cuda").manual_seed(11)))
    assert np.array_equal(other, o2) and np.array_equal(swap, s2) and np.array_equal(mode, m2)
    o3 = ds.device_draws(idx_t, off, O, gen)[0].cpu().numpy()
    assert np.mean(o3 != other) > 0.3                                   # the generator moved on
    cur = np.concatenate([np.arange(k) for k in n]); nn = np.repeat(n, n)
    assert other.shape[0] == cur.shape[0] and np.all(other != cur) and np.all((other >= 0) & (other < nn))
    N = other.shape[0]
    assert abs(swap.mean() - 0.5) < 4 * 0.5 / np.sqrt(N)
    # uniform partner: position of `other` among the n - 1 others, pooled over rooms of one size
    for k in (3, 6, 10):
        sel = nn == k
        pos = np.where(other[sel] > cur[sel], other[sel] - 1, other[sel])
        cnt = np.bincount(pos, minlength=k - 1)
        exp = sel.sum() / (k - 1)
        assert np.all(np.abs(cnt - exp) < 5 * np.sqrt(exp)), (k, cnt, exp)
    cls = np.concatenate([np.asarray(rooms[i]["objs"]) for i in idx.tolist()])
    known = ds._has_host[cls] != 0
    assert np.all(mode[~known] == 0)
    mk = mode[known]
    for v, p in ((0, 0.5), (1, 0.25), (2, 0.25)):
        assert abs(np.mean(mk == v) - p) < 5 * np.sqrt(p * (1 - p) / mk.shape[0]), (v, np.mean(mk == v))
