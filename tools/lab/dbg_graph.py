import sys, importlib, torch
sys.path.insert(0, '/root/repo')
DR = importlib.import_module('3d_sln_amd.host.diff_render'); syn = importlib.import_module('3d_sln_amd.host.synthetic')
rooms = [syn.synthetic_room(300 + i, n_objects=6, target_faces=400) for i in range(3)]
pk = syn.pack_rooms(rooms, "cuda")
IS = 128
args = (pk["F"], pk["C"], pk["chan"], pk["dch"], pk["K"], pk["R"], pk["t"], IS, 0.001)
g = DR.SceneRenderGraph(pk["V"], *args)
gen = torch.Generator().manual_seed(4)
for trial in range(4):
    V = (pk["V"] + 0.01 * trial * torch.randn(pk["V"].shape, generator=gen).cuda()).detach()
    go = torch.randn(3, 70, IS, IS, generator=gen).cuda()
    image, dV = g(V, go)
    torch.cuda.synchronize()
    a = float(dV.abs().max()); ai = float(image.abs().max())
    Ve = V.clone().requires_grad_(True)
    ref = DR.scene_render_batch(Ve, *args); ref.backward(go)
    torch.cuda.synchronize()
    print(trial, "graph dV max %.3e image max %.3e | eager dV max %.3e | diff %.3e  nonfinite %d" % (a, ai, float(Ve.grad.abs().max()), float((dV - Ve.grad).abs().max()), int((~torch.isfinite(dV)).sum())))
    bad = (dV - Ve.grad).abs() > 1.0
    if bad.any():
        idx = bad.nonzero()[:5].tolist(); print("  bad idx", idx, [float(dV[tuple(i)]) for i in idx], [float(Ve.grad[tuple(i)]) for i in idx], "V rows", pk["V"].shape)
