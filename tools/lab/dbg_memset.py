import sys, importlib, torch
sys.path.insert(0, '/root/repo')
L = importlib.import_module('3d_sln_amd._lib')
dev = "cuda"
dummy = torch.zeros(64, device=dev); di = torch.zeros(64, dtype=torch.int32, device=dev)
for V in (1167, 1168, 389 * 3, 4, 5, 1001):
    for off in (0, 1, 3):
        base = torch.full((3 * V + 64,), 7.0, device=dev)
        buf = base[off:off + 3 * V]
        def call():
            L.check(L.lib().sln_project_faces_backward(L.ptr(dummy), L.ptr(di), L.ptr(dummy), L.ptr(dummy), L.ptr(dummy), 1, V, 0, 512.0, 1e-9,
                                                       L.ptr(dummy), L.ptr(buf), L.current_stream_ptr()), "memset")
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): call()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        eager_ok = bool((buf == 0).all()) and bool((base[off + 3 * V:] == 7).all()) and bool((base[:off] == 7).all())
        base.fill_(7.0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): call()
        base.fill_(7.0); g.replay(); torch.cuda.synchronize()
        nz = (buf != 0).nonzero().flatten().tolist()
        print("V %5d bytes %6d offset %d: eager ok %s | graph: %d of %d not zeroed %s, neighbours intact %s" % (
            V, 12 * V, off, eager_ok, len(nz), 3 * V, nz[:4] + nz[-2:], bool((base[off + 3 * V:] == 7).all()) and bool((base[:off] == 7).all())))
