import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import test_refine_golden_gpu as T
from conftest import load_golden, pkg
R = pkg("host.refine")
g = load_golden("refine_loop")
model, cfg = T._loop_model(g); bank = T._bank(g); rm = T._loop_rooms(g, [0,1]); it = 4
rb = R.RefineBatch(model, rm, bank=bank, image_size=96, iters=it)
for i in (0,1):
    a,n = rb.row0[i], rb.rows[i]; rb.z[a:a+n] = torch.from_numpy(g["room%d:z0"%i]).cuda()
rel=lambda a,b: float(np.abs(np.asarray(a,np.float64)-np.asarray(b,np.float64)).max()/max(np.abs(np.asarray(b)).max(),1e-30))
zprev=[g["room%d:z0"%i] for i in (0,1)]
for k in range(it):
    rb.run(1)
    for i in (0,1):
        a,n = rb.row0[i], rb.rows[i]; p="room%d:"%i
        z=rb.z[a:a+n].cpu().numpy()
        print(k,i,"loss %.1e boxes %.1e idx %.1e z %.1e zstep %.1e"%(rel(float(rb.losses[k,i]),g[p+"loss"][k]), rel(rb.boxes[a:a+n].cpu().numpy(),g[p+"boxes"][k]), rel(rb.idx[a:a+n].cpu().numpy(),g[p+"idx"][k]), rel(z,g[p+"z"][k]), rel(z-zprev[i], g[p+"z"][k]-zprev[i])))
        zprev[i]=g[p+"z"][k]
for i in (0,1):
    p="room%d:"%i
    for key in [k for k in g.files if k.startswith(p+"param:")]:
        name=key[len(p)+6:]; t=dict(model.named_parameters())[name]; off=(t.data_ptr()-model.flat_params.data_ptr())//4
        got=rb.params[i,off:off+t.numel()].reshape(t.shape).cpu().numpy(); p0=g["state:"+name]
        print(" ",i,name,"%.1e"%rel(got-p0, g[key][-1]-p0))
rb.close()
