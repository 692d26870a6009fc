"""Diagnostic: ocean.gen's vertices from the library in use; first call saves them, second call (another library) compares."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from datum_amd import capi, host_api
N, size, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
p = host_api.OceanParams(N, **host_api.EXAMPLE_TUNABLES)
p.seed_ocean(1000)
with capi.Ocean(N, 1) as oc:
    oc.set_cascade(0, 22.0, 1.35)
    oc.upload_state(0, p.height)
    oc.update(np.float32(1 / 60))
    oc.displace()
    verts = torch.zeros(size * size * 12, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    oc.gen(0, p.oceanset(), size, size, verts.data_ptr())
    oc.sync()
    v = verts.cpu().numpy().reshape(size, size, 12)
if os.path.exists(path):
    w = np.load(path)
    d = (v != w)
    print(f"N={N} mesh {size}: differing floats {int(d.sum())} of {v.size}")
    if d.any():
        ys, xs, cs = np.nonzero(d)
        print("   rows", np.unique(ys)[:24], "n rows", len(np.unique(ys)), " cols", np.unique(xs)[:40], "n cols", len(np.unique(xs)), " components", np.unique(cs))
        for k in range(min(6, len(ys))):
            print("   ", ys[k], xs[k], cs[k], "this", v[ys[k], xs[k]], "saved", w[ys[k], xs[k]])
else:
    np.save(path, v)
    print("saved", path)
