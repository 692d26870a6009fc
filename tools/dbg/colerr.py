import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from datum_amd import capi
from oracle import oracle as o
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = o.EXAMPLE
_, h0 = o.seed(N, 1000)
with capi.Ocean(N, 1) as oc:
    oc.set_cascade(0, p["wavescale"], p["choppiness"]); oc.upload_state(0, h0)
    oc.update(np.float32(1/60)); oc.displace(); a = oc.read_maps(0); oc.displace(); b = oc.read_maps(0)
ph = np.zeros((N, N), np.float32); o.update(ph, p["wavescale"], np.float32(1/60))
w = o.displace(h0, ph, p["wavescale"], p["choppiness"], w=o.weights(N, reduced=True))
print("run1 vs run2 differing texels:", int((a != b).any(axis=-1).sum()))
for layer in (0, 1):
    for comp in range(4):
        e = np.abs(a[layer, ..., comp] - w[layer, ..., comp])
        bad = e > 1e-4
        print(f"layer {layer} comp {comp}: max err {e.max():.3e} bad {int(bad.sum())}", "rows:", np.unique(np.argwhere(bad)[:, 0])[:12], "cols:", np.unique(np.argwhere(bad)[:, 1])[:20])
