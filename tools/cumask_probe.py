"""Run ON the GPU box: which compute units a CU-masked stream covers.  For a few masks, launch 1024 sleeping workgroups on a stream
made by hipExtStreamCreateWithCUMask and list the distinct (XCC, SE, CU) they ran on.
usage: python tools/cumask_probe.py [hexmask ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from datum_amd.farm import cu_masked_stream          # noqa: E402

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "datum_amd", "lib", "libdatum_farm_standin.so"))
lib.datum_farm_standin_where.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)

masks = [int(m, 16) for m in sys.argv[1:]] or [0xFF, 0xFF00, 0x1, 0x101, 0xFFFF, (1 << 256) - 1 - 0xFFFF, 0xFFFFFFFF]
W = 4096

for mask in masks:
    st = cu_masked_stream(dev, mask)
    out = torch.zeros(2 * W, dtype=torch.int32, device=dev)
    rc = lib.datum_farm_standin_where(out.data_ptr(), W, st.cuda_stream)
    assert rc == 0, rc
    st.synchronize()
    o = out.cpu().numpy().astype("uint32").reshape(W, 2)
    cus = sorted({(int(x) & 15, (int(h) >> 13) & 7, (int(h) >> 12) & 1, (int(h) >> 8) & 15) for x, h in o})
    perxcc = {}
    for x, se, sh, cu in cus:
        perxcc.setdefault(x, []).append(f"{se}.{sh}.{cu}")
    print(f"mask {mask:#x} ({bin(mask).count('1')} bits): {len(cus)} compute units")
    for x in sorted(perxcc):
        print(f"   xcc {x}: {' '.join(perxcc[x])}")
