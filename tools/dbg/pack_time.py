"""Run ON the GPU box: the payload pack kernel alone (datum_ocean_pack_displacement), per format and shape: us per call and GB/s on the bytes it reads
(the 16-byte halves of the 24-byte texels it needs, or the whole map block) and writes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datum_amd import capi, farm, host_api          # noqa: E402

torch.zeros(1, device="cuda:0")
for N, C in ((1024, 4), (2048, 1), (512, 1), (1024, 16), (4096, 1), (256, 1)):
    oc = capi.Ocean(N, C)
    for c in range(C):
        p = host_api.OceanParams(N, **dict(host_api.EXAMPLE_TUNABLES, wavescale=farm.grid_wavescale(c, 4)))
        p.seed_ocean(farm.grid_seed(c))
        oc.set_cascade(c, farm.grid_wavescale(c, 4), 1.35)
        oc.upload_state(c, p.height)
        del p
    st = torch.cuda.Stream()
    oc.set_stream(st.cuda_stream)
    oc.update(1 / 60)
    oc.displace()
    for fmt in ("xyz32", "xyz16", "maps"):
        code, dtype, _ = farm.PAYLOADS[fmt]
        nbytes = oc.payload_bytes(code)
        buf = torch.empty(farm.payload_numel(N, C, fmt), dtype=dtype, device="cuda:0")
        for _ in range(20):
            oc.pack_displacement(code, buf.data_ptr(), nbytes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(200):
            oc.pack_displacement(code, buf.data_ptr(), nbytes)
        e1.record(st)
        e1.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        pts = N * N * C
        read = pts * (24 if fmt == "maps" else 16)
        print(f"{N:5d}^2 x {C:2d}  {fmt:6s} {us:8.2f} us per pack   {(read + nbytes) / us / 1e3:7.0f} GB/s on {read / 1e6:.0f} MB read + {nbytes / 1e6:.0f} MB written")
    img = torch.empty(2 * N * N * 4, dtype=torch.float32, device="cuda:0")
    for _ in range(20):
        oc.export_maps(0, img.data_ptr(), img.numel() * 4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(200):
        oc.export_maps(0, img.data_ptr(), img.numel() * 4)
    e1.record(st)
    e1.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"{N:5d}^2 x  1  export_maps (one cascade -> RGBA32F x 2) {us:8.2f} us   {N * N * 56 / us / 1e3:7.0f} GB/s on {N * N * 24 / 1e6:.0f} MB read + {N * N * 32 / 1e6:.0f} MB written")
    oc.set_stream(None)
    del oc
