"""ocean.gen test cases shared by the CPU pins and the GPU parity tests, and an independent float64 restatement of
data/ocean.gen.comp:67-137 (vectorised numpy, written from the shader, NOT from oracle/ocean_oracle.cpp).

Every `qi * ...` term of gen.comp:97-105 (the horizontal Gerstner offset, normal.z, tangent.xy) is multiplied by a
non-zero swellsteepness in every case but "example"; the cameras cover a pitched view, rays above the horizon
(costheta <= 0 -> dist = 1e6, gen.comp:89), a rolled view, a camera high enough for `margin` (gen.comp:79) to be
close to 1, a sea level that is not z = 0 (plane.w != 0) and swell directions off the default.
"""

import numpy as np

# name -> (camera position, target, up, overrides of oracle.EXAMPLE)
CASES = {
    # examples/ocean/ocean.cpp:63 with the example's zero steepness: the round-1/2 case
    "example": dict(position=(0, 0, 8), target=(1, 0, 8), up=(0, 0, 1), params={}),
    # pitched 30 degrees down, steep swell
    "pitched_steep": dict(position=(3, -2, 8), target=(3 + 0.8660254, -2, 8 - 0.5), up=(0, 0, 1),
                          params=dict(swellsteepness=0.8, swellamplitude=0.8, swelllength=40.0)),
    # looking 10 degrees above the horizon and a little sideways: about two thirds of the rays miss the plane
    "above_horizon": dict(position=(0, 0, 8), target=(0.9659258, 0.2, 8 + 0.1736482), up=(0, 0, 1),
                          params=dict(swellsteepness=0.3, swellamplitude=2.0, swelllength=90.0, swelldirection=(-0.6, 0.8))),
    # rolled 20 degrees about the view axis, pitched 15 degrees down
    "rolled": dict(position=(-5, 7, 12), target=(-5 + 0.9330127, 7 + 0.25, 12 - 0.2588190), up=(0, 0.3420201, 0.9396926),
                   params=dict(swellsteepness=0.8, swellamplitude=1.3, swelllength=60.0, swelldirection=(0.0, 1.0))),
    # 200 m up, 45 degrees down: margin = 1 + sqrt((2a + 0.5) / h) = 1.1
    "high": dict(position=(30, -50, 200), target=(30 + 0.7071068, -50, 200 - 0.7071068), up=(0, 0, 1),
                 params=dict(swellsteepness=0.3, swellamplitude=0.8, swelllength=40.0, swelldirection=(0.6, -0.8))),
    # sea level at z = 1.5 (plane.w = -1.5), camera 6.5 m above it
    "plane_w": dict(position=(0, 0, 8), target=(0.9396926, 0.1, 8 - 0.3420201), up=(0, 0, 1),
                    params=dict(swellsteepness=0.8, swellamplitude=0.5, swelllength=25.0, plane=(0.0, 0.0, 1.0, -1.5))),
}

FOV = 60.0 * np.pi / 180.0          # examples/ocean/ocean.h:17-18
ASPECT = 1920.0 / 1080.0


def oceanset(oracle, N, case, swellphase=0.7, wavescale=None):
    c = CASES[case]
    params = dict(c["params"])
    if wavescale is not None:
        params["wavescale"] = wavescale
    return oracle.oceanset(N, position=c["position"], target=c["target"], up=c["up"], params=params, swellphase=swellphase)


def gen_f64(s, maps, sizex, sizey):
    """data/ocean.gen.comp:67-137 in float64 from the 216-byte OceanSet header `s` (oracle.OceanSet or capi.OceanSet)
    and maps [2, N, N, 4].  Returns (vertices [sizey, sizex, 12], dist, costheta)."""
    f = lambda a: np.array(list(a), np.float64)
    invproj = f(s.invproj).reshape(4, 4)
    real, dual, plane = f(s.camera_real), f(s.camera_dual), f(s.plane)
    N = maps.shape[1]
    m = maps.astype(np.float64)

    def qmul(a, b):                                        # transform.inc:19-28, (w, x, y, z)
        aw, ax, ay, az = a
        bw, bx, by, bz = b
        return np.array([aw * bw - ax * bx - ay * by - az * bz,
                         aw * bx + ax * bw + ay * bz - az * by,
                         aw * by + ay * bw + az * bx - ax * bz,
                         aw * bz + az * bw + ax * by - ay * bx])

    camerapos = 2 * qmul(dual, real * np.array([1, -1, -1, -1]))[1:]                      # gen.comp:75
    cameraheight = plane[:3] @ camerapos + plane[3]
    margin = 1 + np.sqrt((2 * s.swellamplitude + 0.5) / cameraheight)

    xx, yy = np.meshgrid(np.arange(sizex, dtype=np.float64), np.arange(sizey, dtype=np.float64))
    u = (2 * xx / (sizex - 1) - 1) * margin
    v = (1 - 2 * yy / (sizey - 1)) * margin

    view = np.stack([invproj[r, 0] * u + invproj[r, 1] * v + invproj[r, 3] for r in range(3)], -1)      # row_major M * (u, v, 0, 1)
    view /= np.linalg.norm(view, axis=-1, keepdims=True)
    q = real[1:]
    t = 2 * np.cross(q, view)
    worlddir = view + real[0] * t + np.cross(q, t)                                         # transform.inc:32-37

    costheta = -(worlddir @ plane[:3])
    hit = costheta > 0
    dist = np.where(hit, cameraheight / np.where(hit, costheta, 1.0), 1e6)

    base = np.stack([camerapos[0] + dist * worlddir[..., 0], camerapos[1] + dist * worlddir[..., 1], np.full_like(dist, -plane[3])], -1)

    amplitude = float(s.swellamplitude)
    frequency = 2 * np.pi / s.swelllength
    d = f(s.swelldirection)
    qi = s.swellsteepness / (frequency * amplitude * 4 + 1e-6)
    phi = frequency * amplitude
    theta = frequency * (d[0] * base[..., 0] + d[1] * base[..., 1]) + s.swellphase
    ct, st = np.cos(theta), np.sin(theta)

    pos = base + np.stack([qi * amplitude * d[0] * ct, qi * amplitude * d[1] * ct, amplitude * st], -1)
    normal = np.stack([phi * d[0] * ct / 6, phi * d[1] * ct / 6, qi * phi * st], -1)
    tangent = np.stack([qi * phi * d[0] * d[0] * st, qi * phi * d[1] * d[0] * st, phi * d[0] * ct / 6], -1)

    unit = lambda a: a / np.linalg.norm(a, axis=-1, keepdims=True)
    tbn2 = unit(np.stack([-normal[..., 0], -normal[..., 1], 1 - normal[..., 2]], -1))
    tbn0 = unit(np.stack([1 - tangent[..., 0], -tangent[..., 1], tangent[..., 2]], -1))
    tbn1 = np.cross(tbn0, tbn2)

    def sample(layer):                                      # linear, REPEAT, lod 0: texel centres at (i + 0.5) / N
        fx = pos[..., 0] * s.scale * N - 0.5
        fy = pos[..., 1] * s.scale * N - 0.5
        flx, fly = np.floor(fx), np.floor(fy)
        ax, ay = (fx - flx)[..., None], (fy - fly)[..., None]
        i0, j0 = np.mod(flx, N).astype(np.int64), np.mod(fly, N).astype(np.int64)
        i1, j1 = (i0 + 1) % N, (j0 + 1) % N
        return ((1 - ax) * (1 - ay) * layer[j0, i0, :3] + ax * (1 - ay) * layer[j0, i1, :3]
                + (1 - ax) * ay * layer[j1, i0, :3] + ax * ay * layer[j1, i1, :3])

    disp, dn = sample(m[0]), sample(m[1])

    smoothing = np.clip(dist * s.smoothing - 0.35, 0.0, 1.0) ** 0.2
    tn = dn[..., 0:1] * tbn0 + dn[..., 1:2] * tbn1 + dn[..., 2:3] * tbn2              # tbn * displacementnormal
    tbn2 = unit(tn * (1 - smoothing[..., None]) + plane[:3] * smoothing[..., None])   # mix(a, b, t) = a (1 - t) + b t
    tbn0 = unit(np.array([1.0, 0.0, 0.0]) - tbn2[..., 0:1] * tbn2)

    out = np.empty((sizey, sizex, 12))
    out[..., 0] = pos[..., 0] - disp[..., 0]
    out[..., 1] = pos[..., 1] - disp[..., 1]
    out[..., 2] = pos[..., 2] + disp[..., 2]
    out[..., 3:5] = 0.1 * pos[..., :2]
    out[..., 5:8] = tbn2
    out[..., 8:11] = tbn0
    out[..., 11] = -1
    return out, dist, costheta


def well_conditioned(s, dist, costheta, eps=6e-8, ulps=4.0, budget=2e-5):
    """Vertices where `ulps` ulp of fp32 error in the view ray cannot move the swell phase theta (gen.comp:99) by more
    than `budget` radians: d(hit point) ~ dist / costheta * d(ray direction), theta = frequency * (direction . p)."""
    frequency = 2 * np.pi / s.swelllength
    with np.errstate(divide="ignore", invalid="ignore"):
        amplification = np.where(costheta > 0, dist / np.maximum(costheta, 1e-30), np.inf)
    return frequency * amplification * ulps * eps < budget


def compare(got, want):
    """The stated ocean.gen tolerance: position abs error relative to 1 + |p| per component, texcoords relative to
    1 + the largest |texcoord|, unit vectors absolute.  Returns the three maxima."""
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    pos = (np.abs(got[..., 0:3] - want[..., 0:3]) / (1 + np.abs(want[..., 0:3]))).max()
    tex = np.abs(got[..., 3:5] - want[..., 3:5]).max() / (1 + np.abs(want[..., 3:5]).max())
    frame = np.abs(got[..., 5:11] - want[..., 5:11]).max()
    return pos, tex, frame
