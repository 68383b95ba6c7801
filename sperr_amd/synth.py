"""Synthetic turbulence-like volumes: 48 Fourier modes with a k^-5/3 energy
spectrum, splitmix64(seed) parameters, evaluated in fp64 and narrowed to the requested dtype.

The bytes are generated ONCE per test / bench run and fed to both the HIP path and the oracle, so
libm/torch differences in sin() never enter a parity comparison.
"""
import math

import numpy as np

_M64 = (1 << 64) - 1


def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & _M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return state, z ^ (z >> 31)


def mode_table(seed=42, nmodes=48):
    """Returns (kvec[nmodes,3], phase[nmodes], amp[nmodes]) as float64 numpy arrays."""
    st = seed & _M64
    kv, ph, am = [], [], []

    def u01():
        nonlocal st
        st, r = _splitmix64(st)
        return (r >> 11) * (1.0 / (1 << 53))

    for _ in range(nmodes):
        k = 1.0 + 31.0 * u01()
        ct = 2.0 * u01() - 1.0
        az = 2.0 * math.pi * u01()
        st_ = math.sqrt(max(0.0, 1.0 - ct * ct))
        kv.append((k * st_ * math.cos(az), k * st_ * math.sin(az), k * ct))
        ph.append(2.0 * math.pi * u01())
        am.append(k ** (-5.0 / 6.0))
    return np.array(kv), np.array(ph), np.array(am)


def turbulence(shape_zyx, seed=42, dtype=np.float32, origin=(0, 0, 0), period=256.0):
    """numpy generator (small volumes). shape/origin are (z, y, x)."""
    kv, ph, am = mode_table(seed)
    dz, dy, dx = shape_zyx
    z = (np.arange(dz, dtype=np.float64) + origin[0]) * (2.0 * math.pi / period)
    y = (np.arange(dy, dtype=np.float64) + origin[1]) * (2.0 * math.pi / period)
    x = (np.arange(dx, dtype=np.float64) + origin[2]) * (2.0 * math.pi / period)
    out = np.zeros(shape_zyx, dtype=np.float64)
    for m in range(len(ph)):
        arg = (kv[m, 0] * x)[None, None, :] + (kv[m, 1] * y)[None, :, None] + \
              (kv[m, 2] * z)[:, None, None] + ph[m]
        out += am[m] * np.sin(arg)
    return out.astype(dtype)


def turbulence_torch(shape_zyx, device, seed=42, dtype=None, period=256.0, slab=32, origin_z=0):
    """torch generator for large volumes, evaluated on `device` slab by slab.  `origin_z`: the z
    index of the first plane (a volume too large for the device is generated in pieces)."""
    import torch

    dtype = dtype or torch.float32
    kv, ph, am = mode_table(seed)
    dz, dy, dx = shape_zyx
    out = torch.empty(shape_zyx, dtype=dtype, device=device)
    w = 2.0 * math.pi / period
    y = torch.arange(dy, dtype=torch.float64, device=device) * w
    x = torch.arange(dx, dtype=torch.float64, device=device) * w
    for z0 in range(0, dz, slab):
        z1 = min(dz, z0 + slab)
        z = torch.arange(origin_z + z0, origin_z + z1, dtype=torch.float64, device=device) * w
        acc = torch.zeros((z1 - z0, dy, dx), dtype=torch.float64, device=device)
        for m in range(len(ph)):
            arg = (kv[m, 0] * x)[None, None, :] + (kv[m, 1] * y)[None, :, None] + \
                  (kv[m, 2] * z)[:, None, None] + ph[m]
            acc += am[m] * torch.sin(arg)
        out[z0:z1] = acc.to(dtype)
    return out
