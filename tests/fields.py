"""Deterministic test fields. `smooth_field` uses only integer arithmetic and exact float
conversions, so its bytes are identical on every machine (golden vectors depend on that)."""
import numpy as np

_M64 = (1 << 64) - 1


def _splitmix_array(n, seed):
    idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
    z = idx
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def smooth_field(shape_zyx, seed=7, dtype=np.float32, passes=3):
    """Integer noise, box-smoothed `passes` times along each axis by exact integer sums, scaled
    by a power of two: every value is exactly representable in float32."""
    n = int(np.prod(shape_zyx))
    with np.errstate(over="ignore"):
        r = _splitmix_array(n, seed)
    v = ((r >> np.uint64(48)).astype(np.int64) - 32768).reshape(shape_zyx)
    for _ in range(passes):
        for ax in range(3):
            v = v + np.roll(v, 1, axis=ax) + np.roll(v, -1, axis=ax)
    # |v| < 2^15 * 27^passes < 2^15 * 2^15 ; keep 20 significant bits so float32 is exact
    v = v >> 10
    return (v.astype(np.float64) / 64.0).astype(dtype)


def ramp_field(shape_zyx, dtype=np.float32):
    z, y, x = np.meshgrid(*[np.arange(s, dtype=np.int64) for s in shape_zyx], indexing="ij")
    return ((x * 3 + y * 5 - z * 7) % 97).astype(dtype)
