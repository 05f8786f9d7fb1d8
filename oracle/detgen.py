"""Deterministic tensor generator shared by the golden-vector script and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Values come from an integer hash (splitmix64) of (name, flat index), mapped to an
Irwin-Hall(4) approximately-normal variate with exact float64 arithmetic, so the
same (name, shape, std) gives bit-identical float32 data on any machine and with
any numpy/torch version.  Neither side of a parity test depends on an RNG stream.
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def det_array(name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """float32 array of `shape`, ~N(mean, std^2), fully determined by `name`."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode()) * 0x100000001B3 & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(n, dtype=np.uint64) ^ seed)
    # four 16-bit fields -> Irwin-Hall(4): mean 2, variance 4/12
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.float64)
    u = (s / 65536.0 - 2.0) * np.sqrt(3.0)  # unit variance
    return (u * std + mean).astype(np.float32).reshape(shape)


def det_labels(name: str, n: int, num_classes: int) -> np.ndarray:
    seed = np.uint64(zlib.crc32(name.encode()))
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(n, dtype=np.uint64) ^ seed)
    return (h % np.uint64(num_classes)).astype(np.int64)
