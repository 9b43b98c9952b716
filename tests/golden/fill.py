"""Closed-form deterministic tensor fill shared by the golden generator and the tests
(so large-configuration fixtures can be checksums only instead of 20 MB of weights)."""
import zlib

import numpy as np


def det_fill(shape, key, scale=1.0, offset=0.0):
    """Pseudo-random-looking but closed-form values in [-scale, scale] + offset, float32."""
    n = int(np.prod(shape)) if len(shape) else 1
    phase = (zlib.crc32(key.encode()) % 100003) * 0.001
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + phase) * 43758.5453
    v = (v - np.floor(v)) * 2.0 - 1.0
    return (v * scale + offset).astype(np.float32).reshape(shape)
