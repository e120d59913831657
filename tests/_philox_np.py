"""Philox4x32-10 and the contract's counter convention (include/vbnn_philox.h) vectorised in NumPy -- test infrastructure: it gives
the tests the WORDS behind a window of normals (u1 and the angle of every pair), which the C oracle only turns into normals.
Pinned by the Random123 known-answer vectors (tests/golden/philox_kat.json) and against the oracle's vbo_philox_raw."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c = [np.asarray(v, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = c[0] * M0, c[2] * M1
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & np.uint64(0xFFFFFFFF), p1 >> np.uint64(32), p1 & np.uint64(0xFFFFFFFF)
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def window_words(rows, cols, seed, stream, layer, draw, row0):
    """The two Philox words (x0 -> radius, x1 -> angle) behind element (r, c) of a rows x cols window of normals: arrays of shape
    (rows, cols // 2) for the pairs (c = 2p, 2p + 1)."""
    assert cols % 4 == 0
    r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    q = np.arange(cols // 4, dtype=np.uint64)[None, :]
    w = philox4x32_10(q, r, draw, (layer << 8) | stream, seed & 0xFFFFFFFF, seed >> 32)
    x0 = np.stack([w[0], w[2]], axis=-1).reshape(rows, cols // 2)
    x1 = np.stack([w[1], w[3]], axis=-1).reshape(rows, cols // 2)
    return x0, x1


def ideal_box_muller(x0, x1):
    """float64 Box-Muller on the contract's reading of the words: (z0, z1, r)."""
    u1 = ((x0 >> 8).astype(np.float64) + 1.0) * 2.0 ** -24
    t = (x1 >> 8).astype(np.float64) * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2 * np.pi * t), r * np.sin(2 * np.pi * t), r
