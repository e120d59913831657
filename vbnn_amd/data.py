"""data.lua: the datasets and `create_minibatch` (data.lua:6-32), feeding the hot path from the host.

`getMnist(root)` reads the published IDX files (train-images-idx3-ubyte etc., optionally .gz) -- the torch `mnist`
rock the reference requires (data.lua:1) is un-vendored and downloads exactly those. This environment has no network
and no dataset, so `synthetic_digits` provides a stand-in of the same geometry (28 x 28 floats, 10 classes) for the
end-to-end training test; it is NOT MNIST and no accuracy quoted on it is comparable with the reference's.

A dataset is {'inputs': float32 [n, 28, 28] (normalised in place, utils.normalize, data.lua:25,30),
'targets': int64 [n] in 0..classes-1}. The reference's targets are 1-based (`targets[i] + 1`, data.lua:16); the
device criterion takes 0-based int32 (include/vbnn_hip.h vbnn_head_forward), so nothing is added here.
"""
import gzip
import os
import struct

import numpy as np

from . import utils as u


def _read_idx(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        zero, dtype_code, nd = struct.unpack(">HBB", f.read(4))
        if zero != 0 or dtype_code != 0x08:
            raise ValueError(f"{path}: not an unsigned-byte IDX file")
        shape = struct.unpack(f">{nd}I", f.read(4 * nd))
        a = np.frombuffer(f.read(), dtype=np.uint8)
    if a.size != int(np.prod(shape)):
        raise ValueError(f"{path}: {a.size} bytes of payload for shape {shape}")
    return a.reshape(shape)


def _find(root, stem):
    for name in (stem, stem + ".gz", stem.replace("-idx", ".idx"), stem.replace("-idx", ".idx") + ".gz"):
        p = os.path.join(root, name)
        if os.path.isfile(p):
            return p
    raise FileNotFoundError(f"{stem}[.gz] not found under {root}")


class Dataset(dict):
    def create_minibatch(self, index, batchSize, n, geometry=None):
        """Rows [index, min(index + batchSize, n)) as a [batchSize, 1, h, w] block (data.lua:9-20). `index` is 0-based.
        As in the reference a short last batch keeps its full size; the reference leaves those rows uninitialised,
        here they are zero (inputs) and class 0 (targets)."""
        hi = min(index + batchSize, n)
        x = self["inputs"]
        inputs = np.zeros((batchSize, 1) + tuple(x.shape[1:]), dtype=np.float32)
        targets = np.zeros(batchSize, dtype=np.int64)
        inputs[: hi - index, 0] = x[index:hi]
        targets[: hi - index] = self["targets"][index:hi]
        return inputs, targets


def _finish(images, labels):
    d = Dataset(inputs=np.ascontiguousarray(images, dtype=np.float32), targets=np.asarray(labels, dtype=np.int64))
    u.normalize(d["inputs"])                                     # data.lua:25,30: each split by its own statistics
    return d


def getMnist(root):
    train = _finish(_read_idx(_find(root, "train-images-idx3-ubyte")), _read_idx(_find(root, "train-labels-idx1-ubyte")))
    test = _finish(_read_idx(_find(root, "t10k-images-idx3-ubyte")), _read_idx(_find(root, "t10k-labels-idx1-ubyte")))
    return train, test


def synthetic_digits(n_train, n_test, classes=10, geometry=(28, 28), seed=3, noise=1.0):
    """Stand-in for MNIST: class c = a fixed random 28 x 28 prototype (blurred so neighbouring pixels correlate),
    a sample = prototype + noise * N(0,1). Same seed -> same data."""
    rs = np.random.RandomState(seed)
    protos = rs.randn(classes, *geometry).astype(np.float32)
    for _ in range(2):                                           # cheap separable blur
        protos = (protos + np.roll(protos, 1, 1) + np.roll(protos, -1, 1) + np.roll(protos, 1, 2) + np.roll(protos, -1, 2)) / 5

    def split(n):
        t = rs.randint(0, classes, n)
        x = protos[t] + noise * protos.std() * rs.randn(n, *geometry).astype(np.float32)
        return _finish(x, t)
    return split(n_train), split(n_test)


def getBacteriaFold(d, i, k):
    """Fold i (1-based) of k over a {'inputs' [n, f], 'targets' [n]} set loaded from the reference's
    `gutbacteria_shuffled_binary.torch` (data.lua:35-88, readable with vbnn_amd.t7file.load): the test rows are
    [(i-1) fold, (i-1) fold + test_size), fold = round(n / k), test_size = n - (k-1) fold."""
    x, t = np.asarray(d["inputs"], dtype=np.float32), np.asarray(d["targets"], dtype=np.int64)
    n = x.shape[0]
    fold = int(np.floor(n / k + 0.5))
    test_size = n - (k - 1) * fold
    lo = (i - 1) * fold
    mask = np.zeros(n, dtype=bool)
    mask[lo:min(lo + test_size, n)] = True
    return Dataset(inputs=x[~mask], targets=t[~mask]), Dataset(inputs=x[mask], targets=t[mask])
