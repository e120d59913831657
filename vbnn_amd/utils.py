"""utils.lua, function for function (host-side helpers around the hot path; torch tensors or numpy arrays).

get_accuracy (utils.lua:11-27), normalize (:29-35), select_data (:37-47), isnan (:64), norm_pdf (:66-71),
safe_save (:73-80, through vbnn_amd.t7file), shallow_copy (:82-88), shuffle (:90-94), file_exists (:96-99).
`num_grad` (:49-62) is the central difference the tests use for the KL gradients.
"""
import math
import os

import numpy as np

from . import t7file


def get_accuracy(outputs, targets):
    """Percentage of rows whose arg-max equals the target (utils.lua:11-27). `targets` are class indices in the
    SAME base as the arg-max positions (the reference: 1-based both; here: 0-based both). A 1-D `outputs` is one
    row scored against every target, as the reference does. Ties: first maximum, like torch's max."""
    out = np.asarray(outputs.detach().cpu() if hasattr(outputs, "detach") else outputs)
    tgt = np.asarray(targets.detach().cpu() if hasattr(targets, "detach") else targets).astype(np.int64).reshape(-1)
    if tgt.size == 0:
        return float("nan")                                      # 0/0 in Lua
    idx = np.full(tgt.shape, int(out.argmax())) if out.ndim == 1 else out.argmax(axis=1)[:tgt.size]
    return float((idx == tgt).sum()) / tgt.size * 100.0


def normalize(data):
    """In place: data <- (data - mean) / std with torch's unbiased std (utils.lua:29-35); returns (mean, std)."""
    if hasattr(data, "std") and hasattr(data, "add_"):           # torch tensor
        std, mean = float(data.std()), float(data.mean())
        data.add_(-mean).mul_(1.0 / std)
        return mean, std
    std, mean = float(data.std(ddof=1)), float(data.mean())
    data -= mean
    data *= 1.0 / std
    return mean, std


def select_data(trainData, indices):
    """Rows `indices` (0-based here) of a {'inputs','targets'} set (utils.lua:37-47)."""
    idx = np.asarray(list(indices), dtype=np.int64)
    return {"inputs": trainData["inputs"][idx], "targets": trainData["targets"][idx]}


def num_grad(to_check, func):
    """Central difference of `func()` with respect to a uniform shift of `to_check`, the reference's step rule
    eps = 2 sqrt(1e-12) (1 + ||x||) (utils.lua:49-62). `to_check` is modified in place and restored."""
    eps = 2.0 * math.sqrt(1e-12) * (1.0 + float(np.linalg.norm(np.asarray(to_check, dtype=np.float64))))
    to_check += eps
    f1 = np.array(func(), dtype=np.float64, copy=True)
    to_check -= 2 * eps
    f2 = np.array(func(), dtype=np.float64, copy=True)
    to_check += eps
    return (f1 - f2) / (2 * eps)


def isnan(x):
    return x != x


def norm_pdf(x, mu, sigma):
    """N(x; mu, sigma^2) elementwise (utils.lua:66-71)."""
    x, mu, sigma = (np.asarray(v, dtype=np.float64) for v in (x, mu, sigma))
    s2 = sigma ** 2
    return np.exp(-0.5 * (x - mu) ** 2 / s2) / np.sqrt(2 * math.pi * s2)


def safe_save(obj, folder, name):
    """torch.save with the previous file kept as `<name>.old` (utils.lua:73-80)."""
    os.makedirs(folder, exist_ok=True)
    filename = os.path.join(folder, name)
    if os.path.isfile(filename):
        os.replace(filename, filename + ".old")
    t7file.save(filename, obj)
    return filename


def shallow_copy(t):
    return dict(t)


def shuffle(seq, rng=None):
    """A random permutation of `seq` (utils.lua:90-94: torch.randperm over the batch start indices)."""
    rng = rng or np.random
    seq = list(seq)
    return [seq[i] for i in rng.permutation(len(seq))]


def file_exists(name):
    return os.path.isfile(name)
