"""Host-side rows next to the hot path: Torch7 serialisation (utils.lua:73-80, mainviz.lua:11-15), utils.lua helpers,
data.lua minibatches. The byte-level vectors below are written out by hand from the format description in
vbnn_amd/t7file.py (torch7 File.lua, un-vendored): they pin the WRITER; nothing here can check it against a real
torch7 reader (no Lua in this environment)."""
import gzip
import os
import struct

import numpy as np
import pytest

from vbnn_amd import data, t7file, utils


def i32(*v):
    return struct.pack(f"<{len(v)}i", *v)


def i64(*v):
    return struct.pack(f"<{len(v)}q", *v)


def s(text):
    b = text.encode()
    return i32(len(b)) + b


def test_scalar_known_answers():
    assert t7file.dumps(None) == i32(0)
    assert t7file.dumps(1.5) == i32(1) + struct.pack("<d", 1.5)
    assert t7file.dumps(7) == i32(1) + struct.pack("<d", 7.0)               # Lua has one number type
    assert t7file.dumps(True) == i32(5, 1) and t7file.dumps(False) == i32(5, 0)
    assert t7file.dumps("vb") == i32(2) + s("vb")


def test_table_known_answer():
    want = i32(3, 1, 2) + i32(2) + s("S") + i32(1) + struct.pack("<d", 30.0) + i32(2) + s("type") + i32(2) + s("vb")
    assert t7file.dumps({"S": 30, "type": "vb"}) == want
    # list -> keys 1..n; nil values do not exist in a Lua table
    assert t7file.dumps([4.0]) == i32(3, 1, 1) + i32(1) + struct.pack("<d", 1.0) + i32(1) + struct.pack("<d", 4.0)
    assert t7file.dumps({"a": None}) == i32(3, 1, 0)


def test_float_tensor_known_answer():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    want = (i32(4, 1) + s("V 1") + s("torch.FloatTensor") + i32(2) + i64(2, 3) + i64(3, 1) + i64(1)
            + i32(4, 2) + s("V 1") + s("torch.FloatStorage") + i64(6) + a.tobytes())
    assert t7file.dumps(a) == want
    # a transposed view is written contiguous
    assert t7file.dumps(np.ascontiguousarray(a.T).T) == want


def test_round_trip_nested_and_shared():
    w = np.random.RandomState(0).randn(4, 5).astype(np.float32)
    lay = t7file.T7Object("nn.VBLinear", {"means": w, "lvars": np.log(np.full((4, 5), 1e-3, np.float32)), "W": 20,
                                          "bias": np.zeros(4, np.float32)})
    obj = {"model": t7file.T7Object("nn.Sequential", {"modules": [lay, t7file.T7Object("nn.ReLU", {"inplace": False})]}),
           "alias": w, "vb_indices": [2], "opt": {"hidden": [400, 400], "var_init": 1e-3, "msr_init": False, "name": "exp"},
           "long": np.array([1, 2, 3], np.int64), "dbl": np.array([[0.5]]), "byte": np.array([1, 255], np.uint8),
           "empty": np.zeros((0,), np.float32)}
    back = t7file.loads(t7file.dumps(obj))
    m = back["model"]
    assert m.className == "nn.Sequential" and m.fields["modules"][0].className == "nn.VBLinear"
    got = m.fields["modules"][0].fields
    assert np.array_equal(got["means"], w) and got["means"].dtype == np.float32 and got["W"] == 20
    assert back["alias"] is got["means"]                         # one object written once, read back as one
    assert back["vb_indices"] == [2] and back["opt"] == obj["opt"]
    assert back["long"].dtype == np.int64 and back["dbl"].dtype == np.float64 and list(back["byte"]) == [1, 255]
    assert back["empty"].size == 0
    assert m.fields["modules"][1].fields == {"inplace": False}


def test_reader_takes_strided_tensors_and_rejects_functions():
    # a 2 x 2 view with strides (1, 2) and storage offset 2 (1-based) into a 6-element storage: what a Lua-side
    # `t:t()` or `narrow` produces
    st = np.arange(6, dtype=np.float64)
    blob = (i32(4, 1) + s("V 1") + s("torch.DoubleTensor") + i32(2) + i64(2, 2) + i64(1, 2) + i64(2)
            + i32(4, 2) + s("V 1") + s("torch.DoubleStorage") + i64(6) + st.tobytes())
    assert np.array_equal(t7file.loads(blob), np.array([[1.0, 3.0], [2.0, 4.0]]))
    with pytest.raises(ValueError, match="Lua function"):
        t7file.loads(i32(6, 1))
    with pytest.raises(EOFError):
        t7file.loads(i32(1) + b"\0\0")
    with pytest.raises(TypeError):
        t7file.dumps(lambda: 0)
    with pytest.raises(TypeError):
        t7file.dumps(np.zeros(2, np.complex64))


def test_safe_save_keeps_the_previous_file(tmp_path):
    d = str(tmp_path / "exp" / "parameters")
    utils.safe_save(np.ones(3, np.float32), d, "means")
    utils.safe_save(np.zeros(3, np.float32), d, "means")
    assert np.array_equal(t7file.load(os.path.join(d, "means")), np.zeros(3, np.float32))
    assert np.array_equal(t7file.load(os.path.join(d, "means.old")), np.ones(3, np.float32))
    assert utils.file_exists(os.path.join(d, "means")) and not utils.file_exists(os.path.join(d, "vars"))


def test_get_accuracy_and_normalize():
    out = np.array([[0.1, 0.9], [0.8, 0.2], [0.5, 0.5]])
    assert utils.get_accuracy(out, np.array([1, 0, 1])) == pytest.approx(200.0 / 3)     # the tie goes to the first maximum
    assert utils.get_accuracy(np.array([0.1, 0.9]), np.array([1, 1, 0])) == pytest.approx(200.0 / 3)   # 1-D: one row vs all
    x = np.random.RandomState(1).rand(50, 4).astype(np.float32) * 9 + 3
    x0 = x.copy()
    mean, std = utils.normalize(x)
    assert mean == pytest.approx(x0.mean(), rel=1e-6) and std == pytest.approx(x0.std(ddof=1), rel=1e-6)   # torch's unbiased std
    assert abs(x.mean()) < 1e-5 and x.std(ddof=1) == pytest.approx(1.0, rel=1e-5)
    import torch
    xt = torch.from_numpy(x0.copy())
    m2, s2 = utils.normalize(xt)
    assert m2 == pytest.approx(mean, rel=1e-6) and s2 == pytest.approx(std, rel=1e-6) and np.allclose(xt.numpy(), x, atol=1e-5)
    assert utils.isnan(float("nan")) and not utils.isnan(1.0)
    assert utils.norm_pdf(0.0, 0.0, 1.0) == pytest.approx(0.3989422804014327)
    assert sorted(utils.shuffle(range(10), np.random.RandomState(0))) == list(range(10))
    sel = utils.select_data({"inputs": x, "targets": np.arange(50)}, [3, 1])
    assert np.array_equal(sel["targets"], [3, 1]) and np.array_equal(sel["inputs"][0], x[3])


def test_num_grad_matches_the_analytic_kl_gradient():
    # d/dlv of sum(exp(lv)) under a uniform shift = sum(exp(lv)) (utils.lua:49-62's usage in main.lua:121-135)
    lv = np.log(np.full(20, 1e-3))
    g = utils.num_grad(lv, lambda: np.exp(lv).sum())
    assert g == pytest.approx(np.exp(lv).sum(), rel=1e-5)
    assert np.allclose(lv, np.log(1e-3))


def _write_idx(path, a, gz=False):
    head = struct.pack(">HBB", 0, 8, a.ndim) + struct.pack(f">{a.ndim}I", *a.shape)
    (gzip.open if gz else open)(path, "wb").write(head + a.astype(np.uint8).tobytes())


def test_mnist_idx_reader_and_minibatches(tmp_path):
    rs = np.random.RandomState(0)
    tri, trl = rs.randint(0, 256, (30, 28, 28)), rs.randint(0, 10, 30)
    tei, tel = rs.randint(0, 256, (12, 28, 28)), rs.randint(0, 10, 12)
    _write_idx(str(tmp_path / "train-images-idx3-ubyte"), tri)
    _write_idx(str(tmp_path / "train-labels-idx1-ubyte"), trl)
    _write_idx(str(tmp_path / "t10k-images-idx3-ubyte.gz"), tei, gz=True)
    _write_idx(str(tmp_path / "t10k-labels-idx1-ubyte.gz"), tel, gz=True)
    train, test = data.getMnist(str(tmp_path))
    assert train["inputs"].shape == (30, 28, 28) and train["inputs"].dtype == np.float32
    assert np.array_equal(train["targets"], trl) and np.array_equal(test["targets"], tel)
    ref = tri.astype(np.float32)
    ref = (ref - ref.mean()) / ref.std(ddof=1)                  # data.lua:25: each split by its own statistics
    assert np.allclose(train["inputs"], ref, atol=1e-5)
    x, t = train.create_minibatch(10, 8, 30, (28, 28))
    assert x.shape == (8, 1, 28, 28) and np.array_equal(x[:, 0], train["inputs"][10:18]) and np.array_equal(t, trl[10:18])
    x, t = train.create_minibatch(25, 8, 30, (28, 28))          # short last batch: full-size block, 5 live rows
    assert x.shape == (8, 1, 28, 28) and np.array_equal(x[:5, 0], train["inputs"][25:]) and not x[5:].any()
    with pytest.raises(FileNotFoundError):
        data.getMnist(str(tmp_path / "nowhere"))
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad"), "wb").write(b"\x00\x00\x0d\x01\x00\x00\x00\x01\x00")
        data._read_idx(str(tmp_path / "bad"))


def test_bacteria_folds_follow_the_reference_index_rule():
    # data.lua:52-70 with n = 95, k = 10: fold = round(9.5) = 10, train 90, test 5 = rows [(i-1) 10, (i-1) 10 + 5)
    d = {"inputs": np.arange(95 * 3, dtype=np.float32).reshape(95, 3), "targets": np.arange(95) % 2}
    for i in range(1, 11):
        tr, te = data.getBacteriaFold(d, i, 10)
        lo = (i - 1) * 10
        assert tr["inputs"].shape[0] == 90 and te["inputs"].shape[0] == 5
        assert np.array_equal(te["inputs"][:, 0], d["inputs"][lo:lo + 5, 0])
        assert np.array_equal(te["targets"], d["targets"][lo:lo + 5])
        assert not set(te["inputs"][:, 0]) & set(tr["inputs"][:, 0])


def test_synthetic_digits_are_reproducible():
    a, _ = data.synthetic_digits(64, 16, seed=3)
    b, _ = data.synthetic_digits(64, 16, seed=3)
    assert np.array_equal(a["inputs"], b["inputs"]) and np.array_equal(a["targets"], b["targets"])
    assert a["inputs"].shape == (64, 28, 28) and set(np.unique(a["targets"])) <= set(range(10))


def test_t7_round_trip_property():
    """Random nested tables / tensors survive dumps -> loads (hypothesis; the reader is the writer's only judge here)."""
    hyp = pytest.importorskip("hypothesis")
    st = pytest.importorskip("hypothesis.strategies")
    hnp = pytest.importorskip("hypothesis.extra.numpy")

    dtypes = st.sampled_from([np.float32, np.float64, np.int64, np.int32, np.int16, np.uint8, np.int8])
    tensors = dtypes.flatmap(lambda dt: hnp.arrays(dt, hnp.array_shapes(min_dims=1, max_dims=3, min_side=1, max_side=5),
                                                   elements=st.integers(0, 100)))
    leaves = st.one_of(st.booleans(), st.integers(-2 ** 40, 2 ** 40), st.floats(allow_nan=False, allow_infinity=True),
                       st.text(alphabet=st.characters(min_codepoint=32, max_codepoint=126), max_size=12), tensors)
    keys = st.text(alphabet="abcdefghijklmnopqrstuvwxyz_", min_size=1, max_size=6)
    trees = st.recursive(leaves, lambda ch: st.one_of(st.lists(ch, min_size=1, max_size=4),
                                                      st.dictionaries(keys, ch, min_size=1, max_size=4)), max_leaves=12)

    def same(a, b):
        if isinstance(a, np.ndarray):
            return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
        if isinstance(a, dict):
            return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, (list, tuple)):
            return isinstance(b, list) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, bool) or isinstance(b, bool):
            return a is b
        return a == b                                  # Lua has one number type: 3 and 3.0 are the same value

    @hyp.settings(max_examples=150, deadline=None)
    @hyp.given(trees)
    def check(obj):
        assert same(obj, t7file.loads(t7file.dumps(obj)))

    check()
