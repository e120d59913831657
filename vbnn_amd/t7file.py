"""Torch7 binary serialisation (`torch.save` / `torch.load`), the format of the reference's on-disk files:
`<network_name>/model` (utils.lua:73-80, main.lua:181), `<dir>/parameters/{means,vars}` and `<dir>/opt`
(mainviz.lua:11-15), `gutbacteria_shuffled_binary.torch` (data.lua:36).

torch7's `File.lua` is un-vendored and unpinned in the reference; this restates its published binary layout
(little-endian, `long` = 8 bytes):

    object  := int32 type, payload
    NIL 0   : -
    NUMBER 1: float64
    STRING 2: int32 length, bytes
    TABLE 3 : int32 index; first time only: int32 n, n x (object key, object value)
    TORCH 4 : int32 index; first time only: string "V 1", string className, class payload
    BOOLEAN 5: int32 0/1
    torch.XTensor  payload: int32 nDim, int64 size[nDim], int64 stride[nDim], int64 storageOffset (1-based), object storage
    torch.XStorage payload: int64 n, n raw elements
    any other class (Lua-side `torch.class`, e.g. nn.Sequential, nn.VBLinear): one TABLE object with the fields

`index` numbers tables, torch objects and storages in order of first appearance; an object referenced twice is written
once and read back as one object (tensors are written contiguous, each with its own storage). Lua functions (types
6-8, `string.dump` byte code) cannot be produced or consumed here: `save` refuses them, `load` raises on them -- which is why this package writes the DATA of
a run (parameters, opt, the nn.Sequential's tensors), not the reference's `mlp` table with its methods.

Python mapping: None, bool, int/float (-> NUMBER), str, dict and list/tuple (-> TABLE, lists get keys 1..n),
numpy arrays / torch tensors (-> torch.FloatTensor etc. by dtype), `T7Object(className, fields)`.
"""
import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5
TYPE_FUNCTION, LEGACY_TYPE_RECUR_FUNCTION, TYPE_RECUR_FUNCTION = 6, 7, 8

_KINDS = {                                   # numpy dtype -> torch7 class stem
    np.dtype("float32"): "Float", np.dtype("float64"): "Double", np.dtype("int64"): "Long",
    np.dtype("int32"): "Int", np.dtype("int16"): "Short", np.dtype("uint8"): "Byte", np.dtype("int8"): "Char",
}
_DTYPES = {v: k for k, v in _KINDS.items()}


class T7Object:
    """A Lua-side torch class instance: class name + field table (e.g. T7Object('nn.VBLinear', {...}))."""

    def __init__(self, className, fields):
        self.className, self.fields = className, fields

    def __repr__(self):
        return f"T7Object({self.className!r}, {list(self.fields)!r})"


def _as_numpy(x):
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "detach") and hasattr(x, "cpu"):              # torch tensor, without importing torch here
        return x.detach().cpu().numpy()
    return None


class _Writer:
    def __init__(self, f):
        self.f, self.seen, self.keep, self.n = f, {}, [], 0

    def i32(self, v):
        self.f.write(struct.pack("<i", v))

    def i64(self, *v):
        self.f.write(struct.pack(f"<{len(v)}q", *v))

    def string(self, s):
        b = s if isinstance(s, bytes) else s.encode("utf-8")
        self.i32(len(b))
        self.f.write(b)

    def _index(self, obj):
        """Writes the object's index; True when the body still has to follow."""
        key = id(obj)
        if key in self.seen:
            self.i32(self.seen[key])
            return False
        self.n += 1
        self.seen[key] = self.n
        self.keep.append(obj)                                     # ids stay unique while we hold the object
        self.i32(self.n)
        return True

    def table(self, items):
        items = [(k, v) for k, v in items if v is not None]      # a Lua table holds no nil values
        self.i32(len(items))
        for k, v in items:
            self.obj(k)
            self.obj(v)

    def obj(self, o):
        if o is None:
            self.i32(TYPE_NIL)
        elif isinstance(o, (bool, np.bool_)):
            self.i32(TYPE_BOOLEAN)
            self.i32(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.i32(TYPE_NUMBER)
            self.f.write(struct.pack("<d", float(o)))
        elif isinstance(o, (str, bytes)):
            self.i32(TYPE_STRING)
            self.string(o)
        elif isinstance(o, dict):
            self.i32(TYPE_TABLE)
            if self._index(o):
                self.table(list(o.items()))
        elif isinstance(o, (list, tuple)):
            self.i32(TYPE_TABLE)
            if self._index(o):
                self.table([(i + 1, v) for i, v in enumerate(o)])
        elif isinstance(o, T7Object):
            self.i32(TYPE_TORCH)
            if self._index(o):
                self.string("V 1")
                self.string(o.className)
                self.obj(o.fields)
        elif callable(o):
            raise TypeError("t7file: Lua functions (string.dump byte code) cannot be written from here")
        else:
            a = _as_numpy(o)
            if a is None or a.dtype not in _KINDS:
                raise TypeError(f"t7file: cannot serialise {type(o).__name__}" + (f" of dtype {a.dtype}" if a is not None else ""))
            self.tensor(o, a)

    def tensor(self, key, a):
        kind = _KINDS[a.dtype]
        self.i32(TYPE_TORCH)
        if not self._index(key):
            return
        self.string("V 1")
        self.string(f"torch.{kind}Tensor")
        a = np.ascontiguousarray(a)
        self.i32(a.ndim)
        self.i64(*a.shape)
        self.i64(*[s // a.itemsize for s in a.strides])
        self.i64(1)                                               # storageOffset, 1-based
        if a.size == 0:
            self.i32(TYPE_NIL)                                    # an empty tensor has no storage
            return
        self.i32(TYPE_TORCH)
        self.n += 1
        self.i32(self.n)                                          # the storage is an object of its own
        self.string("V 1")
        self.string(f"torch.{kind}Storage")
        self.i64(a.size)
        self.f.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())


class _Reader:
    def __init__(self, f):
        self.f, self.objects = f, {}

    def take(self, fmt):
        n = struct.calcsize(fmt)
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError("t7file: truncated file")
        return struct.unpack(fmt, b)

    def i32(self):
        return self.take("<i")[0]

    def i64(self, n=1):
        return self.take(f"<{n}q")

    def string(self):
        n = self.i32()
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError("t7file: truncated string")
        return b.decode("utf-8", errors="replace")

    def obj(self):
        t = self.i32()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            v = self.take("<d")[0]
            return int(v) if (v == v and abs(v) < 2 ** 53 and v == int(v)) else v
        if t == TYPE_BOOLEAN:
            return self.i32() == 1
        if t == TYPE_STRING:
            return self.string()
        if t == TYPE_TABLE:
            idx = self.i32()
            if idx in self.objects:
                return self.objects[idx]
            d = {}
            self.objects[idx] = d
            for _ in range(self.i32()):
                k = self.obj()
                d[k] = self.obj()
            n = len(d)                                            # an array-like table comes back as a list
            if n and all(isinstance(k, int) for k in d) and set(d) == set(range(1, n + 1)):
                lst = [d[i] for i in range(1, n + 1)]
                self.objects[idx] = lst
                return lst
            return d
        if t == TYPE_TORCH:
            idx = self.i32()
            if idx in self.objects:
                return self.objects[idx]
            version = self.string()
            cls = self.string() if version.startswith("V ") else version     # pre-versioning files: the name comes first
            if cls.startswith("torch.") and cls.endswith("Tensor"):
                kind = cls[6:-6]
                if kind not in _DTYPES:
                    raise ValueError(f"t7file: unsupported tensor class {cls}")
                nd = self.i32()
                size, stride = self.i64(nd), self.i64(nd)
                off = self.i64()[0] - 1
                storage = self.obj()
                if storage is None or nd == 0:
                    a = np.zeros(size if nd else (0,), dtype=_DTYPES[kind])
                else:
                    a = np.lib.stride_tricks.as_strided(storage[off:], shape=size,
                                                        strides=[s * storage.itemsize for s in stride]).copy()
                self.objects[idx] = a
                return a
            if cls.startswith("torch.") and cls.endswith("Storage"):
                kind = cls[6:-7]
                n = self.i64()[0]
                dt = _DTYPES[kind].newbyteorder("<")
                b = self.f.read(n * dt.itemsize)
                if len(b) != n * dt.itemsize:
                    raise EOFError("t7file: truncated storage")
                a = np.frombuffer(b, dtype=dt).astype(_DTYPES[kind])
                self.objects[idx] = a
                return a
            o = T7Object(cls, None)
            self.objects[idx] = o
            o.fields = self.obj()
            return o
        if t in (TYPE_FUNCTION, LEGACY_TYPE_RECUR_FUNCTION, TYPE_RECUR_FUNCTION):
            raise ValueError("t7file: the file holds a Lua function (byte code); only data objects can be read here")
        raise ValueError(f"t7file: unknown object type {t}")


def save(path, obj):
    """torch.save(path, obj) in torch7's binary format."""
    with open(path, "wb") as f:
        _Writer(f).obj(obj)


def dumps(obj):
    import io
    b = io.BytesIO()
    _Writer(b).obj(obj)
    return b.getvalue()


def load(path):
    """torch.load(path): tensors come back as numpy arrays, tables as dict / list, other classes as T7Object."""
    with open(path, "rb") as f:
        return _Reader(f).obj()


def loads(data):
    import io
    return _Reader(io.BytesIO(data)).obj()
