"""Record / replay the call sequence a host program makes on a `sutraWrap`-shaped module.

BUILD-CONTAINER TOOL + test helper (test infrastructure).  `Recorder.install(sw, cw)` wraps the
classes of a facade module in proxies; everything the reference's Python then does at the drop-in
boundary -- constructors, load_arrays, per-frame calls, attribute reads,
`np.array(d_xxx)` copies -- is appended to a log of plain data:

    {"op": "new",  "cls": "Sensors", "id": 3, "args": [...], "kwargs": {...}}
    {"op": "call", "path": [("root", 3), ("attr", "d_wfs"), ("item", 0), ("attr", "comp_image")],
     "args": [...], "kwargs": {...}, "result": <plain value or None>}
    {"op": "read", "path": [...], "value": ndarray}       # np.array(obj)
    {"op": "get",  "path": [...], "value": scalar/tuple/str}
    {"op": "set",  "path": [...], "value": ...}

Objects inside arguments are stored as {"__ref__": path}; the carma context as {"__ctx__": 1}.
`replay(log, sw, cw, compare)` re-issues the same sequence on ANOTHER module with the same surface
(ao_marl_amd.sutra_facade over libaomarl_hip.so on the GPU box) and hands every recorded read to
`compare(entry, got)`: no reference code is needed at replay time, only this data.
"""

import numpy as np

PLAIN = (int, float, bool, str, bytes, type(None), np.integer, np.floating, np.bool_)
ROOT_CLASSES = ("Telescope", "Atmos", "Dms", "Sensors", "Target", "Rtc_FFF")


def _is_plain(v):
    if isinstance(v, PLAIN) or isinstance(v, np.ndarray):
        return True
    if isinstance(v, (tuple, list)):
        return all(_is_plain(x) for x in v)
    if isinstance(v, dict):
        return all(_is_plain(x) for x in v.values())
    return False


def _to_plain(v):
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, np.ndarray):
        return np.array(v, subok=False, copy=True)       # plain ndarray (never a subclass)
    if isinstance(v, (tuple, list)):
        return [_to_plain(x) for x in v]
    if isinstance(v, dict):
        return {k: _to_plain(x) for k, x in v.items()}
    return v


class Proxy(object):
    """Stands in front of a facade object (or a list of them, or a bound method)."""

    def __init__(self, rec, real, path):
        object.__setattr__(self, "_rec", rec)
        object.__setattr__(self, "_real", real)
        object.__setattr__(self, "_path", tuple(path))

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return self._rec.wrap(getattr(self._real, name), self._path + (("attr", name),))

    def __setattr__(self, name, value):
        self._rec.log.append({"op": "set", "path": list(self._path + (("attr", name),)),
                              "value": self._rec.encode(value)})
        setattr(self._real, name, self._rec.unwrap(value))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self._real)))]
        if i < 0:
            i += len(self._real)
        return self._rec.wrap(self._real[i], self._path + (("item", int(i)),))

    def __len__(self):
        return len(self._real)

    def __iter__(self):
        for i in range(len(self._real)):
            yield self[i]

    def __bool__(self):
        return True

    def __array__(self, dtype=None, copy=None):
        arr = np.array(self._real)
        self._rec.log.append({"op": "read", "path": list(self._path), "value": np.array(arr, subok=False, copy=True)})
        return arr.astype(dtype) if dtype is not None else arr

    def __call__(self, *a, **k):
        rec = self._rec
        entry = {"op": "call", "path": list(self._path), "args": rec.encode(a), "kwargs": rec.encode(k)}
        rec.log.append(entry)
        ret_id = len(rec.log)                         # replay: index of this entry + 1
        res = self._real(*rec.unwrap(a), **rec.unwrap(k))
        if _is_plain(res):
            entry["result"] = _to_plain(res)
            return res
        return rec.wrap(res, self._path + (("ret", ret_id),))


class Recorder(object):
    def __init__(self):
        self.log = []
        self.nroots = 0
        self.ctx_type = None

    # ---- argument coding
    def encode(self, v):
        if isinstance(v, Proxy):
            return {"__ref__": list(v._path)}
        if self.ctx_type is not None and isinstance(v, self.ctx_type):
            return {"__ctx__": 1}
        if isinstance(v, (tuple, list)):
            return [self.encode(x) for x in v]
        if isinstance(v, dict):
            return {k: self.encode(x) for k, x in v.items()}
        if _is_plain(v):
            return _to_plain(v)
        return {"__repr__": repr(v)}

    def unwrap(self, v):
        if isinstance(v, Proxy):
            return v._real
        if isinstance(v, tuple):
            return tuple(self.unwrap(x) for x in v)
        if isinstance(v, list):
            return [self.unwrap(x) for x in v]
        if isinstance(v, dict):
            return {k: self.unwrap(x) for k, x in v.items()}
        return v

    def wrap(self, v, path):
        if callable(v) and not isinstance(v, type):
            return Proxy(self, v, path)
        if _is_plain(v):
            self.log.append({"op": "get", "path": list(path), "value": _to_plain(v)})
            return v
        return Proxy(self, v, path)

    # ---- installation
    def install(self, sw, cw):
        """Replace the root classes of module `sw` by recording constructors."""
        self.ctx_type = cw.context
        rec = self
        for name in ROOT_CLASSES:
            real_cls = getattr(sw, name)

            def ctor(*a, _cls=real_cls, _name=name, **k):
                rid = rec.nroots
                rec.nroots += 1
                rec.log.append({"op": "new", "cls": _name, "id": rid, "args": rec.encode(a),
                                "kwargs": rec.encode(k)})
                return Proxy(rec, _cls(*rec.unwrap(a), **rec.unwrap(k)), (("root", rid),))
            setattr(sw, name, ctor)

    def save(self, path, max_read=4096):
        """Recorded READS larger than max_read elements keep a strided subsample (replay compares
        the same subsample); equal input arrays are stored once."""
        import hashlib
        pool = {}

        def intern(v):
            if isinstance(v, np.ndarray) and v.nbytes > 1024:
                key = (v.shape, str(v.dtype), hashlib.sha1(np.ascontiguousarray(v).view(np.uint8)).hexdigest())
                return pool.setdefault(key, v)
            if isinstance(v, list):
                return [intern(x) for x in v]
            if isinstance(v, dict):
                return {k: intern(x) for k, x in v.items()}
            return v
        for e in self.log:
            if e["op"] == "read" and e["value"].size > max_read:
                step = -(-e["value"].size // max_read)
                e["shape"], e["stride"] = list(e["value"].shape), step
                e["value"] = e["value"].reshape(-1)[::step].copy()
            for k in ("args", "kwargs", "value"):
                if k in e:
                    e[k] = intern(e[k])
        save_log(self.log, path)
        n = {}
        for e in self.log:
            n[e["op"]] = n.get(e["op"], 0) + 1
        print("wrote", path, n)


def save_log(log, path):
    """The call log as ONE .npz without pickled objects: the structure as JSON (key "_log", utf-8 bytes), every
    array as its own entry ("a<i>", referenced from the JSON as {"__nd__": i}; equal arrays stored once)."""
    import json
    arrays, index = [], {}

    def enc(v):
        if isinstance(v, np.ndarray):
            if id(v) not in index:
                index[id(v)] = len(arrays)
                arrays.append(v)
            return {"__nd__": index[id(v)]}
        if isinstance(v, (np.integer,)):
            return int(v)
        if isinstance(v, (np.floating,)):
            return {"__f__": float(v), "dtype": str(v.dtype)}
        if isinstance(v, (np.bool_,)):
            return bool(v)
        if isinstance(v, tuple):
            return {"__t__": [enc(x) for x in v]}
        if isinstance(v, list):
            return [enc(x) for x in v]
        if isinstance(v, dict):
            return {str(k): enc(x) for k, x in v.items()}
        if v is None or isinstance(v, (str, int, float, bool)):
            return v
        raise TypeError("cannot store %r in the call log" % type(v))
    doc = json.dumps(enc(log)).encode("utf-8")
    np.savez_compressed(path, _log=np.frombuffer(doc, dtype=np.uint8), **{"a%d" % i: a for i, a in enumerate(arrays)})


def load(path):
    """Inverse of save_log (np.load with allow_pickle=False: data only)."""
    import json
    z = np.load(path, allow_pickle=False)
    cache = {}

    def dec(v):
        if isinstance(v, dict):
            if "__nd__" in v:
                i = v["__nd__"]
                if i not in cache:
                    cache[i] = z["a%d" % i]
                return cache[i]
            if "__t__" in v:
                return tuple(dec(x) for x in v["__t__"])
            if "__f__" in v:
                return np.dtype(v["dtype"]).type(v["__f__"])
            return {k: dec(x) for k, x in v.items()}
        if isinstance(v, list):
            return [dec(x) for x in v]
        return v
    return dec(json.loads(bytes(z["_log"]).decode("utf-8")))


# ------------------------------------------------------------------------------------ replay
def replay(log, sw, cw, compare, stop_after=None):
    """Re-issue `log` on the classes of module `sw`.  compare(entry, got) is called for every
    recorded read / get / plain call result."""
    roots, rets = {}, {}
    ctx = cw.context.get_instance_1gpu(0)

    def resolve(path):
        obj = None
        for kind, key in path:
            if kind == "root":
                obj = roots[key]
            elif kind == "attr":
                obj = getattr(obj, key)
            elif kind == "item":
                obj = obj[key]
            elif kind == "ret":
                obj = rets[key]
        return obj

    def decode(v):
        if isinstance(v, dict):
            if "__ref__" in v:
                return resolve([tuple(p) for p in v["__ref__"]])
            if "__ctx__" in v:
                return ctx
            if "__repr__" in v:
                raise ValueError("unreplayable argument %s" % v["__repr__"])
            return {k: decode(x) for k, x in v.items()}
        if isinstance(v, list):
            return [decode(x) for x in v]
        return v

    for n, e in enumerate(log):
        if stop_after is not None and n >= stop_after:
            break
        op = e["op"]
        if op == "new":
            roots[e["id"]] = getattr(sw, e["cls"])(*decode(e["args"]), **decode(e["kwargs"]))
        elif op == "call":
            path = [tuple(p) for p in e["path"]]
            res = resolve(path)(*decode(e["args"]), **decode(e["kwargs"]))
            if "result" in e:
                if e["result"] is not None:
                    compare(e, res)
            else:
                rets[n + 1] = res             # the recorder numbered it len(log) AFTER appending
        elif op == "read":
            got = np.array(resolve([tuple(p) for p in e["path"]]))
            if "stride" in e:                 # large array: the recorded value is a strided subsample
                assert list(got.shape) == e["shape"], (path_str([tuple(p) for p in e["path"]]), got.shape)
                got = got.reshape(-1)[::e["stride"]]
            compare(e, got)
        elif op == "get":
            compare(e, resolve([tuple(p) for p in e["path"]]))
        elif op == "set":
            path = [tuple(p) for p in e["path"]]
            setattr(resolve(path[:-1]), path[-1][1], decode(e["value"]))
    return roots


def path_str(path):
    out = ""
    for kind, key in path:
        out += {"root": "#%s", "attr": ".%s", "item": "[%s]", "ret": "()%s"}[kind] % (key,)
    return out
