"""A minimal reader for the HDF5 files Keras writes its weights into -- enough of the format (HDF5 File Format Specification 3.0) to
load `weights_best.h5` / `weights_last.h5` of a csbdeep / StarDist model folder (stardist/models/base.py:232-252,
csbdeep BaseModel._find_and_load_weights) without h5py:

  superblock           versions 0 - 3
  groups               old style (symbol-table message: v1 B-tree + local heap + symbol-table nodes -- what HDF5's default
                       "earliest" file format, i.e. h5py's default, writes) and new-style compact groups (link messages)
  object headers       version 1 (with continuation blocks) and version 2 ("OHDR" / "OCHK")
  attributes           message versions 1 - 3; fixed-length strings, variable-length strings (global heap), integers, floats
  datasets             contiguous, compact and chunked (v1 B-tree, no filters) layouts; IEEE floats and integers, either byte order

Not supported (clear errors): compressed / filtered chunks, dense groups and dense attribute storage (fractal heaps, only written with
libver='latest'), compound / array / reference datatypes.  Tested on files written by the real HDF5 library in Keras' layout
(tests/golden/make_keras_h5_fixture.py, tests/test_cpu_pretrained.py).
"""
from collections import OrderedDict

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(IOError):
    pass


class _Datatype(object):
    __slots__ = ("cls", "size", "dtype", "vlen_string", "base")


class File(object):
    """read-only view of an HDF5 file held in memory: .root is the root Group"""

    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            self.buf = bytes(path_or_bytes)
        elif hasattr(path_or_bytes, "read"):
            self.buf = path_or_bytes.read()
        else:
            with open(path_or_bytes, "rb") as fh:
                self.buf = fh.read()
        self._superblock()

    # ------------------------------------------------------------------ low level
    def _u(self, off, n):
        return int.from_bytes(self.buf[off:off + n], "little")

    def _addr(self, off):
        return self._u(off, self.so)

    def _len(self, off):
        return self._u(off, self.sl)

    def _superblock(self):
        sig = b"\x89HDF\r\n\x1a\n"
        base = 0
        while True:                                   # the superblock may sit at 0, 512, 1024, ...
            if self.buf[base:base + 8] == sig:
                break
            base = 512 if base == 0 else base * 2
            if base >= len(self.buf):
                raise HDF5Error("not an HDF5 file (no superblock signature)")
        ver = self.buf[base + 8]
        if ver in (0, 1):
            self.so, self.sl = self.buf[base + 13], self.buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base_addr = self._addr(p)
            p += 4 * self.so                          # base, free-space info, end of file, driver info
            # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch
            self.root_addr = self._addr(p + self.so)
        elif ver in (2, 3):
            self.so, self.sl = self.buf[base + 9], self.buf[base + 10]
            p = base + 12
            self.base_addr = self._addr(p)
            self.root_addr = self._addr(p + 3 * self.so)
        else:
            raise HDF5Error("superblock version %d not supported" % ver)
        if self.base_addr in (UNDEF, UNDEF >> (64 - 8 * self.so)):
            self.base_addr = 0
        self.base_addr += 0
        self.root = Group(self, self.root_addr, "/")

    # ------------------------------------------------------------------ object headers
    def messages(self, addr):
        """[(type, flags, payload offset, payload size)] of the object header at addr (all continuation blocks followed)"""
        a = self.base_addr + addr
        out = []
        if self.buf[a:a + 4] == b"OHDR":
            if self.buf[a + 4] != 2:
                raise HDF5Error("object header version %d" % self.buf[a + 4])
            flags = self.buf[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16                               # access, modification, change, birth times
            if flags & 0x10:
                p += 4                                # max compact / min dense attributes
            nsz = 1 << (flags & 3)
            chunk0 = self._u(p, nsz)
            p += nsz
            track = bool(flags & 0x04)
            blocks = [(p, chunk0)]
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 <= end - 0:
                    t, sz, fl = self.buf[p], self._u(p + 1, 2), self.buf[p + 3]
                    p += 4 + (2 if track else 0)
                    if p + sz > end:
                        break
                    if t == 0x10:
                        ca, cl = self._addr(p), self._len(p + self.so)
                        cb = self.base_addr + ca
                        if self.buf[cb:cb + 4] != b"OCHK":
                            raise HDF5Error("bad object header continuation")
                        blocks.append((cb + 4, cl - 8))   # signature ... checksum
                    elif t != 0:
                        out.append((t, fl, p, sz))
                    p += sz
            return out
        if self.buf[a] != 1:
            raise HDF5Error("object header version %d at %d" % (self.buf[a], addr))
        nmsg, hsize = self._u(a + 2, 2), self._u(a + 8, 4)
        blocks = [(a + 16, hsize)]                    # 12 bytes of prefix + 4 of padding: messages start 8-byte aligned
        while blocks and len(out) < nmsg + 64:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end:
                t, sz, fl = self._u(p, 2), self._u(p + 2, 2), self.buf[p + 4]
                p += 8
                if t == 0x10:
                    blocks.append((self.base_addr + self._addr(p), self._len(p + self.so)))
                elif t != 0:
                    out.append((t, fl, p, sz))
                p += sz
        return out

    # ------------------------------------------------------------------ datatype / dataspace
    def datatype(self, p):
        cv = self.buf[p]
        cls, ver = cv & 15, cv >> 4
        b0, b8, b16 = self.buf[p + 1], self.buf[p + 2], self.buf[p + 3]
        size = self._u(p + 4, 4)
        dt = _Datatype()
        dt.cls, dt.size, dt.vlen_string, dt.base, dt.dtype = cls, size, False, None, None
        order = ">" if (b0 & 1) else "<"
        if cls == 0:                                  # fixed point
            dt.dtype = np.dtype("%s%s%d" % (order, "i" if (b0 & 8) else "u", size))
        elif cls == 1:                                # floating point (IEEE layouts only)
            if size not in (2, 4, 8):
                raise HDF5Error("float of %d bytes" % size)
            dt.dtype = np.dtype("%sf%d" % (order, size))
        elif cls == 3:                                # fixed-length string
            dt.dtype = np.dtype("S%d" % size)
        elif cls == 9:                                # variable length: string (type 1) or sequence
            dt.vlen_string = (b0 & 15) == 1
            dt.base = self.datatype(p + 8)
            if not dt.vlen_string:
                raise HDF5Error("variable-length sequences are not supported")
        else:
            raise HDF5Error("datatype class %d is not supported" % cls)
        return dt

    def dataspace(self, p):
        ver, rank = self.buf[p], self.buf[p + 1]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.buf[p + 3] == 2:                  # null dataspace
                return None
            q = p + 4
        else:
            raise HDF5Error("dataspace version %d" % ver)
        return tuple(self._len(q + k * self.sl) for k in range(rank))

    def global_heap_object(self, coll_addr, index):
        a = self.base_addr + coll_addr
        if self.buf[a:a + 4] != b"GCOL":
            raise HDF5Error("bad global heap collection")
        size = self._len(a + 8)
        p, end = a + 8 + self.sl, a + size
        while p + 8 + self.sl <= end:
            idx, osz = self._u(p, 2), self._len(p + 8)
            if idx == 0:
                break
            if idx == index:
                return self.buf[p + 8 + self.sl:p + 8 + self.sl + osz]
            p += 8 + self.sl + ((osz + 7) // 8) * 8
        raise HDF5Error("global heap object %d not found" % index)

    def decode(self, dt, shape, raw):
        n = 1 if shape in (None, ()) else int(np.prod(shape))
        if dt.vlen_string:
            out = []
            rec = 4 + self.so + 4
            for k in range(n):
                q = k * rec
                ln = int.from_bytes(raw[q:q + 4], "little")
                ca = int.from_bytes(raw[q + 4:q + 4 + self.so], "little")
                ix = int.from_bytes(raw[q + 4 + self.so:q + rec], "little")
                out.append(self.global_heap_object(ca, ix)[:ln] if ln else b"")
            arr = np.array(out, dtype=object)
        else:
            arr = np.frombuffer(raw, dtype=dt.dtype, count=n)
        if shape in (None, ()):
            return arr[0]
        return arr.reshape(shape)

    def attributes(self, addr):
        out = OrderedDict()
        for t, fl, p, sz in self.messages(addr):
            if t == 0x15:
                if self._addr(p + 2) not in (UNDEF, UNDEF >> (64 - 8 * self.so)):
                    raise HDF5Error("dense attribute storage (libver='latest' files) is not supported")
                continue
            if t != 0x0C:
                continue
            ver = self.buf[p]
            nsz, dsz, ssz = self._u(p + 2, 2), self._u(p + 4, 2), self._u(p + 6, 2)
            if ver == 1:
                q = p + 8
                pad = lambda v: (v + 7) // 8 * 8
            elif ver in (2, 3):
                q = p + 8 + (1 if ver == 3 else 0)
                pad = lambda v: v
                if self.buf[p + 1] & 3:
                    raise HDF5Error("shared attribute datatype / dataspace")
            else:
                raise HDF5Error("attribute message version %d" % ver)
            name = self.buf[q:q + nsz].split(b"\0")[0].decode("utf8")
            q += pad(nsz)
            dt = self.datatype(q)
            q += pad(dsz)
            shape = self.dataspace(q)
            q += pad(ssz)
            n = 1 if shape in (None, ()) else int(np.prod(shape))
            nbytes = n * (4 + self.so + 4 if dt.vlen_string else dt.dtype.itemsize)
            out[name] = self.decode(dt, shape, self.buf[q:q + nbytes])
        return out


class Dataset(object):
    def __init__(self, f, addr, name):
        self.f, self.addr, self.name = f, addr, name
        self._dt = self._shape = self._layout = None
        self._filtered = False
        for t, fl, p, sz in f.messages(addr):
            if t == 0x01:
                self._shape = f.dataspace(p)
            elif t == 0x03:
                self._dt = f.datatype(p)
            elif t == 0x08:
                self._layout = p
            elif t == 0x0B:
                self._filtered = True
        if self._dt is None or self._layout is None:
            raise HDF5Error("%s: not a dataset" % name)

    @property
    def shape(self):
        return self._shape

    @property
    def attrs(self):
        return self.f.attributes(self.addr)

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a.astype(dtype) if dtype is not None else a

    def read(self):
        f, p = self.f, self._layout
        ver = f.buf[p]
        shape = self._shape if self._shape is not None else ()
        n = int(np.prod(shape)) if shape else 1
        item = self._dt.dtype.itemsize if not self._dt.vlen_string else 4 + f.so + 4
        if ver in (3, 4):                            # (version 4 differs from 3 only in its chunked layouts)
            cls = f.buf[p + 1]
            if cls == 0:                              # compact
                size = f._u(p + 2, 2)
                raw = f.buf[p + 4:p + 4 + size]
            elif cls == 1:                            # contiguous
                a, size = f._addr(p + 2), f._len(p + 2 + f.so)
                if a in (UNDEF, UNDEF >> (64 - 8 * f.so)):
                    raw = bytes(n * item)             # never written: fill value (zeros)
                else:
                    raw = f.buf[f.base_addr + a:f.base_addr + a + n * item]
            elif cls == 2 and ver == 3:               # chunked, v1 B-tree
                if self._filtered:
                    raise HDF5Error("%s: filtered (compressed) chunks are not supported" % self.name)
                rank = f.buf[p + 2] - 1
                bt = f._addr(p + 3)
                cdims = tuple(f._u(p + 3 + f.so + 4 * k, 4) for k in range(rank))
                return self._read_chunked(bt, cdims, shape)
            else:
                raise HDF5Error("%s: layout class %d" % (self.name, cls))
        elif ver in (1, 2):
            rank, cls = f.buf[p + 1], f.buf[p + 2]
            q = p + 8
            if cls == 1:
                a = f._addr(q)
                raw = f.buf[f.base_addr + a:f.base_addr + a + n * item]
            elif cls == 0:
                q += 4 * rank
                size = f._u(q, 4)
                raw = f.buf[q + 4:q + 4 + size]
            else:
                raise HDF5Error("%s: chunked layout of message version %d" % (self.name, ver))
        else:
            raise HDF5Error("%s: layout message version %d (libver='latest' files) is not supported" % (self.name, ver))
        return np.array(f.decode(self._dt, shape, raw))

    def _read_chunked(self, btree, cdims, shape):
        f = self.f
        out = np.zeros(shape, self._dt.dtype)
        rank = len(shape)

        def walk(addr):
            a = f.base_addr + addr
            if f.buf[a:a + 4] != b"TREE" or f.buf[a + 4] != 1:
                raise HDF5Error("bad chunk B-tree node")
            level, used = f.buf[a + 5], f._u(a + 6, 2)
            p = a + 8 + 2 * f.so
            keysz = 8 + 8 * (rank + 1)
            for k in range(used):
                csize = f._u(p, 4)
                offs = tuple(f._u(p + 8 + 8 * d, 8) for d in range(rank))
                child = f._addr(p + keysz)
                if level:
                    walk(child)
                else:
                    raw = f.buf[f.base_addr + child:f.base_addr + child + csize]
                    chunk = np.frombuffer(raw, self._dt.dtype, count=int(np.prod(cdims))).reshape(cdims)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                    out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
                p += keysz + f.so
        if btree not in (UNDEF, UNDEF >> (64 - 8 * f.so)):
            walk(btree)
        return out


class Group(object):
    def __init__(self, f, addr, name):
        self.f, self.addr, self.name = f, addr, name
        self._links = None

    @property
    def attrs(self):
        return self.f.attributes(self.addr)

    def _load(self):
        if self._links is not None:
            return
        f = self.f
        links = OrderedDict()
        for t, fl, p, sz in f.messages(self.addr):
            if t == 0x11:                             # symbol table: v1 B-tree + local heap
                bt, heap = f._addr(p), f._addr(p + f.so)
                h = f.base_addr + heap
                if f.buf[h:h + 4] != b"HEAP":
                    raise HDF5Error("bad local heap")
                data = f.base_addr + f._addr(h + 8 + 2 * f.sl)

                def walk(addr):
                    a = f.base_addr + addr
                    if f.buf[a:a + 4] == b"SNOD":
                        nsym = f._u(a + 6, 2)
                        q = a + 8
                        for k in range(nsym):
                            noff, oaddr = f._addr(q), f._addr(q + f.so)
                            s = data + noff
                            e = f.buf.index(b"\0", s)
                            if f._u(q + 2 * f.so, 4) != 2:           # cache type 2 = symbolic link (no object header): skipped
                                links[f.buf[s:e].decode("utf8")] = oaddr
                            q += 2 * f.so + 4 + 4 + 16
                        return
                    if f.buf[a:a + 4] != b"TREE" or f.buf[a + 4] != 0:
                        raise HDF5Error("bad group B-tree node")
                    used = f._u(a + 6, 2)
                    q = a + 8 + 2 * f.so + f.sl       # past the first key
                    for k in range(used):
                        walk(f._addr(q))
                        q += f.so + f.sl
                walk(bt)
            elif t == 0x06:                           # link message (compact new-style group)
                ver, flags = f.buf[p], f.buf[p + 1]
                q = p + 2
                ltype = 0
                if flags & 0x08:
                    ltype = f.buf[q]; q += 1
                if flags & 0x04:
                    q += 8
                if flags & 0x10:
                    q += 1
                lsz = 1 << (flags & 3)
                ln = f._u(q, lsz); q += lsz
                name = f.buf[q:q + ln].decode("utf8"); q += ln
                if ltype == 0:
                    links[name] = f._addr(q)
            elif t == 0x02:                           # link info: dense storage?
                flags = f.buf[p + 1]
                q = p + 2 + (8 if flags & 1 else 0)
                if f._addr(q) not in (UNDEF, UNDEF >> (64 - 8 * f.so)):
                    raise HDF5Error("dense link storage (libver='latest' files) is not supported")
        self._links = links

    def keys(self):
        self._load()
        return list(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [s for s in path.split("/") if s]:
            if not isinstance(node, Group):
                raise KeyError(path)
            node._load()
            if part not in node._links:
                raise KeyError(path)
            addr = node._links[part]
            kinds = set(t for t, _, _, _ in node.f.messages(addr))
            child_name = node.name.rstrip("/") + "/" + part
            node = Dataset(node.f, addr, child_name) if (0x08 in kinds and 0x03 in kinds) else Group(node.f, addr, child_name)
        return node


def read_keras_weights(src):
    """OrderedDict "<layer>/<variable>" -> ndarray of a Keras weight file (model.save_weights) or a full model file (model.save: the
    weights sit under /model_weights), in the order of the `layer_names` / `weight_names` attributes (= Keras graph order); kernels stay
    in Keras layout (k..., cin, cout).  Same result as keras_h5_to_npz with h5py (models/pretrained.py)."""
    f = File(src)
    g = f.root["model_weights"] if "model_weights" in f.root else f.root
    text = lambda n: n.decode("utf8") if isinstance(n, (bytes, np.bytes_)) else str(n)
    out = OrderedDict()
    attrs = g.attrs
    names = attrs.get("layer_names")
    if names is None:                                 # Keras splits very long name lists over layer_names0, layer_names1, ...
        names, k = [], 0
        while "layer_names%d" % k in attrs:
            names += list(np.atleast_1d(attrs["layer_names%d" % k])); k += 1
    for ln in map(text, np.atleast_1d(names)):
        lg = g[ln]
        la = lg.attrs
        wn = la.get("weight_names")
        if wn is None:
            wn, k = [], 0
            while "weight_names%d" % k in la:
                wn += list(np.atleast_1d(la["weight_names%d" % k])); k += 1
        for w in map(text, np.atleast_1d(wn)):
            out[w if w.startswith(ln) else ln + "/" + w.split("/")[-1]] = np.asarray(lg[w].read())
    return out
