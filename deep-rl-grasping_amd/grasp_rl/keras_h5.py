"""Keras weight files (`model.h5`) without h5py.

The reference stores and loads its depth auto-encoder through Keras 2.2.4 ``save_weights`` /
``load_weights`` (/root/reference/manipulation_main/gripperEnv/encoders.py:27-31, 48) -- HDF5 files in
the "earliest" library format.  This module reads and writes exactly that subset of HDF5 from first
principles (HDF5 File Format Specification v1.1/2.0), so that the encoders shipped under
``encoder_files/`` feed ``grl_encode`` directly and auto-encoders trained on the device can be loaded
by the reference's own ``SimpleAutoEncoder.load_weights``:

* superblock version 0, 8-byte offsets / lengths;
* groups = version-1 object header with a symbol-table message -> v1 B-tree ("TREE") -> symbol-table
  nodes ("SNOD") -> names in a local heap ("HEAP");
* datasets = version-1 object header with dataspace (v1/v2), datatype (IEEE float / fixed-point /
  fixed-length string), layout v3 (contiguous or compact) messages; no chunking, no filters;
* attributes (message 0x000C, versions 1-3) with the same datatypes: Keras keeps ``layer_names`` /
  ``weight_names`` / ``backend`` / ``keras_version`` there.

The layout Keras gives a weight file (encoders.py model: nested ``encoder`` / ``decoder`` models):
``/<layer>`` groups listed in the root attribute ``layer_names``; each with ``weight_names`` and the
datasets ``/<layer>/<weight_name>`` (e.g. ``/encoder/conv2d_1/kernel:0``).
"""
import struct
from collections import OrderedDict

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF

MSG_DATASPACE, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LAYOUT = 0x0001, 0x0003, 0x0004, 0x0005, 0x0008
MSG_ATTRIBUTE, MSG_CONTINUATION, MSG_SYMTAB, MSG_MTIME = 0x000C, 0x0010, 0x0011, 0x0012


class H5FormatError(ValueError):
    pass


def _pad8(n):
    return (n + 7) & ~7


# ================================================================================================ reading
class Node:
    """A group (``children``: name -> Node) or a dataset (``data``: ndarray)."""

    def __init__(self):
        self.attrs = OrderedDict()
        self.children = None
        self.data = None
        self.messages = []          # (type, flags, raw bytes) in file order -- format introspection / tests

    @property
    def is_group(self):
        return self.children is not None

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not node.is_group or part not in node.children:
                raise KeyError(path)
            node = node.children[part]
        return node

    def visit(self, prefix=""):
        """Yield (path, node) for every descendant, groups before their members, names sorted."""
        for name in sorted(self.children or ()):
            child = self.children[name]
            path = prefix + name
            yield path, child
            if child.is_group:
                for item in child.visit(path + "/"):
                    yield item


class _Reader:
    def __init__(self, buf):
        self.buf = buf
        if buf[:8] != SIGNATURE:
            raise H5FormatError("not an HDF5 file (signature missing at offset 0; user blocks are not supported)")
        ver = buf[8]
        if ver not in (0, 1):
            raise H5FormatError("superblock version %d is not supported (Keras / h5py 'earliest' writes version 0)" % ver)
        if buf[13] != 8 or buf[14] != 8:
            raise H5FormatError("only 8-byte offsets and lengths are supported")
        pos = 24 + (4 if ver == 1 else 0)
        self.base, _free, self.eof, _drv = struct.unpack_from("<QQQQ", buf, pos)
        entry = pos + 32
        _name_off, self.root_header = struct.unpack_from("<QQ", buf, entry)

    # ------------------------------------------------------------------ object headers
    def messages(self, addr):
        b = self.buf
        addr += self.base
        if b[addr] != 1:
            raise H5FormatError("object header version %d at %d is not supported (version 1 expected)" % (b[addr], addr))
        n_msgs, _refs, size = struct.unpack_from("<HII", b, addr + 2)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < n_msgs:
            pos, length = blocks.pop(0)
            end = pos + length
            while pos + 8 <= end and len(out) < n_msgs:
                mtype, msize, flags = struct.unpack_from("<HHB", b, pos)
                data = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == MSG_CONTINUATION:
                    off, ln = struct.unpack_from("<QQ", data, 0)
                    blocks.append((off + self.base, ln))
                out.append((mtype, flags, bytes(data)))
        return out

    # ------------------------------------------------------------------ datatypes / dataspaces
    @staticmethod
    def parse_datatype(d):
        cls, version = d[0] & 0x0F, d[0] >> 4
        bits0 = d[1]
        size = struct.unpack_from("<I", d, 4)[0]
        order = ">" if (bits0 & 1) else "<"
        if cls == 0:                                   # fixed point
            signed = bool(bits0 & 0x08)
            return np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), 8 + 4
        if cls == 1:                                   # floating point (IEEE layouts only)
            if size not in (2, 4, 8):
                raise H5FormatError("floating-point size %d is not supported" % size)
            return np.dtype("%sf%d" % (order, size)), 8 + 12
        if cls == 3:                                   # fixed-length string
            return np.dtype("S%d" % size), 8
        if cls == 9 and (bits0 & 0x0F) == 1:            # variable-length string: elements are global-heap references
            return np.dtype([("len", "<u4"), ("addr", "<u8"), ("idx", "<u4")]), 8
        raise H5FormatError("datatype class %d (version %d) is not supported" % (cls, version))

    @staticmethod
    def parse_dataspace(d):
        version, rank, flags = d[0], d[1], d[2]
        if version == 1:
            pos = 8
        elif version == 2:
            if d[3] == 2:                               # null dataspace
                return None, 4
            pos = 4
        else:
            raise H5FormatError("dataspace version %d is not supported" % version)
        dims = struct.unpack_from("<%dQ" % rank, d, pos) if rank else ()
        used = pos + 8 * rank * (2 if (flags & 1) else 1)
        return tuple(int(x) for x in dims), used

    def parse_attribute(self, d):
        version = d[0]
        if version == 1:
            name_sz, dt_sz, ds_sz = struct.unpack_from("<HHH", d, 2)
            pos = 8
            step = _pad8
        elif version in (2, 3):
            name_sz, dt_sz, ds_sz = struct.unpack_from("<HHH", d, 2)
            pos = 8 + (1 if version == 3 else 0)
            step = lambda n: n                          # noqa: E731  (no padding from version 2 on)
        else:
            raise H5FormatError("attribute version %d is not supported" % version)
        name = d[pos:pos + name_sz].split(b"\0")[0].decode()
        pos += step(name_sz)
        dtype, _ = self.parse_datatype(d[pos:pos + dt_sz])
        pos += step(dt_sz)
        shape, _ = self.parse_dataspace(d[pos:pos + ds_sz])
        pos += step(ds_sz)
        if shape is None:
            return name, np.zeros((0,), dtype)
        count = int(np.prod(shape)) if shape else 1
        arr = np.frombuffer(d, dtype=dtype, count=count, offset=pos).reshape(shape).copy()
        if dtype.names:                                 # variable-length strings (h5py stores Python bytes / str so)
            flat = [self.global_heap_object(int(r["addr"]), int(r["idx"]))[:int(r["len"])] for r in arr.ravel()]
            arr = np.array(flat, dtype="S").reshape(shape)
        return name, arr

    def global_heap_object(self, addr, index):
        b = self.buf
        a = addr + self.base
        if b[a:a + 4] != b"GCOL":
            raise H5FormatError("global heap collection signature missing at %d" % a)
        size = struct.unpack_from("<Q", b, a + 8)[0]
        pos, end = a + 16, a + size
        while pos + 16 <= end:
            idx, _ref, osize = struct.unpack_from("<HH4xQ", b, pos)
            if idx == 0:
                break
            if idx == index:
                return bytes(b[pos + 16:pos + 16 + osize])
            pos += 16 + _pad8(osize)
        raise H5FormatError("object %d not found in the global heap collection at %d" % (index, a))

    # ------------------------------------------------------------------ groups
    def heap_name(self, heap_addr, offset):
        b = self.buf
        heap_addr += self.base
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5FormatError("local heap signature missing at %d" % heap_addr)
        data_addr = struct.unpack_from("<Q", b, heap_addr + 24)[0] + self.base
        end = b.index(b"\0", data_addr + offset)
        return b[data_addr + offset:end].decode()

    def group_entries(self, btree_addr, heap_addr):
        """[(name, object header address)] of a symbol-table group, in B-tree (= name) order."""
        b = self.buf
        a = btree_addr + self.base
        if b[a:a + 4] != b"TREE":
            raise H5FormatError("B-tree signature missing at %d" % a)
        node_type, level, used = struct.unpack_from("<BBH", b, a + 4)
        if node_type != 0:
            raise H5FormatError("B-tree node type %d where a group node was expected" % node_type)
        out = []
        pos = a + 24
        for k in range(used):
            child = struct.unpack_from("<Q", b, pos + 8)[0]
            pos += 16
            if level > 0:
                out.extend(self.group_entries(child, heap_addr))
                continue
            s = child + self.base
            if b[s:s + 4] != b"SNOD":
                raise H5FormatError("symbol-table node signature missing at %d" % s)
            n = struct.unpack_from("<H", b, s + 6)[0]
            for e in range(n):
                name_off, hdr = struct.unpack_from("<QQ", b, s + 8 + 40 * e)
                out.append((self.heap_name(heap_addr, name_off), hdr))
        return out

    # ------------------------------------------------------------------ objects
    def load(self, header_addr, depth=0):
        if depth > 32:
            raise H5FormatError("group nesting deeper than 32 (cycle?)")
        node = Node()
        msgs = self.messages(header_addr)
        node.messages = msgs
        shape = dtype = layout = None
        for mtype, _flags, d in msgs:
            if mtype == MSG_SYMTAB:
                btree, heap = struct.unpack_from("<QQ", d, 0)
                node.children = OrderedDict()
                for name, hdr in self.group_entries(btree, heap):
                    node.children[name] = self.load(hdr, depth + 1)
            elif mtype == MSG_DATASPACE:
                shape, _ = self.parse_dataspace(d)
            elif mtype == MSG_DATATYPE:
                dtype, _ = self.parse_datatype(d)
            elif mtype == MSG_LAYOUT:
                layout = d
            elif mtype == MSG_ATTRIBUTE:
                name, arr = self.parse_attribute(d)
                node.attrs[name] = arr
        if node.children is None:
            if dtype is None or layout is None:
                raise H5FormatError("object at %d is neither a symbol-table group nor a dataset" % header_addr)
            if layout[0] != 3:
                raise H5FormatError("data layout message version %d is not supported" % layout[0])
            count = int(np.prod(shape)) if shape else (0 if shape is None else 1)
            if layout[1] == 1:                          # contiguous
                addr, size = struct.unpack_from("<QQ", layout, 2)
                if addr == UNDEF:
                    node.data = np.zeros(shape or (), dtype)
                else:
                    if size < count * dtype.itemsize or addr + self.base + size > len(self.buf):
                        raise H5FormatError("dataset storage outside the file")
                    node.data = np.frombuffer(self.buf, dtype=dtype, count=count, offset=addr + self.base).reshape(shape or ()).copy()
            elif layout[1] == 0:                        # compact
                size = struct.unpack_from("<H", layout, 2)[0]
                node.data = np.frombuffer(layout[4:4 + size], dtype=dtype, count=count).reshape(shape or ()).copy()
            else:
                raise H5FormatError("chunked datasets are not supported (Keras weight files are contiguous)")
        return node


def read_h5(path):
    """Parse an HDF5 file of the subset above; returns the root ``Node``."""
    with open(path, "rb") as f:
        buf = f.read()
    r = _Reader(buf)
    return r.load(r.root_header)


def _names(arr):
    return [x.decode() if isinstance(x, bytes) else str(x) for x in np.asarray(arr).ravel()]


def read_keras_weights(path):
    """OrderedDict ``"<layer>/<weight_name>" -> ndarray`` in Keras' own order (root ``layer_names``, then each
    layer group's ``weight_names``), e.g. ``"encoder/conv2d_1/kernel:0"``."""
    root = read_h5(path)
    if "layer_names" not in root.attrs:
        raise H5FormatError("%s has no 'layer_names' attribute: not a Keras weight file" % path)
    out = OrderedDict()
    for layer in _names(root.attrs["layer_names"]):
        g = root[layer]
        for w in _names(g.attrs.get("weight_names", [])):
            out[layer + "/" + w] = np.asarray(g[w].data)
    return out


# ================================================================================================ writing
class _Writer:
    """Append-only builder; every structure is 8-byte aligned.  Group B-trees get one level-0 node with up to
    2*K_INTERNAL symbol-table nodes of up to 2*K_LEAF entries -- 256 members per group."""
    K_LEAF, K_INTERNAL = 4, 16

    def __init__(self):
        self.buf = bytearray(96)                       # superblock, patched at the end

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    # ------------------------------------------------------------------ messages
    @staticmethod
    def msg(mtype, data, flags=0):
        data = bytes(data) + b"\0" * (_pad8(len(data)) - len(data))
        return struct.pack("<HHB3x", mtype, len(data), flags) + data

    @staticmethod
    def datatype(dtype):
        dtype = np.dtype(dtype)
        if dtype.kind == "f":
            size = dtype.itemsize
            exp_bits, man_bits, bias = {2: (5, 10, 15), 4: (8, 23, 127), 8: (11, 52, 1023)}[size]
            return struct.pack("<BBBBI", 0x11, 0x20, 8 * size - 1, 0, size) + \
                struct.pack("<HHBBBBI", 0, 8 * size, man_bits, exp_bits, 0, man_bits, bias)
        if dtype.kind in "iu":
            return struct.pack("<BBBBI", 0x10, 0x08 if dtype.kind == "i" else 0, 0, 0, dtype.itemsize) + \
                struct.pack("<HH", 0, 8 * dtype.itemsize)
        if dtype.kind == "S":
            return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, max(1, dtype.itemsize))   # null-padded ASCII (h5py's numpy 'S')
        raise TypeError("dtype %s cannot be stored" % dtype)

    @staticmethod
    def dataspace(shape):
        shape = tuple(shape)
        body = struct.pack("<BBB5x", 1, len(shape), 1 if shape else 0)
        body += b"".join(struct.pack("<Q", n) for n in shape)
        body += b"".join(struct.pack("<Q", n) for n in shape)     # maximum dimensions = current (as h5py writes)
        return body

    def attribute(self, name, value):
        value = np.asarray(value)
        if value.dtype.kind == "U":
            value = np.char.encode(value, "ascii")
        if value.dtype.kind == "f" and value.size == 0:
            value = value.astype("<f8")
        le = value.dtype.newbyteorder("<") if value.dtype.kind in "fiu" else value.dtype
        nm = name.encode() + b"\0"
        dt, ds = self.datatype(le), self.dataspace(value.shape)
        body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds))
        for part in (nm, dt, ds):
            body += part + b"\0" * (_pad8(len(part)) - len(part))
        body += np.ascontiguousarray(value.astype(le)).tobytes()
        return self.msg(MSG_ATTRIBUTE, body)

    def object_header(self, messages):
        body = b"".join(messages)
        return self.alloc(struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body)

    # ------------------------------------------------------------------ objects
    def dataset(self, arr, attrs=None):
        arr = np.asarray(arr)
        le = arr.dtype.newbyteorder("<") if arr.dtype.kind in "fiu" else arr.dtype
        raw = np.ascontiguousarray(arr.astype(le)).tobytes()
        addr = self.alloc(raw) if raw else UNDEF
        msgs = [self.msg(MSG_DATASPACE, self.dataspace(arr.shape)),
                self.msg(MSG_DATATYPE, self.datatype(le), flags=1),
                self.msg(MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1),   # as h5py: v2, late allocation, default fill
                self.msg(MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, addr, len(raw)))]
        for k, v in (attrs or {}).items():
            msgs.append(self.attribute(k, v))
        return self.object_header(msgs)

    def group(self, members, attrs=None):
        """members: {name: object header address}.  Returns (header address, btree address, heap address)."""
        names = sorted(members, key=lambda s: s.encode())
        # local heap: offset 0 = empty string (8 bytes), then the names, then one free block to the end
        seg = bytearray(8)
        offs = {}
        for n in names:
            offs[n] = len(seg)
            e = n.encode() + b"\0"
            seg += e + b"\0" * (_pad8(len(e)) - len(e))
        free_at = len(seg)
        seg += struct.pack("<QQ", 1, 16)               # free block: next = H5HL_FREE_NULL (1), size 16
        heap_data = self.alloc(bytes(seg))
        heap = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), free_at, heap_data))
        # symbol-table nodes
        per = 2 * self.K_LEAF
        chunks = [names[i:i + per] for i in range(0, len(names), per)] or [[]]
        if len(chunks) > 2 * self.K_INTERNAL:
            raise ValueError("more than %d members in one group" % (per * 2 * self.K_INTERNAL))
        snods = []
        for ch in chunks:
            body = b"SNOD" + struct.pack("<BxH", 1, len(ch))
            for n in ch:
                body += struct.pack("<QQII16x", offs[n], members[n], 0, 0)
            body += b"\0" * (8 + 40 * per - len(body))
            snods.append(self.alloc(body))
        # B-tree: key[0] = 0 (empty name), key[i+1] = heap offset of the largest name in child i
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(chunks) if names else 0, UNDEF, UNDEF)
        tree += struct.pack("<Q", 0)
        for ch, sn in zip(chunks, snods):
            tree += struct.pack("<QQ", sn, offs[ch[-1]] if ch else 0)
        tree += b"\0" * (24 + 8 * (2 * 2 * self.K_INTERNAL + 1) - len(tree))
        btree = self.alloc(tree)
        msgs = [self.msg(MSG_SYMTAB, struct.pack("<QQ", btree, heap))]
        for k, v in (attrs or {}).items():
            msgs.append(self.attribute(k, v))
        return self.object_header(msgs), btree, heap

    def finish(self, root_header, root_btree, root_heap):
        while len(self.buf) % 8:
            self.buf.append(0)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.K_LEAF, self.K_INTERNAL, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, root_header, 1, 0) + struct.pack("<QQ", root_btree, root_heap)
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def _build(w, tree):
    """tree: {"attrs": {...}, "members": {name: subtree | ndarray}} -> header address (+ btree, heap for groups)."""
    members = OrderedDict()
    for name, sub in tree.get("members", {}).items():
        if isinstance(sub, dict):
            members[name] = _build(w, sub)[0]
        else:
            members[name] = w.dataset(sub)
    return w.group(members, tree.get("attrs"))


def write_h5(path, tree):
    """Write a nested ``{"attrs": {...}, "members": {name: subtree | ndarray}}`` description."""
    w = _Writer()
    hdr, btree, heap = _build(w, tree)
    data = w.finish(hdr, btree, heap)
    with open(path, "wb") as f:
        f.write(data)


def write_keras_weights(path, weights, backend="tensorflow", keras_version="2.2.4", extra_layers=()):
    """``weights``: OrderedDict ``"<layer>/<weight_name>" -> ndarray`` (the form ``read_keras_weights`` returns).
    ``extra_layers``: weight-less layers to list first (Keras lists e.g. ``input_1``)."""
    layers = OrderedDict((n, OrderedDict()) for n in extra_layers)
    for key, arr in weights.items():
        layer, wname = key.split("/", 1)
        layers.setdefault(layer, OrderedDict())[wname] = np.asarray(arr, np.float32)
    root = {"attrs": OrderedDict(), "members": OrderedDict()}
    root["attrs"]["backend"] = np.bytes_(backend)
    root["attrs"]["keras_version"] = np.bytes_(keras_version)
    root["attrs"]["layer_names"] = np.array([n.encode() for n in layers], dtype="S")
    for layer, ws in layers.items():
        g = {"attrs": OrderedDict(), "members": OrderedDict()}
        g["attrs"]["weight_names"] = np.array([n.encode() for n in ws], dtype="S") if ws else np.zeros((0,), np.float64)
        for wname, arr in ws.items():                   # "conv2d_1/kernel:0" -> nested group conv2d_1
            node = g
            parts = wname.split("/")
            for p in parts[:-1]:
                node = node["members"].setdefault(p, {"attrs": OrderedDict(), "members": OrderedDict()})
            node["members"][parts[-1]] = arr
        root["members"][layer] = g
    write_h5(path, root)
