"""A minimal pure-Python HDF5 reader: just enough of the format to read Keras ``save_weights`` files
(reference: cchess_alphazero/agent/model.py:95-101 ``load_weights``; h5py is not available in this image).

Supported (what libhdf5 writes with default settings, i.e. what h5py / Keras 2.0.8 produce):
  * superblock version 0 / 1, 8-byte offsets and lengths
  * old-style groups: symbol-table message -> v1 B-tree -> symbol nodes, names in a local heap
  * version-1 object headers with continuation blocks
  * datasets with contiguous or compact layout; little-endian IEEE floats and integers, fixed-length strings
  * attributes (message versions 1-3) of those types, scalar or array, including empty arrays
Not supported (raises ``Hdf5Error``): chunked / compressed datasets, new-style groups, variable-length data.

    f = File(path); f.attrs["layer_names"]; g = f["conv1"]; g.attrs["weight_names"]; g["conv1/kernel:0"][...] -> ndarray
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


class _Datatype:
    def __init__(self, buf):
        cls_ver, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, 0)
        self.cls = cls_ver & 0x0F
        self.size = size
        self.vlen_str = False
        big = b0 & 1
        if self.cls == 0:                                  # fixed point
            signed = (b0 >> 3) & 1
            self.np = np.dtype(("<", ">")[big] + ("u", "i")[signed] + str(size))
        elif self.cls == 1:                                # floating point
            if size not in (2, 4, 8):
                raise Hdf5Error(f"unsupported float size {size}")
            self.np = np.dtype(("<", ">")[big] + "f" + str(size))
        elif self.cls == 3:                                # fixed-length string
            self.np = np.dtype("S" + str(size))
        elif self.cls == 9 and (b0 & 0x0F) == 1:           # variable-length string: 16-byte global-heap references
            self.np = np.dtype([("len", "<u4"), ("addr", "<u8"), ("idx", "<u4")])
            self.vlen_str = True
        else:
            raise Hdf5Error(f"unsupported HDF5 datatype class {self.cls} (variable-length sequences / compound data)")


def _dataspace(buf):
    ver, rank, flags = struct.unpack_from("<BBB", buf, 0)
    if ver == 1:
        off = 8
    elif ver == 2:
        if buf[3] == 2:                                    # null dataspace
            return None
        off = 4
    else:
        raise Hdf5Error(f"unsupported dataspace version {ver}")
    return tuple(struct.unpack_from("<" + "Q" * rank, buf, off)) if rank else ()


def _pad8(n):
    return (n + 7) & ~7


class _Object:
    """One object header: its messages decoded into datatype / dataspace / layout / attributes / symbol table."""

    def __init__(self, f, addr):
        self.f = f
        self.addr = addr
        self.attrs = {}
        self.dtype = self.shape = self.layout = self.symtab = None
        buf = f.buf
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", buf, addr)
        if ver != 1:
            raise Hdf5Error(f"unsupported object header version {ver} (new-style file: written with libver='latest'?)")
        blocks = [(addr + 16, hsize)]
        seen = 0
        while blocks and seen < nmsg:
            pos, length = blocks.pop(0)
            end = pos + length
            while pos + 8 <= end and seen < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", buf, pos)
                body = buf[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                seen += 1
                if mtype == 0x0010:                        # continuation
                    blocks.append(struct.unpack_from("<QQ", body, 0))
                elif mtype == 0x0001:
                    self.shape = _dataspace(body)
                elif mtype == 0x0003:
                    self.dtype = _Datatype(body)
                elif mtype == 0x0008:
                    self.layout = self._layout(body)
                elif mtype == 0x000C:
                    name, value = self._attribute(body)
                    self.attrs[name] = value
                elif mtype == 0x0011:
                    self.symtab = struct.unpack_from("<QQ", body, 0)
                elif mtype in (0x0002, 0x0006):
                    raise Hdf5Error("new-style group (link messages): write the file with the default libver")

    @staticmethod
    def _layout(body):
        ver, cls = body[0], body[1]
        if ver != 3:
            raise Hdf5Error(f"unsupported data layout version {ver}")
        if cls == 0:
            (size,) = struct.unpack_from("<H", body, 2)
            return ("compact", bytes(body[4:4 + size]))
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", body, 2)
            return ("contiguous", addr, size)
        raise Hdf5Error("chunked / compressed datasets are not supported (Keras save_weights writes contiguous data)")

    def _attribute(self, body):
        ver = body[0]
        nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
        if ver == 1:
            pos = 8
            name = bytes(body[pos:pos + nsz]); pos += _pad8(nsz)
            dt = _Datatype(body[pos:pos + tsz]); pos += _pad8(tsz)
            shape = _dataspace(body[pos:pos + ssz]); pos += _pad8(ssz)
        elif ver in (2, 3):
            pos = 8 if ver == 2 else 9
            name = bytes(body[pos:pos + nsz]); pos += nsz
            dt = _Datatype(body[pos:pos + tsz]); pos += tsz
            shape = _dataspace(body[pos:pos + ssz]); pos += ssz
        else:
            raise Hdf5Error(f"unsupported attribute message version {ver}")
        name = name.split(b"\0", 1)[0].decode("utf8")
        if shape is None:
            return name, None
        count = int(np.prod(shape)) if shape else 1
        arr = np.frombuffer(body, dtype=dt.np, count=count, offset=pos).reshape(shape)
        if dt.vlen_str:
            vals = [self.f.global_heap_object(int(r["addr"]), int(r["idx"]))[:int(r["len"])] for r in arr.reshape(-1)]
            out = np.empty(len(vals), dtype=object)
            out[:] = vals
            return name, (out.reshape(shape) if shape else vals[0])
        return name, (arr.copy() if shape else arr.reshape(()).copy()[()])

    # ---- dataset ----
    def read(self):
        if self.dtype is None or self.layout is None or self.shape is None:
            raise Hdf5Error("not a dataset")
        count = int(np.prod(self.shape)) if self.shape else 1
        if self.layout[0] == "compact":
            raw = self.layout[1]
            arr = np.frombuffer(raw, dtype=self.dtype.np, count=count)
        else:
            _, addr, size = self.layout
            if count == 0 or addr == UNDEF:
                return np.zeros(self.shape, dtype=self.dtype.np)
            arr = np.frombuffer(self.f.buf, dtype=self.dtype.np, count=count, offset=addr + self.f.base)
        return arr.reshape(self.shape).copy()

    # ---- group ----
    def children(self):
        """name -> object header address (old-style group)"""
        if self.symtab is None:
            return {}
        btree, heap = self.symtab
        buf = self.f.buf
        if buf[heap:heap + 4] != b"HEAP":
            raise Hdf5Error("bad local heap signature")
        (data_addr,) = struct.unpack_from("<Q", buf, heap + 24)
        out = {}

        def name_at(off):
            p = data_addr + off
            return bytes(buf[p:buf.index(b"\0", p)]).decode("utf8")

        def walk(node):
            if buf[node:node + 4] == b"SNOD":
                (nsym,) = struct.unpack_from("<H", buf, node + 6)
                for i in range(nsym):
                    name_off, obj = struct.unpack_from("<QQ", buf, node + 8 + 40 * i)
                    out[name_at(name_off)] = obj
                return
            if buf[node:node + 4] != b"TREE":
                raise Hdf5Error("bad B-tree signature")
            ntype, level, used = struct.unpack_from("<BBH", buf, node + 4)
            if ntype != 0:
                raise Hdf5Error("unexpected B-tree node type in a group")
            pos = node + 24
            for i in range(used):
                (child,) = struct.unpack_from("<Q", buf, pos + 8)      # key (8) then child pointer (8)
                walk(child)
                pos += 16

        walk(btree)
        return out


class Node:
    """Group or dataset handle with the small h5py-like surface the importer uses."""

    def __init__(self, f, obj, name):
        self._f, self._o, self.name = f, obj, name
        self.attrs = obj.attrs

    @property
    def is_dataset(self):
        return self._o.layout is not None

    @property
    def shape(self):
        return self._o.shape

    def keys(self):
        return sorted(self._o.children())

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, key):
        if key is Ellipsis or key == ():
            return self._o.read()
        node = self
        for part in [p for p in key.split("/") if p]:
            kids = node._o.children()
            if part not in kids:
                raise KeyError(f"{key!r}: no {part!r} in {node.name!r}")
            node = Node(self._f, _Object(self._f, kids[part] + self._f.base), node.name.rstrip("/") + "/" + part)
        return node

    def read(self):
        return self._o.read()


class File(Node):
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        if self.buf[:8] != SIGNATURE:
            raise Hdf5Error(f"{path}: not an HDF5 file (a user block before the superblock is not supported)")
        ver = self.buf[8]
        if ver not in (0, 1):
            raise Hdf5Error(f"unsupported superblock version {ver} (written with libver='latest'?)")
        if self.buf[13] != 8 or self.buf[14] != 8:
            raise Hdf5Error("only 8-byte offsets / lengths are supported")
        pos = 24 + (4 if ver == 1 else 0)
        self.base, _, _, _ = struct.unpack_from("<QQQQ", self.buf, pos)
        if self.base != 0:
            raise Hdf5Error("non-zero base address is not supported")
        root_entry = pos + 32
        (root_addr,) = struct.unpack_from("<Q", self.buf, root_entry + 8)
        super().__init__(self, _Object(self, root_addr + self.base), "/")

    def global_heap_object(self, addr, idx):
        """bytes of object `idx` of the global heap collection at `addr` (variable-length strings live there)"""
        buf = self.buf
        if buf[addr:addr + 4] != b"GCOL":
            raise Hdf5Error("bad global heap signature")
        (size,) = struct.unpack_from("<Q", buf, addr + 8)
        pos, end = addr + 16, addr + size
        while pos + 16 <= end:
            oidx, _, _, osize = struct.unpack_from("<HHIQ", buf, pos)
            if oidx == 0:
                break
            if oidx == idx:
                return bytes(buf[pos + 16:pos + 16 + osize])
            pos += 16 + _pad8(osize)
        raise Hdf5Error(f"global heap object {idx} not found")

    def close(self):
        self.buf = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
