"""A minimal, dependency-free reader for the HDF5 files Keras writes with ``save_weights`` / ``save``.

The reference loads its pretrained weights with ``model.load_weights(path)`` (detection.py:415-416,
recognition.py:383-404), i.e. through h5py.  h5py is not a dependency of this package (it is not even
installable in the build image's main interpreter), so the default construction path
``Pipeline() -> Detector('clovaai_general') / Recognizer('kurapan')`` reads the ``.h5`` files with this
module instead.  It understands exactly what such files contain (HDF5 File Format Specification 1.x
structures, which h5py/libhdf5 emit with the default ``libver='earliest'``):

  superblock version 0 / 1, old-style groups (symbol-table message -> v1 B-tree -> SNOD symbol nodes ->
  local heap names), version-1 object headers with continuation blocks, simple dataspaces, fixed-size
  little/big-endian IEEE floats and integers, contiguous or compact dataset layout.

Anything else (chunked / compressed datasets, new-style link messages, superblock v2+) raises
``NotImplementedError`` naming the feature.  Attributes are not needed: datasets are found by walking
the group tree, and their HDF5 paths carry the Keras layer and variable names
(``<layer>/<scope...>/<variable>:0``).

``read_datasets(path) -> {"group/sub/name": ndarray}``
"""

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _File:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != _SIG:
            raise ValueError("not an HDF5 file (bad signature)")
        ver = buf[8]
        if ver not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {ver} (written with libver='latest'?) is not supported")
        self.O, self.L = buf[13], buf[14]  # size of offsets / lengths
        if self.O != 8 or self.L != 8:
            raise NotImplementedError("HDF5 files with offset/length sizes other than 8 bytes")
        p = 24 + (4 if ver == 1 else 0)
        self.base = self.u(p, 8)
        p += 4 * 8  # base, free-space, end-of-file, driver-info addresses
        # root group symbol-table entry
        self.root_header = self.u(p + 8, 8)

    def u(self, off, n):
        return int.from_bytes(self.b[off:off + n], "little")

    # ---- object headers ------------------------------------------------------------------------
    def messages(self, addr):
        """version-1 object header at `addr` -> list of (type, payload bytes, flags), continuations followed"""
        a = addr + self.base
        if self.b[a] != 1:
            if self.b[a:a + 4] == b"OHDR":
                raise NotImplementedError("version-2 object headers (libver='latest') are not supported")
            raise ValueError(f"bad object header version {self.b[a]} at {addr}")
        n_msgs = self.u(a + 2, 2)
        size = self.u(a + 8, 4)
        blocks = [(a + 16, size)]  # 12-byte prefix padded to 8-byte alignment
        out = []
        # every block is read to its end: the header's message count also counts NIL padding messages, and a
        # continuation block may be listed by the very last message the count allows
        while blocks:
            p, left = blocks.pop(0)
            end = p + left
            if end > len(self.b):
                raise ValueError("object header block runs past the end of the file")
            while p + 8 <= end:
                mtype, msize, flags = self.u(p, 2), self.u(p + 2, 2), self.b[p + 4]
                body = self.b[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:  # continuation: offset, length
                    blocks.append((int.from_bytes(body[:8], "little") + self.base, int.from_bytes(body[8:16], "little")))
                out.append((mtype, body, flags))
            if len(out) > 4 * max(n_msgs, 16):
                raise ValueError("object header lists far more messages than it declares (corrupt continuation chain?)")
        return out

    # ---- groups --------------------------------------------------------------------------------
    def heap_name(self, heap_addr, off):
        h = heap_addr + self.base
        if self.b[h:h + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        data = self.u(h + 8 + 2 * 8, 8) + self.base
        end = self.b.index(b"\x00", data + off)
        return self.b[data + off:end].decode("utf-8")

    def btree_entries(self, addr, heap_addr):
        """group B-tree (v1, node type 0) -> [(name, object header address)]"""
        a = addr + self.base
        if self.b[a:a + 4] != b"TREE":
            raise ValueError("bad B-tree signature")
        node_type, level, used = self.b[a + 4], self.b[a + 5], self.u(a + 6, 2)
        if node_type != 0:
            raise ValueError("expected a group B-tree node")
        p = a + 8 + 2 * 8  # siblings
        out = []
        for _ in range(used):
            p += 8  # key
            child = self.u(p, 8)
            p += 8
            if level > 0:
                out += self.btree_entries(child, heap_addr)
            else:
                s = child + self.base
                if self.b[s:s + 4] != b"SNOD":
                    raise ValueError("bad symbol node signature")
                n = self.u(s + 6, 2)
                q = s + 8
                for _ in range(n):
                    out.append((self.heap_name(heap_addr, self.u(q, 8)), self.u(q + 8, 8)))
                    q += 40
        return out

    # ---- datasets ------------------------------------------------------------------------------
    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        del ver
        bits0 = body[1]
        size = int.from_bytes(body[4:8], "little")
        order = ">" if (bits0 & 1) else "<"
        if cls == 1:
            return np.dtype(f"{order}f{size}")
        if cls == 0:
            signed = bool(bits0 & 0x08)
            return np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        raise NotImplementedError(f"HDF5 datatype class {cls} (only floats / integers appear in Keras weight files)")

    def _shape(self, body):
        ver, rank = body[0], body[1]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
        else:
            raise NotImplementedError(f"dataspace message version {ver}")
        return tuple(int.from_bytes(body[p + 8 * i:p + 8 * i + 8], "little") for i in range(rank))

    def dataset(self, msgs):
        shape = dtype = None
        raw = None
        for mtype, body, flags in msgs:
            if mtype in (0x0001, 0x0003, 0x0008) and (flags & 0x02):
                # the message body is a reference to a shared / committed message, not the message itself
                raise NotImplementedError("shared (committed) dataspace / datatype / layout messages are not supported")
            if mtype == 0x0001:
                shape = self._shape(body)
            elif mtype == 0x0003:
                dtype = self._dtype(body)
            elif mtype == 0x000B:
                raise NotImplementedError("filtered (compressed) datasets")
            elif mtype == 0x0008:
                ver = body[0]
                if ver == 3:
                    cls = body[1]
                    if cls == 1:
                        addr, size = int.from_bytes(body[2:10], "little"), int.from_bytes(body[10:18], "little")
                        raw = b"" if addr == _UNDEF else self.b[addr + self.base:addr + self.base + size]
                    elif cls == 0:
                        size = int.from_bytes(body[2:4], "little")
                        raw = body[4:4 + size]
                    else:
                        raise NotImplementedError("chunked dataset layout")
                elif ver in (1, 2):
                    ndim, cls = body[1], body[2]
                    if cls == 1:
                        addr = int.from_bytes(body[8:16], "little")
                        raw = ("addr", addr)
                    elif cls == 0:
                        p = 8 + 4 * ndim
                        size = int.from_bytes(body[p:p + 4], "little")
                        raw = body[p + 4:p + 4 + size]
                    else:
                        raise NotImplementedError("chunked dataset layout")
                else:
                    raise NotImplementedError(f"data layout message version {ver}")
        have = [shape is not None, dtype is not None, raw is not None]
        if not any(have):
            return None  # not a dataset (e.g. an empty group without a symbol table)
        if not all(have):
            missing = [n for n, h in zip(("dataspace", "datatype", "data layout"), have) if not h]
            raise NotImplementedError(f"dataset object without a readable {' / '.join(missing)} message")
        count = int(np.prod(shape)) if shape else 1
        if isinstance(raw, tuple):
            a = raw[1] + self.base
            raw = self.b[a:a + count * dtype.itemsize]
        if len(raw) < count * dtype.itemsize:
            return np.zeros(shape, dtype.newbyteorder("="))  # never written (fill value)
        return np.frombuffer(raw, dtype, count).reshape(shape).astype(dtype.newbyteorder("="))

    def walk(self, header_addr, prefix, out, depth=0):
        if depth > 32:
            raise ValueError("group nesting too deep (cycle?)")
        msgs = self.messages(header_addr)
        stab = [body for mtype, body, _ in msgs if mtype == 0x0011]
        if stab:
            btree, heap = int.from_bytes(stab[0][:8], "little"), int.from_bytes(stab[0][8:16], "little")
            for name, child in self.btree_entries(btree, heap):
                self.walk(child, f"{prefix}/{name}" if prefix else name, out, depth + 1)
            return
        if any(mtype in (0x0002, 0x0006) for mtype, _, _ in msgs):
            raise NotImplementedError("new-style (link message) groups are not supported")
        try:
            arr = self.dataset(msgs)
        except NotImplementedError as e:
            raise NotImplementedError(f"{prefix}: {e}") from None
        if arr is not None:
            out[prefix] = arr


def read_datasets(path):
    """All datasets of an HDF5 file as ``{"a/b/c": ndarray}`` (native byte order)."""
    with open(path, "rb") as f:
        buf = f.read()
    hf = _File(buf)
    out = {}
    hf.walk(hf.root_header, "", out)
    return out
