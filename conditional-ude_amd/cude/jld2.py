"""Minimal JLD2 (HDF5-subset) reader and writer for the checkpoints and data files either side of the path.

The reference stores trained parameters and data sets with JLD2.jl (`jldsave`/`jldopen`, e.g.
c-peptide/02-conditional.jl:56-64 `source_data/cude_neural_parameters.jld2`, suppression/suppression.jl:93-104
`results/lambda=*.jld2`, `data/ohashi.jld2`).  JLD2.jl is a third-party package (not under the reference tree);
its files are a subset of HDF5: a 512-byte text header, a version-2 superblock, version-2 object headers
(`OHDR`) with link / dataspace / datatype / layout messages, contiguous or compact data, object references for
arrays of arrays and committed compound datatypes for Julia structs.  This module restates exactly that subset
from the published HDF5 file-format specification (version 3.0) -- enough to read every `.jld2` file the
reference ships and to write parameter checkpoints JLD2.jl's layout conventions are followed for.

Python in, numpy out: scalars -> Python numbers, `Array{Float64,N}` -> ndarray with Julia's column-major shape
(dimensions reversed back to Julia order), `Vector{Vector{Float64}}` -> list of ndarrays, `String` -> str,
structs / NamedTuples -> dict of fields.
"""
import struct

import numpy as np

HEADER_BYTES = 512
_SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------- checksum (Jenkins lookup3)
def _rot(x, k):
    return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF


def lookup3(data, init=0):
    """Bob Jenkins' lookup3 `hashlittle`, the checksum of HDF5 version-2 metadata blocks."""
    n = len(data)
    a = b = c = (0xDEADBEEF + n + init) & 0xFFFFFFFF
    M = 0xFFFFFFFF
    i = 0
    while n - i > 12:
        a = (a + int.from_bytes(data[i:i + 4], "little")) & M
        b = (b + int.from_bytes(data[i + 4:i + 8], "little")) & M
        c = (c + int.from_bytes(data[i + 8:i + 12], "little")) & M
        a = (a - c) & M; a ^= _rot(c, 4); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 6); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 8); b = (b + a) & M
        a = (a - c) & M; a ^= _rot(c, 16); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 19); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 4); b = (b + a) & M
        i += 12
    tail = data[i:]
    if len(tail) == 0:
        return c
    tail = tail + b"\x00" * (12 - len(tail))
    a = (a + int.from_bytes(tail[0:4], "little")) & M
    b = (b + int.from_bytes(tail[4:8], "little")) & M
    c = (c + int.from_bytes(tail[8:12], "little")) & M
    c ^= b; c = (c - _rot(b, 14)) & M
    a ^= c; a = (a - _rot(c, 11)) & M
    b ^= a; b = (b - _rot(a, 25)) & M
    c ^= b; c = (c - _rot(b, 16)) & M
    a ^= c; a = (a - _rot(c, 4)) & M
    b ^= a; b = (b - _rot(a, 14)) & M
    c ^= b; c = (c - _rot(b, 24)) & M
    return c


# ----------------------------------------------------------------------------- reader
class _Datatype:
    """Parsed HDF5 datatype message."""

    def __init__(self, cls, size, **kw):
        self.cls, self.size = cls, size
        self.__dict__.update(kw)


class JLD2File:
    """Read-only view of a JLD2 file: `f = JLD2File(path); f.keys(); f["name"]`."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if not b.startswith(b"HDF5-based Julia Data Format"):
            raise ValueError("not a JLD2 file (missing text header)")
        note = b[:HEADER_BYTES].split(b"\x00")[1].decode("ascii", "replace")      # " (Julia 1.11.1 64-bit LE)"
        self.julia_version = note.split("Julia ")[1].split(" ")[0] if "Julia " in note else None
        sb = HEADER_BYTES
        if b[sb:sb + 8] != _SIG:
            raise ValueError("HDF5 superblock not found at byte 512")
        version, osz, lsz = b[sb + 8], b[sb + 9], b[sb + 10]
        if version not in (2, 3) or osz != 8 or lsz != 8:
            raise ValueError("unsupported superblock (need version 2/3 with 8-byte offsets and lengths)")
        self.base, _ext, self.eof, root = struct.unpack_from("<QQQQ", b, sb + 12)
        if lookup3(b[sb:sb + 44]) != struct.unpack_from("<I", b, sb + 44)[0]:
            raise ValueError("superblock checksum mismatch")
        self._cache = {}
        self.root = self._group(root)

    # -- addressing
    def _abs(self, rel):
        return rel + self.base

    # -- object headers
    def _messages(self, rel):
        """All (type, flags, body) messages of the object header at relative address `rel`."""
        b = self.buf
        p = self._abs(rel)
        if b[p:p + 4] != b"OHDR" or b[p + 4] != 2:
            raise ValueError(f"version-2 object header expected at {p}")
        flags = b[p + 5]
        q = p + 6
        if flags & 0x20:
            q += 16
        if flags & 0x10:
            q += 4
        nsz = 1 << (flags & 3)
        chunk = int.from_bytes(b[q:q + nsz], "little")
        q += nsz
        if lookup3(b[p:q + chunk]) != struct.unpack_from("<I", b, q + chunk)[0]:
            raise ValueError(f"object header checksum mismatch at {p}")
        out = []
        blocks = [(q, q + chunk)]
        track = bool(flags & 0x04)
        while blocks:
            q, end = blocks.pop(0)
            while q + 4 <= end:
                mtype = b[q]
                msize = struct.unpack_from("<H", b, q + 1)[0]
                mflags = b[q + 3]
                q += 4 + (2 if track else 0)
                body = b[q:q + msize]
                q += msize
                if mtype == 0x10:                              # continuation
                    off, ln = struct.unpack_from("<QQ", body, 0)
                    a = self._abs(off)
                    if b[a:a + 4] != b"OCHK":
                        raise ValueError("bad continuation block")
                    blocks.append((a + 4, a + ln - 4))
                elif mtype != 0:
                    out.append((mtype, mflags, body))
        return out

    def _group(self, rel):
        links = {}
        for mtype, _, body in self._messages(rel):
            if mtype == 0x06:
                name, addr = self._link(body)
                links[name] = addr
        return links

    @staticmethod
    def _link(body):
        flags = body[1]
        q = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[q]
            q += 1
        if flags & 0x04:
            q += 8
        if flags & 0x10:
            q += 1
        nsz = 1 << (flags & 3)
        n = int.from_bytes(body[q:q + nsz], "little")
        q += nsz
        name = body[q:q + n].decode("utf-8")
        q += n
        if ltype != 0:
            raise ValueError("only hard links are supported")
        return name, struct.unpack_from("<Q", body, q)[0]

    # -- datatypes
    def _datatype(self, body, mflags=0):
        if mflags & 0x02:                                       # shared: committed datatype
            addr = struct.unpack_from("<Q", body, 2)[0]
            return self._committed(addr)
        return self._parse_dt(body, 0)[0]

    def _committed(self, rel):
        # the `julia_type` attribute of a committed datatype (the Julia type's name and parameters, itself stored
        # with the self-referential DataType datatype) is not needed to decode values and is not followed
        if ("dt", rel) not in self._cache:
            dt = None
            for mtype, _, body in self._messages(rel):
                if mtype == 0x03:
                    dt = self._parse_dt(body, 0)[0]
            dt.committed = rel
            self._cache[("dt", rel)] = dt
        return self._cache[("dt", rel)]

    def _parse_dt(self, b, q):
        cv = b[q]
        cls, ver = cv & 0x0F, cv >> 4
        bits = b[q + 1] | (b[q + 2] << 8) | (b[q + 3] << 16)
        size = struct.unpack_from("<I", b, q + 4)[0]
        q += 8
        if cls == 0:                                            # fixed point
            return _Datatype(0, size, signed=bool(bits & 0x08)), q + 4
        if cls == 1:                                            # floating point
            return _Datatype(1, size), q + 12
        if cls == 3:                                            # fixed-length string
            return _Datatype(3, size), q
        if cls == 4:                                            # bit field
            return _Datatype(4, size), q + 4
        if cls == 5:                                            # opaque
            n = (bits & 0xFF)
            return _Datatype(5, size), q + n
        if cls == 6:                                            # compound
            n = bits & 0xFFFF
            members = []
            for _ in range(n):
                e = b.index(b"\x00", q)
                name = b[q:e].decode("utf-8")
                if ver == 3:
                    q = e + 1
                    osz = 1 if size < 256 else 2 if size < 65536 else 3 if size < 16777216 else 4
                    off = int.from_bytes(b[q:q + osz], "little")
                    q += osz
                else:
                    q += (e - q + 8) // 8 * 8
                    off = struct.unpack_from("<I", b, q)[0]
                    q += 4 if ver == 2 else 4 + 1 + 3 + 4 + 4 + 16
                mdt, q = self._parse_dt(b, q)
                members.append((name, off, mdt))
            return _Datatype(6, size, members=members), q
        if cls == 7:                                            # reference
            return _Datatype(7, size), q
        if cls == 9:                                            # variable length
            base, q = self._parse_dt(b, q)
            return _Datatype(9, size, base=base, is_string=(bits & 0x0F) == 1), q
        raise ValueError(f"unsupported datatype class {cls}")

    def _attribute(self, body):
        ver = body[0]
        nsz, dtsz, dssz = struct.unpack_from("<HHH", body, 2)
        q = 8 + (1 if ver == 3 else 0)
        name = body[q:q + nsz].rstrip(b"\x00").decode("utf-8")
        if ver == 1:
            pad = lambda n: (n + 7) // 8 * 8
            dtb = body[q + pad(nsz):q + pad(nsz) + dtsz]
            dsb = body[q + pad(nsz) + pad(dtsz):q + pad(nsz) + pad(dtsz) + dssz]
            data = body[q + pad(nsz) + pad(dtsz) + pad(dssz):]
        else:
            dtb = body[q + nsz:q + nsz + dtsz]
            dsb = body[q + nsz + dtsz:q + nsz + dtsz + dssz]
            data = body[q + nsz + dtsz + dssz:]
        shared = bool(body[1] & 0x01)
        dt = self._datatype(dtb, 0x02 if shared else 0)
        dims = self._dataspace(dsb)
        return name, self._decode(dt, dims, data)

    @staticmethod
    def _dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            q = 8
        else:
            q = 4
            if body[3] == 2:                                    # null dataspace
                return None
        return tuple(struct.unpack_from("<Q", body, q + 8 * i)[0] for i in range(rank))

    # -- data
    def _decode_one(self, dt, raw):
        if dt.cls == 0:
            return int.from_bytes(raw[:dt.size], "little", signed=dt.signed)
        if dt.cls == 1:
            return struct.unpack("<d" if dt.size == 8 else "<f", raw[:dt.size])[0]
        if dt.cls == 3:
            return raw[:dt.size].rstrip(b"\x00").decode("utf-8")
        if dt.cls in (4, 5):
            return bytes(raw[:dt.size])
        if dt.cls == 7:
            addr = struct.unpack_from("<Q", raw, 0)[0]
            return None if addr in (0, UNDEF) else self._object(addr)
        if dt.cls == 9:
            n, addr, idx = struct.unpack_from("<IQI", raw, 0)
            data = self._global_heap(addr, idx)
            if dt.is_string:
                return data[:n].decode("utf-8")
            return [self._decode_one(dt.base, data[i * dt.base.size:(i + 1) * dt.base.size]) for i in range(n)]
        if dt.cls == 6:
            return {name: self._decode_one(m, raw[off:off + m.size]) for name, off, m in dt.members}
        raise ValueError(f"cannot decode datatype class {dt.cls}")

    def _decode(self, dt, dims, raw):
        if dims is None:
            return None
        if len(dims) == 0:
            return self._decode_one(dt, raw)
        n = int(np.prod(dims)) if dims else 1
        shape = tuple(reversed(dims))                           # HDF5 stores Julia's dimensions reversed
        if dt.cls == 1 or (dt.cls == 0 and dt.size in (1, 2, 4, 8)):
            np_dt = {(1, 8): "<f8", (1, 4): "<f4"}.get((dt.cls, dt.size)) or ("<i" if dt.signed else "<u") + str(dt.size)
            a = np.frombuffer(raw, dtype=np_dt, count=n)
            return a.reshape(shape, order="F").copy() if len(shape) > 1 else a.copy()
        items = [self._decode_one(dt, raw[i * dt.size:(i + 1) * dt.size]) for i in range(n)]
        if len(shape) > 1:
            out = np.empty(shape, dtype=object)
            out.reshape(-1, order="F")[:] = items
            return out
        return items

    def _global_heap(self, rel, index):
        b = self.buf
        p = self._abs(rel)
        if b[p:p + 4] != b"GCOL":
            raise ValueError("global heap collection expected")
        size = struct.unpack_from("<Q", b, p + 8)[0]
        q = p + 16
        while q < p + size:
            idx, _refs = struct.unpack_from("<HH", b, q)
            osize = struct.unpack_from("<Q", b, q + 8)[0]
            if idx == index:
                return b[q + 16:q + 16 + osize]
            if idx == 0:
                break
            q += 16 + (osize + 7) // 8 * 8
        raise KeyError("global heap object not found")

    def _object(self, rel):
        if rel in self._cache:
            return self._cache[rel]
        dt = dims = raw = None
        attrs = {}
        is_group = False
        for mtype, mflags, body in self._messages(rel):
            if mtype == 0x01:
                dims = self._dataspace(body)
            elif mtype == 0x03:
                dt = self._datatype(body, mflags)
            elif mtype == 0x08:
                ver, lc = body[0], body[1]
                if ver not in (3, 4):
                    raise ValueError("unsupported data layout version")
                if lc == 0:
                    n = struct.unpack_from("<H", body, 2)[0]
                    raw = body[4:4 + n]
                elif lc == 1:
                    addr, n = struct.unpack_from("<QQ", body, 2)
                    raw = b"" if addr == UNDEF else self.buf[self._abs(addr):self._abs(addr) + n]
                else:
                    raise ValueError("chunked / compressed datasets are not supported")
            elif mtype in (0x02, 0x06, 0x0A):
                is_group = True
        if dt is None and is_group:
            val = {k: self._object(a) for k, a in self._group(rel).items()}
        else:
            val = self._decode(dt, dims, raw)
        self._cache[rel] = val
        return val

    # -- public
    def keys(self):
        return [k for k in self.root if k != "_types"]

    def __contains__(self, name):
        return name in self.root

    def __getitem__(self, name):
        node = self.root
        parts = name.split("/")
        for i, part in enumerate(parts):
            if part not in node:
                raise KeyError(name)
            if i == len(parts) - 1:
                return self._object(node[part])
            node = self._group(node[part])


def load(path):
    """dict of every top-level entry (the `jldopen(path) do file ... end` pattern of the reference scripts)."""
    f = JLD2File(path)
    return {k: f[k] for k in f.keys()}


# ----------------------------------------------------------------------------- writer
_DT_INT64 = bytes.fromhex("300800000800000000004000")
_DT_FLOAT64 = bytes.fromhex("31203f000800000000004000340b0034ff030000")
_FILL = bytes.fromhex("05020000" "0309")
_TRAILER = b"\x00\x10\x00\x00" + b"\x00" * 16        # room JLD2.jl leaves for a continuation message


def _msg(mtype, body, flags=0):
    return struct.pack("<BHB", mtype, len(body), flags) + body


def _ohdr(messages):
    body = b"".join(messages)
    nsz_code = 0 if len(body) < 256 else 1 if len(body) < 65536 else 2
    head = b"OHDR\x02" + bytes([nsz_code]) + len(body).to_bytes(1 << nsz_code, "little")
    blob = head + body
    return blob + struct.pack("<I", lookup3(blob))


def _dataset_header(value, data_rel):
    """Object header of a scalar (compact) or dense array (contiguous at relative address data_rel)."""
    if isinstance(value, np.ndarray):
        dt = _DT_FLOAT64 if value.dtype.kind == "f" else _DT_INT64
        dims = tuple(reversed(value.shape))
        space = bytes([2, len(dims), 0, 1]) + b"".join(struct.pack("<Q", n) for n in dims)
        layout = b"\x04\x01" + struct.pack("<QQ", data_rel, value.size * 8)
    else:
        dt = _DT_FLOAT64 if isinstance(value, float) else _DT_INT64
        space = bytes([2, 0, 0, 0])
        raw = struct.pack("<d", value) if isinstance(value, float) else struct.pack("<q", value)
        layout = b"\x04\x00" + struct.pack("<H", 8) + raw
    return _ohdr([_FILL, _msg(0x01, space), _msg(0x03, dt, 1), _msg(0x08, layout), _TRAILER])


# ---- Vector{Vector{Float64}} (the reference's `parameters` / `betas` checkpoints, c-peptide/02-conditional.jl:44-50):
# JLD2.jl stores the outer vector as a dataset of object references with a `julia_type` attribute that points at a
# description of the element type Array{Float64,1}; that description is a small graph of objects of the committed
# compound datatype for Core.DataType {name::String, parameters::Vector{Any}} whose strings and reference lists live
# in a 4 KiB global heap.  The block is written once, in front of the first such dataset, and `_types/00000001` in
# the root group links the committed datatype.
_DT_REFERENCE = bytes.fromhex("3700000008000000")
# compound {name: variable-length string @0, parameters: variable-length sequence of references @16}, 32 bytes
_DT_DATATYPE = bytes.fromhex("36020000200000006e616d65000039110100100000003000000001000000000008007061"
                             "72616d6574657273001039000000100000003700000008000000")
_HEAP_BYTES = 4096


def _vlen(length, heap_rel, index):
    return struct.pack("<IQI", length, heap_rel if length else 0, index if length else 0)


def _pad8(out):
    out += b"\x00" * (-len(out) % 8)


def _type_block(out):
    """Appends the Array{Float64,1} type description; returns (address of the committed datatype, address of the
    Array{Float64,1} DataType object), both relative to the end of the file header."""
    t0 = len(out) - HEADER_BYTES
    # sizes are fixed, so the addresses of everything that follows are known up front
    def committed(heap_rel):
        shared = b"\x03\x02" + struct.pack("<Q", t0)
        attr = (bytes.fromhex("02010b000a000400") + b"julia_type\x00" + shared + bytes([2, 0, 0, 0]) +
                _vlen(13, heap_rel, 1) + _vlen(0, 0, 0))
        return _ohdr([_msg(0x03, _DT_DATATYPE, 0x40), _msg(0x0C, attr)])
    heap = (t0 + len(committed(0)) + 7) // 8 * 8
    array_obj = heap + _HEAP_BYTES + 16
    dt_obj_len = len(_datatype_object(0, 0, 0, 0, (0, 0)))
    float_obj = array_obj + dt_obj_len
    one_obj = float_obj + dt_obj_len
    out += committed(heap)
    _pad8(out)
    assert len(out) - HEADER_BYTES == heap
    objects = [b"Core.DataType", b"Core.Array", b"Core.Float64", struct.pack("<QQ", float_obj, one_obj)]
    blob = bytearray(b"GCOL\x01\x00\x00\x00" + struct.pack("<Q", _HEAP_BYTES))
    for k, data in enumerate(objects, 1):
        blob += struct.pack("<HHIQ", k, 1, 0, len(data)) + data + b"\x00" * (-len(data) % 8)
    blob += struct.pack("<HHIQ", 0, 0, 0, _HEAP_BYTES - len(blob))        # free space, its own header included
    out += bytes(blob).ljust(_HEAP_BYTES, b"\x00") + b"\x00" * 16
    out += _datatype_object(t0, heap, 10, 2, (2, 4))                      # Core.Array, parameters = heap object 4
    out += _datatype_object(t0, heap, 12, 3, (0, 0))                      # Core.Float64, no parameters
    assert len(out) - HEADER_BYTES == one_obj
    out += _dataset_header(1, 0)
    return t0, array_obj


def _datatype_object(t0, heap, name_len, name_idx, params):
    """A Core.DataType instance: scalar dataset of the committed compound type, compact layout."""
    shared = b"\x03\x02" + struct.pack("<Q", t0)
    data = _vlen(name_len, heap, name_idx) + _vlen(params[0], heap, params[1])
    return _ohdr([_FILL, _msg(0x01, bytes([2, 0, 0, 0])), _msg(0x03, shared, 3),
                  _msg(0x08, b"\x04\x00" + struct.pack("<H", len(data)) + data), _TRAILER])


def _reference_vector_header(n, julia_type_rel, data_rel):
    attr = (bytes.fromhex("02000b0008000400") + b"julia_type\x00" + _DT_REFERENCE + bytes([2, 0, 0, 0]) +
            struct.pack("<Q", julia_type_rel))
    space = bytes([2, 1, 0, 1]) + struct.pack("<Q", n)
    return _ohdr([_FILL, _msg(0x01, space), _msg(0x0C, attr), _msg(0x03, _DT_REFERENCE, 1),
                  _msg(0x08, b"\x04\x01" + struct.pack("<QQ", data_rel, n * 8)), _TRAILER])


def _append_array(out, value):
    """dense array dataset: header, then the data 8-byte aligned; returns its relative address."""
    addr = len(out)
    hdr_len = len(_dataset_header(value, 0))
    data_at = (addr + hdr_len + 7) // 8 * 8
    out += _dataset_header(value, data_at - HEADER_BYTES)
    out += b"\x00" * (data_at - len(out))
    out += value.tobytes(order="F")
    return addr - HEADER_BYTES


def _group_header(links):
    msgs = [_msg(0x02, b"\x00\x00" + b"\xff" * 16), _msg(0x0A, b"\x00\x00")]
    for name, rel in links:
        nb = name.encode("utf-8")
        if len(nb) > 255:
            raise ValueError("link name too long")
        msgs.append(_msg(0x06, b"\x01\x10\x01" + bytes([len(nb)]) + nb + struct.pack("<Q", rel)))
    if len(links) < 4:                                          # JLD2.jl reserves room for 4 links of 8-char names
        msgs.append(_msg(0x00, b"\x00" * ((4 - len(links)) * 24 - 4)))
    msgs.append(_TRAILER)
    return _ohdr(msgs)


def save(path, entries, julia_version="1.11.1", vectors_as_matrix=False):
    """Write `entries` (ordered dict name -> int | float | float64/int64 ndarray | list of float64 vectors) as a JLD2
    file with the layout JLD2.jl itself produces for `jldopen(path, "w") do f; f[name] = value; ...; end` on these
    types: datasets in insertion order (scalars compact, arrays contiguous and 8-byte aligned after their header), the
    root group last.  An (n1, n2, ...) ndarray becomes the Julia Array of the same size (written in column-major
    order).  A list of 1-D float vectors is written as Julia's Vector{Vector{Float64}} -- a dataset of object
    references, one dataset per inner vector, the element-type description in front of the first one and linked from
    `_types` -- which is how the reference's scripts store and index `parameters` / `betas`
    (c-peptide/02-conditional.jl:44-62: `parameters[best_model_index]`); vectors_as_matrix=True stores it as a matrix
    with one column per vector instead."""
    header = (b"HDF5-based Julia Data Format, version 0.2.0\x00 (Julia " + julia_version.encode() + b" 64-bit LE)\x00")
    out = bytearray(header.ljust(HEADER_BYTES, b"\x00"))
    out += b"\x00" * 48                                         # superblock, filled in at the end
    links = []
    types = None                                                # (committed datatype, Array{Float64,1} description)
    for name, value in entries.items():
        if name == "_types":
            raise ValueError("'_types' is JLD2's own group of committed datatypes: not available as a dataset name")
        if isinstance(value, (list, tuple)) and len(value) and isinstance(value[0], np.ndarray):
            if vectors_as_matrix:
                value = np.stack(value, axis=1)
            else:
                vecs = [np.asarray(v, dtype=np.float64).reshape(-1) for v in value]
                if any(np.asarray(v).dtype.kind != "f" for v in value):
                    raise TypeError(f"{name}: only vectors of Float64 vectors are supported")
                if types is None:
                    types = _type_block(out)
                addr = len(out) - HEADER_BYTES
                refs_at = (len(out) + len(_reference_vector_header(len(vecs), 0, 0)) + 7) // 8 * 8
                out += _reference_vector_header(len(vecs), types[1], refs_at - HEADER_BYTES)
                out += b"\x00" * (refs_at - len(out))
                out += b"\x00" * (8 * len(vecs))                # the references, known once the vectors are placed
                rels = [_append_array(out, v) for v in vecs]
                out[refs_at:refs_at + 8 * len(vecs)] = struct.pack(f"<{len(vecs)}Q", *rels)
                links.append((name, addr))
                continue
        if isinstance(value, (bool, np.bool_)):
            raise TypeError("Bool is not supported")
        if isinstance(value, (np.integer,)):
            value = int(value)
        if isinstance(value, (np.floating,)):
            value = float(value)
        if isinstance(value, np.ndarray):
            kind = value.dtype.kind
            if kind not in "fiu":
                raise TypeError(f"{name}: only float64 / int64 arrays are supported")
            value = np.asarray(value, dtype=np.float64 if kind == "f" else np.int64)
        elif not isinstance(value, (int, float)):
            raise TypeError(f"{name}: unsupported type {type(value).__name__}")
        if isinstance(value, np.ndarray):
            links.append((name, _append_array(out, value)))
        else:
            links.append((name, len(out) - HEADER_BYTES))
            out += _dataset_header(value, 0)
    if types is not None:
        links.append(("_types", len(out) - HEADER_BYTES))
        out += _group_header([("00000001", types[0])])
    root = len(out) - HEADER_BYTES
    out += _group_header(links)
    sb = _SIG + bytes([2, 8, 8, 0]) + struct.pack("<QQQQ", HEADER_BYTES, UNDEF, len(out), root)
    out[HEADER_BYTES:HEADER_BYTES + 48] = sb + struct.pack("<I", lookup3(sb))
    with open(path, "wb") as fh:
        fh.write(out)
