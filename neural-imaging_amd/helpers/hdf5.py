"""
A small pure-Python HDF5 reader / writer - just enough of the file format to exchange Keras weight files
(`model.save_weights(..., save_format='h5')`, the checkpoint container of the reference: models/tfmodel.py:150-182)
without h5py, which this image does not have.

Reader: superblock versions 0 / 1, version-1 object headers (with continuation blocks), old-style groups (symbol table
message -> v1 B-tree -> symbol nodes -> local heap), compact and contiguous dataset layouts (layout message versions
1-3), fixed-point / IEEE float / fixed-length and variable-length string datatypes, attribute messages versions 1-3,
global heap collections (variable-length strings).  That is the feature set of files written by HDF5 1.8 / 1.10 with the
default ("earliest") format, which is what h5py under Keras produces.  Chunked / filtered datasets, new-style groups and
version-2 object headers raise NotImplementedError with the feature named.

Writer: the same subset - superblock 0, symbol-table groups, contiguous little-endian datasets, version-1 attribute
messages with fixed-length strings.  Files it writes are read back by libhdf5 (h5dump / h5py; see
tests/golden/make_keras_h5.py and tests/test_host_logic.py).

Format reference: "HDF5 File Format Specification Version 2.0" (the public HDF Group document), restated from its
section III (disk format level 1: superblock, B-trees, symbol nodes, heaps) and IV (object headers and messages).
"""
import struct
from collections import OrderedDict

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class Group(OrderedDict):
    """Children by name (Group or Dataset); `.attrs` holds the attributes."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.attrs = OrderedDict()

    def visit_datasets(self, prefix=''):
        for name, node in self.items():
            path = prefix + name
            if isinstance(node, Group):
                yield from node.visit_datasets(path + '/')
            else:
                yield path, node

    def __getitem__(self, key):
        if isinstance(key, str) and '/' in key and not OrderedDict.__contains__(self, key):
            node = self
            for part in key.strip('/').split('/'):
                node = OrderedDict.__getitem__(node, part)
            return node
        return OrderedDict.__getitem__(self, key)


class Dataset(object):
    def __init__(self, value, attrs=None):
        self.value = value
        self.attrs = OrderedDict() if attrs is None else attrs

    @property
    def shape(self):
        return self.value.shape

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.value, dtype=dtype)


# ======================================================================================================================
# reader
class _Reader(object):
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != SIGNATURE:
            raise ValueError('not an HDF5 file (signature mismatch at offset 0)')
        ver = buf[8]
        if ver not in (0, 1):
            raise NotImplementedError('HDF5 superblock version {} (only 0 / 1: the "earliest" format Keras files use)'.format(ver))
        self.so, self.sl = buf[13], buf[14]
        if self.so != 8 or self.sl != 8:
            raise NotImplementedError('HDF5 offsets / lengths of {} / {} bytes (only 8 / 8)'.format(self.so, self.sl))
        p = 24 + (4 if ver == 1 else 0)
        self.base = self.u64(p)
        p += 4 * 8                                   # base, free-space, end-of-file, driver-info addresses
        # root symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self.root_header = self.u64(p + 8)
        self._gheaps = {}

    def u8(self, p): return self.b[p]
    def u16(self, p): return struct.unpack_from('<H', self.b, p)[0]
    def u32(self, p): return struct.unpack_from('<I', self.b, p)[0]
    def u64(self, p): return struct.unpack_from('<Q', self.b, p)[0]

    # ---- object headers -------------------------------------------------------------------------------------------
    def messages(self, addr):
        addr += self.base
        if self.b[addr:addr + 4] == b'OHDR':
            raise NotImplementedError('version-2 object headers (file written with libver="latest")')
        if self.u8(addr) != 1:
            raise ValueError('object header version {} at {}'.format(self.u8(addr), addr))
        n_msgs, first_size = self.u16(addr + 2), self.u32(addr + 8)
        blocks = [(addr + 16, first_size)]
        out = []
        while blocks and len(out) < n_msgs:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < n_msgs:
                mtype, msize, flags = self.u16(p), self.u16(p + 2), self.u8(p + 4)
                body = p + 8
                if flags & 2:
                    raise NotImplementedError('shared object header messages')
                if mtype == 0x10:                     # continuation
                    blocks.append((self.u64(body) + self.base, self.u64(body + 8)))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    # ---- datatypes / dataspaces -------------------------------------------------------------------------------------
    def datatype(self, p):
        """-> (kind, numpy dtype or None, element size, bytes consumed)."""
        cv = self.u8(p)
        cls, ver = cv & 15, cv >> 4
        bits0 = self.u8(p + 1)
        size = self.u32(p + 4)
        order = '>' if bits0 & 1 else '<'
        if cls == 0:
            signed = bool(bits0 & 8)
            return 'num', np.dtype('{}{}{}'.format(order, 'i' if signed else 'u', size)), size, 8 + 4
        if cls == 1:
            return 'num', np.dtype('{}f{}'.format(order, size)), size, 8 + 12
        if cls == 3:
            return 'str', np.dtype('S{}'.format(size)), size, 8
        if cls == 9:
            kind = bits0 & 15                          # 0 sequence, 1 string
            if kind != 1:
                raise NotImplementedError('variable-length sequences (only variable-length strings)')
            _, _, _, used = self.datatype(p + 8)
            return 'vstr', None, size, 8 + used
        raise NotImplementedError('HDF5 datatype class {} (version {})'.format(cls, ver))

    def dataspace(self, p):
        ver, rank, flags = self.u8(p), self.u8(p + 1), self.u8(p + 2)
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.u8(p + 3) == 2:
                return None                            # null dataspace
            q = p + 4
        else:
            raise ValueError('dataspace message version {}'.format(ver))
        return tuple(self.u64(q + 8 * i) for i in range(rank))

    def global_heap_object(self, addr, index):
        addr += self.base
        if addr not in self._gheaps:
            if self.b[addr:addr + 4] != b'GCOL':
                raise ValueError('global heap collection expected at {}'.format(addr))
            size = self.u64(addr + 8)
            objs, p = {}, addr + 16
            while p + 16 <= addr + size:
                idx, osize = self.u16(p), self.u64(p + 8)
                if idx == 0:
                    break
                objs[idx] = bytes(self.b[p + 16:p + 16 + osize])
                p += 16 + ((osize + 7) & ~7)
            self._gheaps[addr] = objs
        return self._gheaps[addr][index]

    def decode(self, kind, dtype, esize, shape, raw):
        count = int(np.prod(shape)) if shape else 1
        if kind == 'vstr':
            vals = []
            for i in range(count):
                length, gaddr, gidx = struct.unpack_from('<IQI', raw, i * esize)
                vals.append(self.global_heap_object(gaddr, gidx)[:length] if length else b'')
            arr = np.array(vals, dtype=object)
            return arr.reshape(shape) if shape else arr.reshape(())[()]
        arr = np.frombuffer(raw, dtype=dtype, count=count)
        if kind == 'num' and dtype.byteorder == '>':
            arr = arr.astype(dtype.newbyteorder('<'))
        return arr.reshape(shape).copy() if shape else arr.reshape(())[()]

    # ---- attributes -------------------------------------------------------------------------------------------------
    def attribute(self, p):
        ver = self.u8(p)
        nsz, tsz, ssz = self.u16(p + 2), self.u16(p + 4), self.u16(p + 6)
        if ver == 1:
            q = p + 8
            pad = lambda n: (n + 7) & ~7
        elif ver in (2, 3):
            if self.u8(p + 1) & 3:
                raise NotImplementedError('shared datatype / dataspace in an attribute')
            q = p + 8 + (1 if ver == 3 else 0)
            pad = lambda n: n
        else:
            raise ValueError('attribute message version {}'.format(ver))
        name = bytes(self.b[q:q + nsz]).split(b'\0')[0].decode('utf8')
        q += pad(nsz)
        kind, dtype, esize, _ = self.datatype(q)
        q += pad(tsz)
        shape = self.dataspace(q)
        q += pad(ssz)
        if shape is None:
            return name, None
        count = int(np.prod(shape)) if shape else 1
        return name, self.decode(kind, dtype, esize, shape, self.b[q:q + count * esize])

    # ---- groups -------------------------------------------------------------------------------------------------------
    def heap_name(self, heap_addr, offset):
        heap_addr += self.base
        if self.b[heap_addr:heap_addr + 4] != b'HEAP':
            raise ValueError('local heap expected at {}'.format(heap_addr))
        data = self.u64(heap_addr + 24) + self.base
        end = data + offset
        while self.b[end] != 0:
            end += 1
        return bytes(self.b[data + offset:end]).decode('utf8')

    def btree_entries(self, addr, heap):
        addr += self.base
        if self.b[addr:addr + 4] != b'TREE':
            raise ValueError('B-tree node expected at {}'.format(addr))
        if self.u8(addr + 4) != 0:
            raise ValueError('group B-tree node expected (type 0)')
        level, used = self.u8(addr + 5), self.u16(addr + 6)
        p = addr + 24
        out = []
        for i in range(used):
            child = self.u64(p + 8 + 16 * i)
            if level > 0:
                out += self.btree_entries(child, heap)
            else:
                s = child + self.base
                if self.b[s:s + 4] != b'SNOD':
                    raise ValueError('symbol node expected at {}'.format(s))
                for k in range(self.u16(s + 6)):
                    e = s + 8 + 40 * k
                    out.append((self.heap_name(heap, self.u64(e)), self.u64(e + 8)))
        return out

    def node(self, header_addr):
        msgs = self.messages(header_addr)
        attrs = OrderedDict()
        for mtype, p, _ in msgs:
            if mtype == 0x0C:
                k, v = self.attribute(p)
                attrs[k] = v
        kinds = {m[0]: m for m in msgs}
        if 0x11 in kinds:                              # symbol table message: old-style group
            p = kinds[0x11][1]
            g = Group()
            g.attrs = attrs
            for name, child in self.btree_entries(self.u64(p), self.u64(p + 8)):
                g[name] = self.node(child)
            return g
        if 0x02 in kinds or 0x06 in kinds:
            raise NotImplementedError('new-style groups (link messages); write the file with the default format')
        if 0x08 not in kinds:
            raise ValueError('object at {} is neither a group nor a dataset'.format(header_addr))
        kind, dtype, esize, _ = self.datatype(kinds[0x03][1])
        shape = self.dataspace(kinds[0x01][1])
        if 0x0B in kinds:
            raise NotImplementedError('filtered (compressed) datasets')
        p = kinds[0x08][1]
        ver = self.u8(p)
        count = int(np.prod(shape)) if shape else 1
        nbytes = count * esize
        if ver == 3:
            cls = self.u8(p + 1)
            if cls == 0:
                raw = self.b[p + 4:p + 4 + self.u16(p + 2)]
            elif cls == 1:
                addr = self.u64(p + 2)
                raw = bytes(nbytes) if addr == UNDEF else self.b[addr + self.base:addr + self.base + nbytes]
            else:
                raise NotImplementedError('chunked dataset layout')
        elif ver in (1, 2):
            ndim, cls = self.u8(p + 1), self.u8(p + 2)
            if cls == 1:
                addr = self.u64(p + 8)
                raw = bytes(nbytes) if addr == UNDEF else self.b[addr + self.base:addr + self.base + nbytes]
            elif cls == 0:
                q = p + 8 + 4 * ndim
                raw = self.b[q + 4:q + 4 + self.u32(q)]
            else:
                raise NotImplementedError('chunked dataset layout')
        else:
            raise NotImplementedError('data layout message version {}'.format(ver))
        if shape is None:
            return Dataset(None, attrs)
        return Dataset(self.decode(kind, dtype, esize, shape, raw[:nbytes]), attrs)


def read_hdf5(path):
    """Parse a whole file into a tree: Group (OrderedDict of children, .attrs) / Dataset (.value ndarray, .attrs)."""
    with open(path, 'rb') as f:
        buf = memoryview(f.read())
    r = _Reader(buf)
    root = r.node(r.root_header)
    if not isinstance(root, Group):
        raise ValueError('root object is not a group')
    return root


# ======================================================================================================================
# writer
LEAF_K, INTERNAL_K = 4, 16


def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == 'f':
        spec = {4: (31, 23, 8, 23, 127), 8: (63, 52, 11, 52, 1023), 2: (15, 10, 5, 10, 15)}[dt.itemsize]
        sign, eloc, esz, msz, bias = spec
        return struct.pack('<BBBBI', 0x11, 0x20, sign, 0, dt.itemsize) + \
            struct.pack('<HHBBBBI', 0, 8 * dt.itemsize, eloc, esz, 0, msz, bias)
    if dt.kind in 'iu':
        return struct.pack('<BBBBI', 0x10, 0x08 if dt.kind == 'i' else 0, 0, 0, dt.itemsize) + \
            struct.pack('<HH', 0, 8 * dt.itemsize)
    if dt.kind == 'S':
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, dt.itemsize)      # null-padded ASCII
    raise TypeError('cannot store dtype {} (float, integer or fixed-length bytes)'.format(dt))


def _dataspace_message(shape):
    # version 1: version, rank, flags, 5 reserved bytes, dimensions
    return struct.pack('<BBB5x', 1, len(shape), 0) + b''.join(struct.pack('<Q', int(d)) for d in shape)


def _as_storable(value):
    if isinstance(value, (bytes, str)):
        value = np.array(value.encode('utf8') if isinstance(value, str) else value)
    elif isinstance(value, (list, tuple)) and value and all(isinstance(v, (bytes, str)) for v in value):
        value = np.array([v.encode('utf8') if isinstance(v, str) else v for v in value])
    arr = np.asarray(value)
    if arr.dtype.kind == 'U':
        arr = np.char.encode(arr, 'utf8')
    if arr.dtype.kind == 'S' and arr.dtype.itemsize == 0:
        arr = arr.astype('S1')
    if arr.dtype.kind in 'fiu' and arr.dtype.byteorder == '>':
        arr = arr.astype(arr.dtype.newbyteorder('<'))
    if arr.dtype.kind == 'b':
        arr = arr.astype(np.uint8)
    return np.ascontiguousarray(arr) if arr.shape else arr


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _attribute_message(name, value):
    arr = _as_storable(value)
    nm = name.encode('utf8') + b'\0'
    dt, ds = _dtype_message(arr.dtype), _dataspace_message(arr.shape)
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + arr.tobytes()
    if len(body) > 65000:
        raise ValueError('attribute {} is too large for one object-header message ({} bytes)'.format(name, len(body)))
    return _message(0x0C, body)


class _Writer(object):
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, data, align=8):
        self.buf += b'\0' * (-len(self.buf) % align)
        addr = len(self.buf)
        self.buf += data
        return addr

    def object_header(self, msgs):
        data = b''.join(msgs)
        return self.alloc(struct.pack('<BBHII4x', 1, 0, len(msgs), 1, len(data)) + data)

    def dataset(self, ds):
        arr = _as_storable(ds.value if isinstance(ds, Dataset) else ds)
        raw = arr.tobytes()
        addr = self.alloc(raw) if raw else UNDEF
        msgs = [_message(0x01, _dataspace_message(arr.shape)), _message(0x03, _dtype_message(arr.dtype), flags=1),
                # fill value (version 2): allocation time late, write time "if set", no value defined
                _message(0x05, struct.pack('<BBBB', 2, 2, 2, 0)),
                _message(0x08, struct.pack('<BBQQ', 3, 1, addr, len(raw)))]
        for k, v in (ds.attrs.items() if isinstance(ds, Dataset) else ()):
            msgs.append(_attribute_message(k, v))
        return self.object_header(msgs)

    def group(self, g):
        children = sorted(((name.encode('utf8'), node) for name, node in g.items()), key=lambda t: t[0])
        if len(children) > 2 * LEAF_K * 2 * INTERNAL_K:
            raise ValueError('a group of {} entries needs a two-level B-tree (not written)'.format(len(children)))
        child_addrs = [self.group(node)[0] if isinstance(node, Group) else self.dataset(node) for _, node in children]
        # local heap: the empty string at offset 0, then the names
        heap_data, offsets = bytearray(8), []
        for name, _ in children:
            offsets.append(len(heap_data))
            heap_data += _pad8(name + b'\0')
        free = len(heap_data)
        heap_data += struct.pack('<QQ', 1, 16)          # one free block: next = 1 (none), size 16
        data_addr = self.alloc(bytes(heap_data))
        heap_addr = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap_data), free, data_addr))
        # symbol nodes of up to 2K entries, one B-tree node above them
        node_addrs, keys = [], [0]
        per = 2 * LEAF_K
        for s in range(0, max(len(children), 1), per):
            part = list(range(s, min(s + per, len(children))))
            body = b'SNOD' + struct.pack('<BBH', 1, 0, len(part))
            for i in part:
                body += struct.pack('<QQII16x', offsets[i], child_addrs[i], 0, 0)
            body += b'\0' * (8 + per * 40 - len(body))
            node_addrs.append(self.alloc(body))
            keys.append(offsets[part[-1]] if part else 0)
        tree = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(node_addrs), UNDEF, UNDEF)
        for i, a in enumerate(node_addrs):
            tree += struct.pack('<QQ', keys[i], a)
        tree += struct.pack('<Q', keys[-1])
        tree += b'\0' * (24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(tree))
        tree_addr = self.alloc(tree)
        msgs = [_message(0x11, struct.pack('<QQ', tree_addr, heap_addr))]
        for k, v in g.attrs.items():
            msgs.append(_attribute_message(k, v))
        return self.object_header(msgs), tree_addr, heap_addr


def write_hdf5(path, root):
    """Write a tree of Group / Dataset (or plain ndarray leaves) as an HDF5 file in the "earliest" format."""
    w = _Writer()
    w.buf += b'\0' * 96                               # superblock (version 0) is 96 bytes with 8-byte offsets
    res = w.group(root)
    header, tree_addr, heap_addr = res
    eof = len(w.buf) + (-len(w.buf) % 8)
    w.buf += b'\0' * (eof - len(w.buf))
    sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
    sb += struct.pack('<QQII', 0, header, 1, 0) + struct.pack('<QQ', tree_addr, heap_addr)   # cached symbol table
    assert len(sb) == 96, len(sb)
    w.buf[:96] = sb
    with open(path, 'wb') as f:
        f.write(bytes(w.buf))
