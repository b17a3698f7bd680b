"""TensorFlow V2 checkpoint bundles (``<prefix>.index`` + ``<prefix>.data-00000-of-00001``) without TensorFlow.

What ``tf.train.Saver().save / .restore`` reads and writes in the reference (cifar10/gan_resnet.py:906-925,
mnist/model.py:265,398-425): the variable names of SURVEY Appendix A map to raw little-endian tensors.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/{table_builder,block_builder,format}.cc -- the
LevelDB table format):

* ``.data-00000-of-00001``: the tensors' bytes back to back, in the order they were added (sorted by name, as
  ``Saver`` does).
* ``.index``: an SSTable.  Key ``""`` -> ``BundleHeaderProto`` {num_shards=1, endianness=LITTLE, version.producer=1};
  key ``<tensor name>`` -> ``BundleEntryProto`` {dtype, shape, shard_id=0, offset, size, crc32c = masked CRC-32C of the
  tensor bytes}.  A table = data blocks, an (empty) metaindex block, an index block, a 48-byte footer (the two block
  handles as varint64 pairs, zero-padded to 40 bytes, then the magic 0xdb4775248b80fb57).  A block = prefix-compressed
  entries (varint32 shared, unshared, value length; key suffix; value) + the restart offsets (fixed32 each) + their
  count; every block is followed by a 5-byte trailer: compression type 0 and the masked CRC-32C of block + type byte.
* ``checkpoint``: the CheckpointState text proto (``model_checkpoint_path``, ``all_model_checkpoint_paths``); written by
  ``host.Saver``.

The reference ships no checkpoint and TensorFlow is not installable here, so the byte layout is pinned only against the
format description above and an independent restatement in ``oracle/tf_bundle_ref.py`` (tests/test_tf_bundle_cpu.py): parity
with a TensorFlow-written bundle is *unpinned*.  CRCs come from the C ABI (``rcgan_crc32c``, known-answer tested).
"""
import os
import struct

import numpy as np

from . import _lib as L

MAGIC = 0xDB4775248B80FB57
BLOCK_SIZE = 262144              # table::Options::block_size in TensorFlow
RESTART_INTERVAL = 16
# tensorflow/core/framework/types.proto
DTYPES = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("uint8"): 4, np.dtype("int16"): 5,
          np.dtype("int8"): 6, np.dtype("int64"): 9, np.dtype("bool"): 10, np.dtype("float16"): 19}
NP_OF = {v: k for k, v in DTYPES.items()}


def crc32c(data, crc=0):
    data = bytes(data)
    return int(L.load().rcgan_crc32c(crc, data, len(data))) & 0xFFFFFFFF


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


# ---- protobuf (only the three messages of tensor_bundle.proto / tensor_shape.proto) -------------------------
def _field_varint(num, v):
    return _varint(num << 3) + _varint(v)


def _field_bytes(num, b):
    return _varint((num << 3) | 2) + _varint(len(b)) + b


def header_proto():
    version = _field_varint(1, 1)                                    # VersionDef.producer = kTensorBundleVersion
    return _field_varint(1, 1) + _field_bytes(3, version)            # num_shards = 1; endianness LITTLE (0) is the default


def entry_proto(dtype, shape, offset, size, crc_masked):
    dims = b"".join(_field_bytes(2, _field_varint(1, int(d))) for d in shape)
    out = _field_varint(1, dtype) + _field_bytes(2, dims)
    if offset:
        out += _field_varint(4, offset)                              # shard_id = 0 is the default
    out += _field_varint(5, size)
    out += _varint((6 << 3) | 5) + struct.pack("<I", crc_masked)     # fixed32
    return out


def parse_proto(buf):
    """{field: [values]} of one message (varint -> int, length-delimited -> bytes, fixed32/64 -> int)."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(num, []).append(v)
    return out


# ---- LevelDB table ----------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last_key = b""

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            n = min(len(key), len(self.last_key))
            while shared < n and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _write_block(f, contents):
    """block + trailer; returns the BlockHandle (offset, size without trailer)."""
    off = f.tell()
    trailer_type = b"\x00"                                           # kNoCompression
    crc = mask_crc(crc32c(trailer_type, crc32c(contents)))
    f.write(contents + trailer_type + struct.pack("<I", crc))
    return off, len(contents)


def write_table(path, items):
    """items: [(key bytes, value bytes)] sorted by key."""
    with open(path, "wb") as f:
        index = _BlockBuilder(1)
        data = _BlockBuilder(RESTART_INTERVAL)
        pending = None                                               # (last key, handle) of a finished data block
        n_in_block = 0
        for key, value in items:
            if pending is not None:
                index.add(pending[0], _varint(pending[1][0]) + _varint(pending[1][1]))
                pending = None
            data.add(key, value)
            n_in_block += 1
            if data.size() >= BLOCK_SIZE:
                pending = (key, _write_block(f, data.finish()))
                data = _BlockBuilder(RESTART_INTERVAL)
                n_in_block = 0
        if n_in_block:
            pending = (data.last_key, _write_block(f, data.finish()))
        if pending is not None:
            index.add(pending[0], _varint(pending[1][0]) + _varint(pending[1][1]))
        meta = _write_block(f, _BlockBuilder(RESTART_INTERVAL).finish())
        idx = _write_block(f, index.finish())
        footer = _varint(meta[0]) + _varint(meta[1]) + _varint(idx[0]) + _varint(idx[1])
        footer += b"\x00" * (40 - len(footer))
        f.write(footer + struct.pack("<Q", MAGIC))


def _read_block(buf, off, size):
    contents = buf[off:off + size]
    ctype = buf[off + size:off + size + 1]
    stored = struct.unpack_from("<I", buf, off + size + 1)[0]
    if mask_crc(crc32c(ctype, crc32c(contents))) != stored:
        raise ValueError("table block checksum mismatch at offset %d" % off)
    if ctype != b"\x00":
        raise ValueError("compressed table blocks are not supported")
    n_restarts = struct.unpack_from("<I", contents, len(contents) - 4)[0]
    end = len(contents) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        unshared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = key[:shared] + contents[pos:pos + unshared]
        pos += unshared
        out.append((key, contents[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path):
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad magic)" % path)
    pos = len(buf) - 48
    _, pos = _read_varint(buf, pos)
    _, pos = _read_varint(buf, pos)
    ioff, pos = _read_varint(buf, pos)
    isize, pos = _read_varint(buf, pos)
    items = []
    for _, handle in _read_block(buf, ioff, isize):
        off, p2 = _read_varint(handle, 0)
        size, _ = _read_varint(handle, p2)
        items += _read_block(buf, off, size)
    return items


# ---- bundles ------------------------------------------------------------------------------------------------
def write_bundle(prefix, tensors):
    """tensors: {name: numpy array}.  Writes <prefix>.index and <prefix>.data-00000-of-00001."""
    names = sorted(tensors, key=lambda s: s.encode())
    items = [(b"", header_proto())]
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in names:
            a = np.asarray(tensors[name])                      # (ascontiguousarray would turn a scalar into shape [1])
            a = a if a.flags.c_contiguous else a.copy()
            if a.dtype not in DTYPES:
                raise TypeError("tensor %s: dtype %s has no checkpoint representation here" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            items.append((name.encode(), entry_proto(DTYPES[a.dtype], a.shape, f.tell(), len(raw), mask_crc(crc32c(raw)))))
            f.write(raw)
    write_table(prefix + ".index", items)


def read_bundle(prefix):
    """{name: numpy array}; verifies every block and tensor checksum."""
    items = read_table(prefix + ".index")
    if not items or items[0][0] != b"":
        raise ValueError("bundle header missing in %s.index" % prefix)
    hdr = parse_proto(items[0][1])
    shards = hdr.get(1, [1])[0]
    if shards != 1 or hdr.get(2, [0])[0] != 0:
        raise ValueError("only single-shard little-endian bundles are supported (num_shards=%d)" % shards)
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        data = f.read()
    out = {}
    for key, val in items[1:]:
        e = parse_proto(val)
        if 7 in e:
            raise ValueError("tensor %s is stored in slices (partitioned variable): not supported" % key.decode())
        dtype = e.get(1, [0])[0]
        if dtype not in NP_OF:
            raise TypeError("tensor %s: checkpoint dtype %d not supported" % (key.decode(), dtype))
        shape = []
        for dim in parse_proto(e.get(2, [b""])[0]).get(2, []):
            shape.append(parse_proto(dim).get(1, [0])[0])
        off, size = e.get(4, [0])[0], e.get(5, [0])[0]
        raw = data[off:off + size]
        if len(raw) != size or mask_crc(crc32c(raw)) != e.get(6, [0])[0]:
            raise ValueError("tensor %s: data checksum mismatch" % key.decode())
        out[key.decode()] = np.frombuffer(raw, NP_OF[dtype].newbyteorder("<")).reshape(shape).astype(NP_OF[dtype])
    return out


def exists(prefix):
    return os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
