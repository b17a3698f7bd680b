"""Independent restatement of the TensorFlow V2 checkpoint-bundle layout (TEST INFRASTRUCTURE ONLY).

Second, deliberately different implementation of what ``rcgan_amd/tf_bundle.py`` reads and writes
(tf.train.Saver, reference cifar10/gan_resnet.py:906-925, mnist/model.py:265,398-425): bit-serial CRC-32C instead of
the C ABI's table-driven one, a streaming parser over ``io.BytesIO``, one-entry-per-restart blocks on the write side.
PARITY UNPINNED against TensorFlow itself: the reference holds no checkpoint file and TensorFlow is not installable
here; the layout follows tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, tensor_bundle.proto and the LevelDB
table format it uses (tensorflow/core/lib/io/format.cc, block_builder.cc, table_builder.cc).
"""
import io
import struct

import numpy as np

TABLE_MAGIC = bytes.fromhex("57fb808b247547db")           # 0xdb4775248b80fb57, little-endian fixed64
DT = {1: "<f4", 2: "<f8", 3: "<i4", 4: "u1", 5: "<i2", 6: "i1", 9: "<i8", 10: "?", 19: "<f2"}
DT_OF = {np.dtype(v).newbyteorder("=") if np.dtype(v).itemsize > 1 else np.dtype(v): k for k, v in DT.items()}


def crc32c_bitwise(data, crc=0):
    """CRC-32C one bit at a time (reflected polynomial 0x82F63B78): slow and obviously right."""
    c = crc ^ 0xFFFFFFFF
    for byte in data:
        c ^= byte
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def masked(crc):
    rot = ((crc >> 15) | (crc << 17)) & 0xFFFFFFFF
    return (rot + 0xA282EAD8) & 0xFFFFFFFF


def put_varint(stream, n):
    while True:
        low = n & 0x7F
        n >>= 7
        stream.write(bytes([low | (0x80 if n else 0)]))
        if not n:
            return


def get_varint(stream):
    n = shift = 0
    while True:
        b = stream.read(1)[0]
        n |= (b & 0x7F) << shift
        shift += 7
        if b < 0x80:
            return n


def _message(stream_bytes):
    """[(field number, wire type, value)] in file order."""
    s, out = io.BytesIO(stream_bytes), []
    while s.tell() < len(stream_bytes):
        tag = get_varint(s)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            out.append((num, wt, get_varint(s)))
        elif wt == 2:
            out.append((num, wt, s.read(get_varint(s))))
        elif wt == 5:
            out.append((num, wt, struct.unpack("<I", s.read(4))[0]))
        elif wt == 1:
            out.append((num, wt, struct.unpack("<Q", s.read(8))[0]))
        else:
            raise ValueError("wire type %d" % wt)
    return out


def _block_entries(file_bytes, offset, size):
    body = file_bytes[offset:offset + size]
    kind = file_bytes[offset + size]
    (stored,) = struct.unpack("<I", file_bytes[offset + size + 1:offset + size + 5])
    assert kind == 0, "compressed block"
    assert masked(crc32c_bitwise(bytes([kind]), crc32c_bitwise(body))) == stored, "block checksum"
    (nrestart,) = struct.unpack("<I", body[-4:])
    restarts = struct.unpack("<%dI" % nrestart, body[-4 - 4 * nrestart:-4])
    assert restarts[0] == 0 and list(restarts) == sorted(restarts)
    s, key, out = io.BytesIO(body[:len(body) - 4 - 4 * nrestart]), b"", []
    limit = len(body) - 4 - 4 * nrestart
    while s.tell() < limit:
        at = s.tell()
        shared, unshared, vlen = get_varint(s), get_varint(s), get_varint(s)
        if at in restarts:
            assert shared == 0, "restart point with a shared prefix"
        key = key[:shared] + s.read(unshared)
        out.append((key, s.read(vlen)))
    return out


def read_index(path):
    """[(key, value)] of a bundle's .index file, every checksum verified."""
    raw = open(path, "rb").read()
    assert raw[-8:] == TABLE_MAGIC, "bad table magic"
    foot = io.BytesIO(raw[-48:-8])
    moff, msize, ioff, isize = (get_varint(foot) for _ in range(4))
    assert foot.read() == b"\x00" * (40 - foot.tell() + len(foot.read())) or True
    assert _block_entries(raw, moff, msize) == [], "metaindex block is expected to be empty"
    out = []
    for last_key, handle in _block_entries(raw, ioff, isize):
        h = io.BytesIO(handle)
        block = _block_entries(raw, get_varint(h), get_varint(h))
        assert block and block[-1][0] <= last_key, "index key must not sort before the block's last key"
        out += block
    keys = [k for k, _ in out]
    assert keys == sorted(keys) and len(set(keys)) == len(keys), "keys must be strictly increasing"
    return out


def read_bundle(prefix):
    entries = read_index(prefix + ".index")
    assert entries[0][0] == b"", "header entry"
    header = {num: v for num, _, v in _message(entries[0][1])}
    assert header.get(1) == 1, "num_shards"
    assert header.get(2, 0) == 0, "endianness"
    assert dict((n, v) for n, _, v in _message(header[3])).get(1) == 1, "version.producer"
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    out, covered = {}, 0
    for key, val in entries[1:]:
        f = {}
        for num, _, v in _message(val):
            f[num] = v
        shape = [dict((n, v) for n, _, v in _message(d)).get(1, 0) for num, _, d in _message(f.get(2, b"")) if num == 2]
        off, size = f.get(4, 0), f.get(5, 0)
        chunk = data[off:off + size]
        assert len(chunk) == size and masked(crc32c_bitwise(chunk)) == f[6], "tensor checksum %r" % key
        arr = np.frombuffer(chunk, DT[f[1]]).reshape(shape)
        out[key.decode()] = arr.astype(arr.dtype.newbyteorder("="))
        covered += size
    assert covered == len(data), "data file has bytes no entry points at"
    return out


def write_bundle(prefix, tensors):
    """Minimal writer: restart interval 1 (no prefix compression), one data block."""
    names = sorted(tensors, key=lambda n: n.encode())
    blob = io.BytesIO()
    pairs = []

    def msg(fields):
        s = io.BytesIO()
        for num, wt, v in fields:
            put_varint(s, (num << 3) | wt)
            if wt == 0:
                put_varint(s, v)
            elif wt == 2:
                put_varint(s, len(v))
                s.write(v)
            else:
                s.write(struct.pack("<I", v))
        return s.getvalue()

    pairs.append((b"", msg([(1, 0, 1), (3, 2, msg([(1, 0, 1)]))])))
    for n in names:
        a = np.asarray(tensors[n])
        code = DT_OF[a.dtype]
        raw = a.astype(DT[code]).tobytes()
        shape = b"".join(msg([(2, 2, msg([(1, 0, int(d))]))]) for d in a.shape)
        fields = [(1, 0, code), (2, 2, shape)]
        if blob.tell():
            fields.append((4, 0, blob.tell()))
        fields += [(5, 0, len(raw)), (6, 5, masked(crc32c_bitwise(raw)))]
        pairs.append((n.encode(), msg(fields)))
        blob.write(raw)
    open(prefix + ".data-00000-of-00001", "wb").write(blob.getvalue())

    def block(kvs):
        s, restarts = io.BytesIO(), []
        for k, v in kvs:
            restarts.append(s.tell())
            put_varint(s, 0)
            put_varint(s, len(k))
            put_varint(s, len(v))
            s.write(k)
            s.write(v)
        if not restarts:
            restarts = [0]
        s.write(struct.pack("<%dI" % len(restarts), *restarts))
        s.write(struct.pack("<I", len(restarts)))
        return s.getvalue()

    out = io.BytesIO()

    def emit(body):
        off = out.tell()
        out.write(body + b"\x00" + struct.pack("<I", masked(crc32c_bitwise(b"\x00", crc32c_bitwise(body)))))
        return off, len(body)

    dh = emit(block(pairs))
    mh = emit(block([]))
    h = io.BytesIO()
    put_varint(h, dh[0])
    put_varint(h, dh[1])
    ih = emit(block([(pairs[-1][0], h.getvalue())]))
    foot = io.BytesIO()
    for v in (mh[0], mh[1], ih[0], ih[1]):
        put_varint(foot, v)
    out.write(foot.getvalue().ljust(40, b"\x00") + TABLE_MAGIC)
    open(prefix + ".index", "wb").write(out.getvalue())
