"""Reader and writer for TensorFlow checkpoint files (the "tensor bundle" V2 format: `<prefix>.index` +
`<prefix>.data-00000-of-00001`), without TensorFlow.

The reference restores its trunks from an Object-Detection-API checkpoint and saves / restores its own with
tf.train.Saver (core/checkpoint_utils.py:64-117, core/trainer.py:86,..., monopsr_model.py:1225-1266).  This module
lets the same files drive this package directly: `read_checkpoint(prefix)` -> {variable name: numpy array}, which
core/checkpoint_utils.py's restore functions take as they take an .npz mapping; `write_checkpoint(prefix, tensors)`
produces a bundle of the same layout.

Format, restated from TensorFlow's published sources (tensor_bundle.proto / tensor_bundle.cc, lib/io/table_format):
  * `.index` is an immutable sorted string table (the LevelDB table format): data blocks of prefix-compressed
    key/value entries with a restart array, each followed by a 5-byte trailer (compression type, masked CRC-32C);
    a metaindex block, an index block mapping separator keys to block handles, and a 48-byte footer ending in the
    magic 0xdb4775248b80fb57.  Key "" holds a BundleHeaderProto, every other key is a variable name whose value is a
    BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (masked, of the tensor bytes), slices=7}.
  * `.data-XXXXX-of-YYYYY` holds the raw little-endian tensor bytes at the recorded offsets.
PARITY UNPINNED against a TensorFlow-written file: none exists offline (TensorFlow is not installed and the
reference ships no checkpoint).  What is pinned: CRC-32C known answers, the LevelDB magic / mask constants, and the
reader/writer round trip (tests/test_tf_checkpoint.py).  Snappy-compressed blocks (TensorFlow writes bundles
uncompressed, but the table format allows them) are decoded too.  Partitioned variables (`slices`) are rejected.
"""
import os
import struct

import numpy as np

from monopsr_amd import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
_BLOCK_TRAILER = 5
_FOOTER = 48
_RESTART_INTERVAL = 16
_BLOCK_SIZE = 256 << 10

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


class CheckpointError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------ primitives

def crc32c(data, crc=0):
    data = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    return _lib.lib().mpsr_crc32c(crc, bytes(data), len(data))


def _crc32c_array(arr):
    a = np.ascontiguousarray(arr)
    return _lib.lib().mpsr_crc32c(0, a.ctypes.data, a.nbytes)


def mask_crc(crc):
    """crc32c::Mask: rotate right by 15 bits and add a constant (so CRCs of data holding CRCs stay robust)."""
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def _put_varint(out, v):
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointError("varint too long")


def _snappy_uncompress(src):
    """Raw snappy block format (only needed for third-party tables; TensorFlow bundles are uncompressed)."""
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 2], "little")
            pos += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------------ protobuf

def _parse_fields(buf):
    """Minimal protobuf wire-format walk -> list of (field number, wire type, value)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wt)
        out.append((field, wt, v))
    return out


def _parse_shape(buf):
    dims = []
    for field, _, v in _parse_fields(buf):
        if field == 2:  # Dim
            size = 0
            for f2, _, v2 in _parse_fields(v):
                if f2 == 1:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(size)
        elif field == 3 and v:
            raise CheckpointError("tensor of unknown rank")
    return tuple(dims)


def _parse_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for field, _, v in _parse_fields(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            e["shape"] = _parse_shape(v)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif field == 7:
            e["slices"] += 1
    return e


def _encode_entry(dtype_id, shape, offset, size, crc_masked):
    shape_pb = bytearray()
    for d in shape:
        dim = bytearray([0x08])
        _put_varint(dim, d)
        shape_pb.append(0x12)
        _put_varint(shape_pb, len(dim))
        shape_pb += dim
    out = bytearray([0x08])
    _put_varint(out, dtype_id)
    out.append(0x12)
    _put_varint(out, len(shape_pb))
    out += shape_pb
    if offset:
        out.append(0x20)
        _put_varint(out, offset)
    out.append(0x28)
    _put_varint(out, size)
    out.append(0x35)
    out += struct.pack("<I", crc_masked)
    return bytes(out)


_HEADER_PB = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])  # num_shards: 1, endianness LITTLE (0), version{producer: 1}


# ------------------------------------------------------------------------------------------------ table reader

def _read_block(data, offset, size, verify):
    raw = data[offset:offset + size]
    trailer = data[offset + size:offset + size + _BLOCK_TRAILER]
    if len(raw) != size or len(trailer) != _BLOCK_TRAILER:
        raise CheckpointError("block handle outside the file")
    if verify:
        want = struct.unpack("<I", trailer[1:5])[0]
        if mask_crc(crc32c(raw + trailer[0:1])) != want:
            raise CheckpointError("block checksum mismatch at offset %d" % offset)
    if trailer[0] == 0:
        return raw
    if trailer[0] == 1:
        return _snappy_uncompress(raw)
    raise CheckpointError("unknown block compression type %d" % trailer[0])


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError("block too small")
    num_restarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise CheckpointError("bad restart array")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise CheckpointError("corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_index(index_path, verify=True):
    """-> (header bytes, {name: entry dict}) of a `.index` file."""
    with open(index_path, "rb") as f:
        data = f.read()
    if len(data) < _FOOTER:
        raise CheckpointError("%s is too short to be a checkpoint index" % index_path)
    footer = data[-_FOOTER:]
    if struct.unpack("<Q", footer[-8:])[0] != TABLE_MAGIC:
        raise CheckpointError("%s: bad table magic (not a TensorFlow V2 checkpoint index)" % index_path)
    _, pos = _get_varint(footer, 0)          # metaindex handle
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)     # index handle
    isize, pos = _get_varint(footer, pos)
    header, entries = None, {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p2 = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p2)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b"":
                header = value
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    if header is None:
        raise CheckpointError("%s: no bundle header entry" % index_path)
    return header, entries


def _header_info(header):
    num_shards, endianness = 1, 0
    for field, _, v in _parse_fields(header):
        if field == 1:
            num_shards = v
        elif field == 2:
            endianness = v
    if endianness != 0:
        raise CheckpointError("big-endian bundles are not supported")
    return num_shards


def resolve_prefix(path):
    """Accepts a checkpoint prefix, a `.index` / `.data-*` file, or a directory holding a `checkpoint` state file
    (the text file tf.train.Saver maintains: `model_checkpoint_path: "<prefix>"`)."""
    if os.path.isdir(path):
        state = os.path.join(path, "checkpoint")
        if not os.path.exists(state):
            raise CheckpointError("%s has no 'checkpoint' state file" % path)
        for line in open(state):
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                return p if os.path.isabs(p) else os.path.join(path, p)
        raise CheckpointError("%s names no model_checkpoint_path" % state)
    if path.endswith(".index"):
        return path[:-len(".index")]
    if ".data-" in os.path.basename(path):
        return path[:path.rindex(".data-")]
    return path


def list_variables(path):
    """[(name, shape)] like tf.train.list_variables."""
    _, entries = read_index(resolve_prefix(path) + ".index")
    return [(k, list(e["shape"])) for k, e in sorted(entries.items())]


def read_checkpoint(path, names=None, verify=True):
    """{variable name: numpy array} for every (or the named) variable of the checkpoint at `path`."""
    prefix = resolve_prefix(path)
    header, entries = read_index(prefix + ".index", verify)
    num_shards = _header_info(header)
    wanted = sorted(entries) if names is None else list(names)
    shards, out = {}, {}
    try:
        for name in wanted:
            if name not in entries:
                raise CheckpointError("variable %r is not in the checkpoint" % name)
            e = entries[name]
            if e["slices"]:
                raise CheckpointError("variable %r is partitioned; sliced entries are not supported" % name)
            if e["dtype"] not in _DTYPES:
                raise CheckpointError("variable %r has unsupported dtype enum %d" % (name, e["dtype"]))
            dt = np.dtype(_DTYPES[e["dtype"]])
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if count * dt.itemsize != e["size"]:
                raise CheckpointError("variable %r: %d bytes recorded for shape %s" % (name, e["size"], e["shape"]))
            sid = e["shard_id"]
            if sid not in shards:
                shards[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "rb")
            f = shards[sid]
            f.seek(e["offset"])
            arr = np.fromfile(f, dtype=dt, count=count)
            if arr.size != count:
                raise CheckpointError("variable %r: data file is truncated" % name)
            if verify and e["crc32c"] is not None and mask_crc(_crc32c_array(arr)) != e["crc32c"]:
                raise CheckpointError("variable %r: payload checksum mismatch" % name)
            out[name] = arr.reshape(e["shape"])
    finally:
        for f in shards.values():
            f.close()
    return out


# ------------------------------------------------------------------------------------------------ table writer

class _BlockBuilder:
    def __init__(self, restart_interval=None):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""
        self.interval = _RESTART_INTERVAL if restart_interval is None else restart_interval

    def add(self, key, value):
        shared = 0
        if self.count and self.count % self.interval == 0:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:]
        self.buf += value
        self.last = key
        self.count += 1

    def finish(self):
        out = bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) \
            + struct.pack("<I", len(self.restarts))
        return out

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _shortest_separator(start, limit):
    """A short key k with start <= k < limit (bytewise order), as the table builder puts between two data blocks."""
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xff and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _short_successor(key):
    """A short key >= key: the index entry of the last data block."""
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def _write_block(f, contents):
    offset = f.tell()
    f.write(contents)
    f.write(b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00"))))
    handle = bytearray()
    _put_varint(handle, offset)
    _put_varint(handle, len(contents))
    return bytes(handle)


def write_checkpoint(prefix, tensors, block_size=_BLOCK_SIZE):
    """Write {name: array} as `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, uncompressed blocks,
    entries in key order, as tensor_bundle's BundleWriter lays them out).  Returns the prefix."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    items = []
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in names:
            if name == "":
                raise CheckpointError("the empty name is reserved for the bundle header")
            shape = np.shape(tensors[name])  # ascontiguousarray would turn a scalar into shape (1,)
            a = np.ascontiguousarray(tensors[name])
            if a.dtype not in _DTYPE_IDS:
                raise CheckpointError("variable %r: dtype %s cannot be stored" % (name, a.dtype))
            off = f.tell()
            a.tofile(f)
            items.append((name.encode("utf-8"),
                          _encode_entry(_DTYPE_IDS[a.dtype], shape, off, a.nbytes, mask_crc(_crc32c_array(a)))))
    with open(prefix + ".index", "wb") as f:
        # as TensorFlow's table builder: data blocks restart every 16 entries, the index block at every entry, and an
        # index key is the shortest separator between a block's last key and the next block's first one
        index = _BlockBuilder(restart_interval=1)
        block = _BlockBuilder()
        block.add(b"", _HEADER_PB)
        pending = None  # (last key, handle) of the block just written, waiting for its successor's first key

        def flush():
            nonlocal block, pending
            if block.count:
                pending = (block.last, _write_block(f, block.finish()))
                block = _BlockBuilder()
        for key, value in items:
            if pending is not None and block.count == 0:
                index.add(_shortest_separator(pending[0], key), pending[1])
                pending = None
            block.add(key, value)
            if block.size() >= block_size:
                flush()
        flush()
        if pending is not None:
            index.add(_short_successor(pending[0]), pending[1])
        meta_handle = _write_block(f, _BlockBuilder().finish())
        index_handle = _write_block(f, index.finish())
        footer = meta_handle + index_handle
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    return prefix


def write_checkpoint_state(directory, prefix_basename):
    """The `checkpoint` text file tf.train.Saver keeps next to its bundles."""
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (prefix_basename, prefix_basename))
