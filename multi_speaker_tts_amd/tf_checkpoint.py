"""Reader / writer for TensorFlow checkpoints in the V2 "tensor bundle" format, without TensorFlow.

The reference saves and restores its three variable scopes with tf.train.Saver (MSTTS_SV.py:30-40,223-251): a text file
`checkpoint` naming the latest prefix, `<prefix>.index` and `<prefix>.data-00000-of-00001`.  No TensorFlow exists in this
environment, so the published on-disk format is restated here:

* `.index` is a LevelDB-format sorted table (tensorflow/core/lib/io/table, a port of LevelDB's table): data blocks of
  prefix-compressed (key, value) entries with a restart array, an index block, and a 48-byte footer ending in the magic
  0xdb4775248b80fb57; each block is followed by a 1-byte compression tag (0 = none, 1 = snappy) and a masked CRC-32C.
  Key "" holds a BundleHeaderProto, every other key is a variable name holding a BundleEntryProto
  {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
* `.data-SSSSS-of-NNNNN` holds the raw little-endian tensor bytes at (offset, size).

PARITY UNPINNED: no real checkpoint of the reference is available here; the reader is tested against this module's own
writer (which follows the same specification) and against hand-assembled table bytes.
"""
from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16,
          22: np.uint32, 23: np.uint64}
DTYPE_CODES = {np.dtype(v): k for k, v in DTYPES.items()}


# ---- CRC-32C (Castagnoli), masked as in LevelDB ------------------------------------------------------------------------
def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC = _crc_table()


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _CRC[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- varints / protobuf wire format ------------------------------------------------------------------------------------------
def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """{field: [values]}: varints as int, length-delimited as bytes, fixed32/64 as int."""
    out, pos = {}, 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(buf):
    p = _parse_proto(buf)
    shape = []
    if 2 in p:
        for dim in _parse_proto(p[2][0]).get(2, []):
            shape.append(_signed64(_parse_proto(dim).get(1, [0])[0]))
    return {"dtype": p.get(1, [0])[0], "shape": tuple(shape), "shard_id": p.get(3, [0])[0], "offset": p.get(4, [0])[0], "size": p.get(5, [0])[0],
            "crc32c": p.get(6, [None])[0], "sliced": 7 in p}


def _entry_proto(dtype_code, shape, shard_id, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype_code) + b"\x12" + _put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


# ---- snappy (raw format) decoder, in case an index block is compressed ------------------------------------------------------
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little"); pos += nb
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little"); pos += 4
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ---- table ---------------------------------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if mask_crc(crc32c(data[offset:offset + size + 1])) != stored:
            raise ValueError("table block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        raw = _snappy_decompress(raw)
    elif ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return raw


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a sorted-table file (bad magic)" % path)
    footer = data[-48:]
    _, p = _get_varint(footer, 0); _, p = _get_varint(footer, p)           # metaindex handle
    ioff, p = _get_varint(footer, p); isize, p = _get_varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0); bsize, _ = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


def _build_block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, items, block_size=4096):
    """items: sorted [(key bytes, value bytes)]."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block + b"\x00")
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)
    index, cur, cur_size = [], [], 0
    for k, v in items:
        cur.append((k, v)); cur_size += len(k) + len(v) + 3
        if cur_size >= block_size:
            off, size = emit(_build_block(cur))
            index.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
            cur, cur_size = [], 0
    if cur or not index:
        off, size = emit(_build_block(cur))
        index.append((cur[-1][0] if cur else b"", _put_varint(off) + _put_varint(size)))
    moff, msize = emit(_build_block([]))
    ioff, isize = emit(_build_block(index, restart_interval=1))
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---- bundle -------------------------------------------------------------------------------------------------------------------
def list_variables(prefix, verify=True):
    """{name: entry dict} of a checkpoint prefix (the part before '.index')."""
    out = {}
    for k, v in read_table(prefix + ".index", verify):
        if k == b"":
            hdr = _parse_proto(v)
            if hdr.get(2, [0])[0] != 0:
                raise ValueError("big-endian bundles are not supported")
            out["__num_shards__"] = hdr.get(1, [1])[0]
            continue
        out[k.decode("utf-8")] = _parse_entry(v)
    return out


def read_checkpoint(prefix, names=None, verify=True):
    """{variable name: ndarray}.  `names`: iterable restricting what is loaded (missing names are ignored)."""
    entries = list_variables(prefix, verify)
    shards = entries.pop("__num_shards__", 1)
    want = set(names) if names is not None else None
    files, out = {}, {}
    for name, e in entries.items():
        if want is not None and name not in want:
            continue
        if e["sliced"]:
            raise ValueError("variable '%s' is stored in slices (partitioned variable): not supported" % name)
        if e["dtype"] not in DTYPES:
            continue                                    # strings / resources: nothing the model needs
        sid = e["shard_id"]
        if sid not in files:
            files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, shards), "rb")
        f = files[sid]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if verify and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError("tensor '%s': checksum mismatch" % name)
        out[name] = np.frombuffer(raw, dtype=np.dtype(DTYPES[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    for f in files.values():
        f.close()
    return out


def write_checkpoint(prefix, variables):
    """Single-shard bundle: `<prefix>.index` + `<prefix>.data-00000-of-00001`, and the `checkpoint` state file next to them."""
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    items, offset = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(variables, key=lambda s: s.encode("utf-8")):
            a = np.asarray(variables[name])
            if not a.flags.c_contiguous:               # (np.ascontiguousarray would turn a scalar into shape (1,))
                a = a.copy(order="C")
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            raw = a.tobytes()
            f.write(raw)
            items.append((name.encode("utf-8"), _entry_proto(DTYPE_CODES[np.dtype(a.dtype.name)], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"            # num_shards = 1, version { producer: 1 }
    write_table(prefix + ".index", [(b"", header)] + items)
    base = os.path.basename(prefix)
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by the `checkpoint` state file, or None."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as f:
        for line in f:
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                p = p if os.path.isabs(p) else os.path.join(directory, p)
                return p if os.path.exists(p + ".index") else None
    return None
