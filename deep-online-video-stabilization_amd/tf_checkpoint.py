"""TensorFlow checkpoint reader / writer without TensorFlow (first half): V2 "tensor bundle" and the older
V1 single-file format.

The reference restores `models/v2_93/model-80000` through `tf.train.Saver` (deploy_bundle.py:45-47) and initialises training
from `data_video/resnet_v2_50.ckpt` (train_bundle_nobm.py:184-191).  Neither TF nor any checkpoint exists in the build image, so
this module restates the PUBLISHED on-disk format from its specification and is pinned only by its own round trip
(`tests/test_tf_checkpoint_cpu.py`) -- parity against a checkpoint written by real TF is UNPINNED and said so here:

  <prefix>.index                 a leveldb-format sorted string table (table/format.cc): data blocks of prefix-compressed
                                 (shared, non_shared, value_len) entries + restart array, an index block, a 48-byte footer
                                 with magic 0xdb4775248b80fb57.  Key "" -> BundleHeaderProto, key <variable name> ->
                                 BundleEntryProto {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c}
  <prefix>.data-00000-of-00001   raw little-endian tensor bytes at [offset, offset + size)

  V1 `<prefix>` (one file; the format slim distributes the ImageNet `resnet_v2_50.ckpt` in -- the checkpoint the reference
                                 REQUIRES for its warm start, train_bundle_nobm.py:184-191,208): the same leveldb-format table
                                 (tensorflow/core/util/tensor_slice_writer.cc).  Key "" -> SavedTensorSlices{1: meta =
                                 SavedTensorSliceMeta{1: repeated SavedSliceMeta{1: name, 2: shape, 3: dtype, 4: slices}}};
                                 every other key -> SavedTensorSlices{2: data = SavedSlice{1: name, 2: TensorSliceProto,
                                 3: TensorProto}} with the values in the TYPED repeated field of the TensorProto (float_val = 5,
                                 double_val = 6, int_val = 7, int64_val = 10; packed or not) or in tensor_content = 4.

Only what the path needs is supported: table blocks uncompressed or Snappy, one shard or several, dtypes
float32/float64/int32/int64; V2: no sliced (partitioned) variables; V1: slices are assembled into the full tensor.
Anything else raises with a clear message.
"""
from __future__ import annotations

import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}          # tensorflow/core/framework/types.proto
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------------------ varints / protobuf wire
def _get_varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
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
    """-> {field: [values]}; varint fields as int, fixed32 as int, length-delimited as bytes."""
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
            raise ValueError("tf_checkpoint: unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ------------------------------------------------------------------------------------------------ crc32c (Castagnoli)
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data: bytes, crc: int = 0) -> int:
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c):                       # leveldb / TF crc32c::Mask
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ table reader
def _snappy_decompress(src: bytes) -> bytes:
    """Raw Snappy block format (format_description.txt): varint uncompressed length, then literal / copy elements."""
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                            # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little"); pos += nb
            ln += 1
            out += src[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], "little"); pos += 4
        if off == 0 or off > len(out):
            raise ValueError("tf_checkpoint: corrupt Snappy block")
        for _ in range(ln):                                      # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("tf_checkpoint: Snappy block length mismatch")
    return bytes(out)


def _read_block(buf, offset, size):
    ctype = buf[offset + size]
    if ctype == 1:
        blk = _snappy_decompress(bytes(buf[offset:offset + size]))
    elif ctype == 0:
        blk = buf[offset:offset + size]
    else:
        raise ValueError("tf_checkpoint: table block with unknown compression type %d" % ctype)
    nrestart = struct.unpack_from("<I", blk, len(blk) - 4)[0]
    end = len(blk) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(blk, pos)
        non_shared, pos = _get_varint(blk, pos)
        vlen, pos = _get_varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(blk[pos:pos + vlen]))); pos += vlen
    return out


def _read_table(path):
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("tf_checkpoint: %s is not a tensor-bundle index (bad magic)" % path)
    footer = buf[len(buf) - 48:]
    _, p = _get_varint(footer, 0); _, p = _get_varint(footer, p)             # metaindex handle (unused)
    ioff, p = _get_varint(footer, p); isize, p = _get_varint(footer, p)
    entries = []
    for _, handle in _read_block(buf, ioff, isize):
        off, q = _get_varint(handle, 0)
        size, q = _get_varint(handle, q)
        entries.extend(_read_block(buf, off, size))
    return entries


def read_bundle(prefix: str, verify_crc_below: int = 1 << 20) -> dict:
    """All variables of the checkpoint `<prefix>` as {name: ndarray}.  Tensor CRCs are verified for tensors smaller than
    `verify_crc_below` bytes (the pure-Python CRC is slow; pass a large value to verify everything)."""
    index = prefix + ".index"
    if not os.path.exists(index):
        raise FileNotFoundError(index)
    entries = _read_table(index)
    if not entries or entries[0][0] != b"":
        raise ValueError("tf_checkpoint: missing bundle header")
    header = _parse_proto(entries[0][1])
    num_shards = header.get(1, [1])[0]
    if header.get(2, [0])[0] != 0:
        raise ValueError("tf_checkpoint: big-endian bundle")
    shards = {}
    out = {}
    for key, val in entries[1:]:
        e = _parse_proto(val)
        if 7 in e:
            raise ValueError("tf_checkpoint: variable %r is sliced (partitioned variables are not supported)" % key.decode())
        dt = e.get(1, [0])[0]
        if dt not in _DTYPES:
            raise ValueError("tf_checkpoint: variable %r has unsupported dtype id %d" % (key.decode(), dt))
        dims = []
        if 2 in e:
            for d in _parse_proto(e[2][0]).get(2, []):
                dims.append(_signed64(_parse_proto(d).get(1, [0])[0]))
        shard = e.get(3, [0])[0]
        off, size = e.get(4, [0])[0], e.get(5, [0])[0]
        if shard not in shards:
            shards[shard] = np.memmap("%s.data-%05d-of-%05d" % (prefix, shard, num_shards), dtype=np.uint8, mode="r")
        raw = shards[shard][off:off + size]
        if len(raw) != size:
            raise ValueError("tf_checkpoint: data shard truncated at %r" % key.decode())
        if size < verify_crc_below and 6 in e:
            if _mask_crc(crc32c(bytes(raw))) != e[6][0]:
                raise ValueError("tf_checkpoint: crc mismatch in %r" % key.decode())
        arr = np.frombuffer(bytes(raw), dtype=_DTYPES[dt])
        out[key.decode()] = arr.reshape(dims) if dims else arr.reshape(())
    return out


# ------------------------------------------------------------------------------------------------ V1 (single file)
_V1_VALUE_FIELD = {1: (5, "<f4"), 2: (6, "<f8"), 3: (7, None), 9: (10, None)}      # dtype id -> (TensorProto field, fixed layout or varints)


def _tensor_proto_values(tp: dict, dt: int, name: str) -> np.ndarray:
    """The flat values of a TensorProto as TensorSliceWriter fills it: tensor_content, or the typed repeated field (packed:
    one length-delimited run; unpacked: one fixed32/64 or varint per element)."""
    np_dt = _DTYPES[dt]
    if 4 in tp and tp[4][0]:
        return np.frombuffer(tp[4][0], dtype=np.dtype(np_dt).newbyteorder("<")).astype(np_dt)
    field, layout = _V1_VALUE_FIELD[dt]
    vals = tp.get(field, [])
    if not vals:
        return np.zeros(0, np_dt)
    if isinstance(vals[0], (bytes, bytearray)):                  # packed
        raw = b"".join(vals)
        if layout is not None:
            return np.frombuffer(raw, dtype=layout).astype(np_dt)
        out, pos = [], 0
        while pos < len(raw):
            v, pos = _get_varint(raw, pos)
            out.append(_signed64(v))
        return np.asarray(out, np_dt)
    if layout == "<f4":                                          # unpacked fixed32 came back as ints
        return np.asarray(vals, np.uint32).view(np.float32).astype(np_dt)
    if layout == "<f8":
        return np.asarray(vals, np.uint64).view(np.float64).astype(np_dt)
    return np.asarray([_signed64(v) for v in vals], np_dt)


def _shape_of(shape_proto: bytes):
    return [_signed64(_parse_proto(d).get(1, [0])[0]) for d in _parse_proto(shape_proto).get(2, [])]


def read_v1(path: str) -> dict:
    """All variables of a V1 checkpoint file as {name: ndarray} (slices of partitioned variables assembled)."""
    entries = _read_table(path)
    if not entries or entries[0][0] != b"":
        raise ValueError("tf_checkpoint: %s has no SavedTensorSlices meta entry (not a V1 checkpoint)" % path)
    meta = _parse_proto(_parse_proto(entries[0][1]).get(1, [b""])[0])
    shapes, dtypes = {}, {}
    for t in meta.get(1, []):
        m = _parse_proto(t)
        name = m[1][0].decode()
        dt = m.get(3, [0])[0]
        if dt not in _DTYPES:
            raise ValueError("tf_checkpoint: variable %r has unsupported dtype id %d" % (name, dt))
        shapes[name] = _shape_of(m[2][0]) if 2 in m else []
        dtypes[name] = dt
    out, filled = {}, {}
    for key, val in entries[1:]:
        top = _parse_proto(val)
        if 2 not in top:
            continue
        sl = _parse_proto(top[2][0])
        name = sl[1][0].decode()
        if name not in shapes:
            raise ValueError("tf_checkpoint: slice of %r without a meta entry" % name)
        shape, dt = shapes[name], dtypes[name]
        tp = _parse_proto(sl[3][0]) if 3 in sl else {}
        vals = _tensor_proto_values(tp, dt, name)
        # TensorSliceProto: one Extent per dimension; an empty Extent = the whole dimension
        ext = [_parse_proto(e) for e in _parse_proto(sl[2][0]).get(1, [])] if 2 in sl else []
        idx = []
        for d, full in enumerate(shape):
            e = ext[d] if d < len(ext) else {}
            start = _signed64(e.get(1, [0])[0])
            length = _signed64(e[2][0]) if 2 in e else full - start
            idx.append(slice(start, start + length))
        if name not in out:
            out[name] = np.zeros(shape, _DTYPES[dt])
            filled[name] = 0
        view_shape = [i.stop - i.start for i in idx]
        if int(np.prod(view_shape, dtype=np.int64)) != vals.size:
            raise ValueError("tf_checkpoint: slice of %r holds %d values, its extent %s needs %d" % (name, vals.size, view_shape, int(np.prod(view_shape))))
        out[name][tuple(idx)] = vals.reshape(view_shape)
        filled[name] += vals.size
    for name, shape in shapes.items():
        if filled.get(name, -1) != int(np.prod(shape, dtype=np.int64)):
            raise ValueError("tf_checkpoint: variable %r is incomplete (%d of %d values)" % (name, filled.get(name, 0), int(np.prod(shape))))
    return out


def write_v1(path: str, variables: dict, packed: bool = True, slices_of: dict = None):
    """Writes {name: ndarray} as a V1 checkpoint file (test fixture generator: the same table + SavedTensorSlices layout
    TensorSliceWriter produces; keys are 0x00 + name + slice text, which sorts like TF's ordered code for one slice per tensor).
    slices_of: {name: n} splits that variable along axis 0 into n slices (a partitioned variable)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def lend(b):
        return _put_varint(len(b)) + b

    def shape_proto(shape):
        return b"".join(_field(2, 2, lend(_field(1, 0, _put_varint(int(d))))) for d in shape)

    metas, datas = [], []
    for name in sorted(variables, key=lambda s: s.encode()):
        a = np.asarray(variables[name])
        a = a if a.flags.c_contiguous else a.copy()              # (ascontiguousarray would promote scalars to 1-D)
        if a.dtype not in _DTYPE_IDS:
            raise ValueError("tf_checkpoint: dtype %s of %r is not supported" % (a.dtype, name))
        dt = _DTYPE_IDS[a.dtype]
        nsl = (slices_of or {}).get(name, 1)
        bounds = np.linspace(0, a.shape[0] if a.ndim else 1, nsl + 1).astype(int) if nsl > 1 else None
        slice_protos = []
        for k in range(nsl):
            if nsl == 1:
                part, sp = a, b"".join(_field(1, 2, lend(b"")) for _ in a.shape)
            else:
                lo, hi = int(bounds[k]), int(bounds[k + 1])
                part = a[lo:hi]
                sp = _field(1, 2, lend(_field(1, 0, _put_varint(lo)) + _field(2, 0, _put_varint(hi - lo))))
                sp += b"".join(_field(1, 2, lend(b"")) for _ in a.shape[1:])
            slice_protos.append(sp)
            field, layout = _V1_VALUE_FIELD[dt]
            flat = part.ravel()
            if layout is not None:
                raw = flat.astype(layout).tobytes()
                body = _field(field, 2, lend(raw)) if packed else b"".join(
                    _field(field, 5 if layout == "<f4" else 1, raw[i:i + (4 if layout == "<f4" else 8)])
                    for i in range(0, len(raw), 4 if layout == "<f4" else 8))
            else:
                vs = b"".join(_put_varint(int(v) & ((1 << 64) - 1)) for v in flat)
                body = _field(field, 2, lend(vs)) if packed else b"".join(_field(field, 0, _put_varint(int(v) & ((1 << 64) - 1))) for v in flat)
            tp = _field(1, 0, _put_varint(dt)) + _field(2, 2, lend(shape_proto(part.shape))) + body
            saved = _field(1, 2, lend(name.encode())) + _field(2, 2, lend(sp)) + _field(3, 2, lend(tp))
            datas.append((b"\x00" + name.encode() + b"\x00\x01" + bytes([k]), _field(2, 2, lend(saved))))
        m = _field(1, 2, lend(name.encode())) + _field(2, 2, lend(shape_proto(a.shape))) + _field(3, 0, _put_varint(dt))
        m += b"".join(_field(4, 2, lend(sp)) for sp in slice_protos)
        metas.append(_field(1, 2, lend(m)))
    items = [(b"", _field(1, 2, lend(b"".join(metas))))] + sorted(datas)
    with open(path, "wb") as f:
        index_items, cur, cur_bytes = [], [], 0
        for kv in items:
            cur.append(kv); cur_bytes += len(kv[0]) + len(kv[1]) + 3
            if cur_bytes >= 4096:
                off, size = _emit_block(f, _build_block(cur))
                index_items.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
                cur, cur_bytes = [], 0
        if cur:
            off, size = _emit_block(f, _build_block(cur))
            index_items.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
        moff, msize = _emit_block(f, _build_block([]))
        ioff, isize = _emit_block(f, _build_block(index_items, restart_interval=1))
        footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))


# ------------------------------------------------------------------------------------------------ table writer
def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out)); shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", max(len(restarts), 1))
    return bytes(out)


def _emit_block(f, contents):
    off = f.tell()
    f.write(contents)
    f.write(b"\x00")
    f.write(struct.pack("<I", _mask_crc(crc32c(b"\x00", crc32c(contents)))))
    return off, len(contents)


def _field(num, wt, payload):
    return _put_varint((num << 3) | wt) + payload


def write_bundle(prefix: str, variables: dict, block_size: int = 4096):
    """Writes {name: ndarray} as a single-shard V2 checkpoint `<prefix>.index` + `<prefix>.data-00000-of-00001`."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = [(b"", _field(1, 0, _put_varint(1)) + _field(3, 2, _put_varint(2) + _field(1, 0, _put_varint(1))))]   # num_shards=1, version.producer=1
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for name in sorted(variables, key=lambda s: s.encode()):
            a = np.asarray(variables[name])
            a = a if a.flags.c_contiguous else a.copy()          # (ascontiguousarray would promote scalars to 1-D)
            if a.dtype not in _DTYPE_IDS:
                raise ValueError("tf_checkpoint: dtype %s of %r is not supported" % (a.dtype, name))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            shape = b"".join(_field(2, 2, (lambda d: _put_varint(len(d)) + d)(_field(1, 0, _put_varint(int(s))))) for s in a.shape)
            entry = _field(1, 0, _put_varint(_DTYPE_IDS[a.dtype])) + _field(2, 2, _put_varint(len(shape)) + shape)
            entry += _field(4, 0, _put_varint(data.tell())) + _field(5, 0, _put_varint(len(raw)))
            entry += _field(6, 5, struct.pack("<I", _mask_crc(crc32c(raw))))
            items.append((name.encode(), entry))
            data.write(raw)
    with open(prefix + ".index", "wb") as f:
        index_items, cur, cur_bytes = [], [], 0
        for kv in items:
            cur.append(kv); cur_bytes += len(kv[0]) + len(kv[1]) + 3
            if cur_bytes >= block_size:
                off, size = _emit_block(f, _build_block(cur))
                index_items.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
                cur, cur_bytes = [], 0
        if cur:
            off, size = _emit_block(f, _build_block(cur))
            index_items.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
        moff, msize = _emit_block(f, _build_block([]))
        ioff, isize = _emit_block(f, _build_block(index_items, restart_interval=1))
        footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))


# ------------------------------------------------------------------------------------------------ StabNet variable names
STABNET_SCOPE = "stable_net/resnet/"       # s_net_bundle_nobm.py:251 (`with tf.variable_scope('resnet')` under 'stable_net')


def load_stabnet_variables(prefix: str):
    """(params, extras): params = the regressor's variables keyed the way NetPlan.pack expects (scope stripped); extras = Adam
    slots (`.../Adam`, `.../Adam_1`), `beta1_power`, `beta2_power`, `global_step` and anything else found."""
    allv = read_bundle(prefix)
    params, extras = {}, {}
    for k, v in allv.items():
        name = k[len(STABNET_SCOPE):] if k.startswith(STABNET_SCOPE) else k
        if k.startswith(STABNET_SCOPE) and not (name.endswith("/Adam") or name.endswith("/Adam_1")):
            params[name] = np.asarray(v, np.float32)
        else:
            extras[k] = v
    return params, extras


def checkpoint_format(prefix: str) -> str:
    """'v2' : `<prefix>.index` is a tensor-bundle table (what tf.train.Saver writes since TF 1.0 and what this module reads);
    'v1' : `<prefix>` itself is a table file -- the older single-file format slim distributes `resnet_v2_50.ckpt` in
           (SavedTensorSlices values; read by read_v1);
    'none': neither exists;  'unknown': a file exists but is not a table."""
    def is_table(path):
        try:
            with open(path, "rb") as f:
                f.seek(0, os.SEEK_END)
                if f.tell() < 48:
                    return False
                f.seek(-8, os.SEEK_END)
                return struct.unpack("<Q", f.read(8))[0] == _MAGIC
        except OSError:
            return False
    if os.path.isfile(prefix + ".index"):
        return "v2" if is_table(prefix + ".index") else "unknown"
    if os.path.isfile(prefix):
        return "v1" if is_table(prefix) else "unknown"
    return "none"


def try_load_imagenet_resnet(prefix: str):
    """-> (variables or None, note).  None when the file is missing or is not a TensorFlow checkpoint; the reference cannot
    start without this warm start (restorer.restore, train_bundle_nobm.py:208), so the training driver treats None as an error
    unless --no-imagenet-init is given."""
    fmt = checkpoint_format(prefix)
    if fmt in ("v1", "v2"):
        return load_imagenet_resnet(prefix), "initialised from the %s checkpoint %s" % (fmt.upper(), prefix)
    if fmt == "unknown":
        return None, "%s exists but is not a TensorFlow checkpoint table" % prefix
    return None, "%s not found" % prefix


def imagenet_expected_names(init: dict) -> list:
    """The variables `restorer.restore` fills (train_bundle_nobm.py:185-191): everything under `stable_net/resnet/` except
    `resnet_v2_50/conv1` (13 input channels), the `fc` head, Adam slots and `gen_theta` -- in this build's names (scope stripped)
    every `resnet_v2_50/*` entry of a fresh parameter set but `resnet_v2_50/conv1/*`."""
    return sorted(k for k in init if k.startswith("resnet_v2_50/") and not k.startswith("resnet_v2_50/conv1/"))


def apply_imagenet_init(init: dict, pre: dict) -> int:
    """Copies the warm-start variables over `init` in place; like `tf.train.Saver(vtr).restore` it FAILS on any expected variable
    the checkpoint lacks or holds with another shape (a wrong or truncated checkpoint must not silently become training from
    the seeded initialiser).  -> number of variables initialised."""
    want = imagenet_expected_names(init)
    missing = [k for k in want if k not in pre]
    misshaped = ["%s: checkpoint %s, model %s" % (k, tuple(pre[k].shape), tuple(init[k].shape))
                 for k in want if k in pre and tuple(pre[k].shape) != tuple(init[k].shape)]
    if missing or misshaped:
        def head(lst):
            return ", ".join(lst[:8]) + (" ... (%d in all)" % len(lst) if len(lst) > 8 else "")
        raise ValueError("tf_checkpoint: the ImageNet checkpoint does not cover the backbone: "
                         + ("missing " + head(missing) if missing else "")
                         + ("; " if missing and misshaped else "")
                         + ("wrong shape " + head(misshaped) if misshaped else ""))
    for k in want:
        init[k] = np.asarray(pre[k], np.float32)
    return len(want)


def read_checkpoint(prefix: str) -> dict:
    """{name: ndarray} of a V2 bundle (`<prefix>.index`) or a V1 single-file checkpoint (`<prefix>`)."""
    fmt = checkpoint_format(prefix)
    if fmt == "v2":
        return read_bundle(prefix)
    if fmt == "v1":
        return read_v1(prefix)
    raise FileNotFoundError("tf_checkpoint: no TensorFlow checkpoint at %s (%s)" % (prefix, fmt))


def load_imagenet_resnet(prefix: str):
    """train_bundle_nobm.py:184-191,101-102: the ImageNet `resnet_v2_50.ckpt` initialises every variable under
    `stable_net/resnet/` except `resnet_v2_50/conv1` (13 input channels instead of 3) and the `fc` head; a variable
    `stable_net/resnet/<name>` is read from the checkpoint's `<name>` (name_in_checkpoint strips the 18-character scope).
    -> {name without scope: array} of the variables to copy over a fresh initialisation."""
    out = {}
    for k, v in read_checkpoint(prefix).items():
        if not k.startswith("resnet_v2_50/") or k.startswith("resnet_v2_50/conv1/") or "logits" in k:
            continue
        if "Adam" in k or k.endswith("/Momentum"):
            continue
        out[k] = np.asarray(v, np.float32)
    return out
