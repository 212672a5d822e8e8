"""Reader for TensorFlow-1 checkpoint bundles (`snap-N.index` + `snap-N.data-00000-of-00001`) without TensorFlow.

The reference restores its weights with tf.train.Saver (utils/tester.py:155-162, demo_registration.py:125-135); the
released snapshots are `results/Log_*/snapshots/snap-*.{index,meta,data-00000-of-00001}`.  This module parses the
bundle directly so that the released weights drop into `KernelPointFCNN(weights=...)`:

    entries = read_index(path_prefix + '.index')      # {variable name: BundleEntry(dtype, shape, shard, offset, size, crc)}
    weights = load_checkpoint(path_prefix)             # {name without 'KernelPointNetwork/' root: float32 ndarray}

Format (tensorflow/core/util/tensor_bundle + the LevelDB table format it embeds):
  * `.index` is an SSTable: footer = last 48 bytes (metaindex handle, index handle as varint64 pairs, padding, 8-byte
    magic 0xdb4775248b80fb57); a block = prefix-compressed entries `varint shared | varint non_shared | varint
    value_len | key suffix | value`, then uint32 restart offsets + uint32 restart count; every block is followed by a
    1-byte compression type (0 = none, the only one TF writes here) and a 4-byte masked crc32c;
  * key "" holds the BundleHeaderProto, every other key is a variable name and its value a BundleEntryProto:
    1 dtype (varint), 2 shape {2 dim {1 size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c (fixed32);
  * `.data-00000-of-00001` holds the raw little-endian tensor bytes back to back at [offset, offset+size).
Training-only slots (optimizer `/Momentum`, global step ...) are skipped by `load_checkpoint`.
"""
import collections
import os
import struct

import numpy as np

BundleEntry = collections.namedtuple('BundleEntry', 'dtype shape shard_id offset size crc32c')

_MAGIC = 0xdb4775248b80fb57
_DT_FLOAT, _DT_INT32, _DT_INT64 = 1, 3, 9
_NP_OF_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}
ROOT_SCOPE = 'KernelPointNetwork/'


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not (b & 0x80):
            return out, pos
        shift += 7


def _read_block(data, offset, size):
    """-> list of (key bytes, value bytes) of one SSTable block."""
    if data[offset + size] != 0:
        raise ValueError('compressed SSTable block (type %d) is not supported' % data[offset + size])
    blk = data[offset:offset + size]
    n_restarts = struct.unpack_from('<I', blk, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(blk[pos:pos + vlen])))
        pos += vlen
    return out


def _parse_proto(buf):
    """Minimal protobuf wire parser -> list of (field number, wire type, value)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((fn, wt, v))
    return out


def _parse_entry(value):
    dtype = shard = offset = size = crc = 0
    shape = []
    for fn, wt, v in _parse_proto(value):
        if fn == 1:
            dtype = v
        elif fn == 2:
            for f2, _, dim in _parse_proto(v):
                if f2 == 2:
                    sz = 0
                    for f3, _, x in _parse_proto(dim):
                        if f3 == 1:
                            sz = x if x < (1 << 63) else x - (1 << 64)
                    shape.append(int(sz))
        elif fn == 3:
            shard = v
        elif fn == 4:
            offset = v
        elif fn == 5:
            size = v
        elif fn == 6:
            crc = v
    return BundleEntry(dtype, tuple(shape), shard, offset, size, crc)


def read_index(index_path):
    """-> OrderedDict {variable name: BundleEntry} in file (= lexicographic) order."""
    with open(index_path, 'rb') as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != _MAGIC:
        raise ValueError('%s is not a TensorFlow checkpoint index (bad SSTable magic)' % index_path)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)       # metaindex offset
    _, pos = _varint(footer, pos)       # metaindex size
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    entries = collections.OrderedDict()
    for _, handle in _read_block(data, ioff, isize):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, value in _read_block(data, boff, bsize):
            if key == b'':
                continue                 # BundleHeaderProto
            entries[key.decode()] = _parse_entry(value)
    return entries


# ---- crc32c (Castagnoli) + TF's mask, to verify tensor payloads ---------------------------------------------------
_CRC_TABLE = None


def _crc32c(buf):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl[i] = c
        _CRC_TABLE = tbl
    c = 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in memoryview(buf).cast('B'):
        c = int(tbl[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(buf):
    c = _crc32c(buf)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def is_model_variable(name):
    """Inference variables only: drop optimizer slots and bookkeeping scalars."""
    if not name.startswith(ROOT_SCOPE):
        return False
    tail = name.rsplit('/', 1)[-1]
    return tail in ('weights', 'kernel_points', 'gamma', 'beta', 'moving_mean', 'moving_variance', 'offset', 'biases')


def load_checkpoint(prefix, verify_crc=False):
    """`prefix` = path without extension (e.g. results/Log_contraloss/snapshots/snap-54).
    -> {name without the KernelPointNetwork/ root: float32 ndarray}.  Raises FileNotFoundError when the data shard is
    absent (the public checkout lists it in .MISSING_LARGE_BLOBS)."""
    entries = read_index(prefix + '.index')
    shard = prefix + '.data-00000-of-00001'
    if not os.path.exists(shard):
        raise FileNotFoundError('%s is missing: the index alone carries names and shapes, not values' % shard)
    out = {}
    with open(shard, 'rb') as f:
        for name, e in entries.items():
            if not is_model_variable(name):
                continue
            if e.dtype not in _NP_OF_DT:
                raise ValueError('%s: unsupported dtype %d' % (name, e.dtype))
            f.seek(e.offset)
            raw = f.read(e.size)
            if len(raw) != e.size:
                raise ValueError('%s: data shard truncated' % name)
            if verify_crc and masked_crc32c(raw) != e.crc32c:
                raise ValueError('%s: crc32c mismatch' % name)
            arr = np.frombuffer(raw, dtype=_NP_OF_DT[e.dtype]).reshape(e.shape)
            out[name[len(ROOT_SCOPE):]] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def load_weight_dumps(folder):
    """The side dumps written by utils/trainer.py:503-557 (`kernel_points/epochN/*.npy|*.ply`): file stem =
    '_'.join(scope parts), e.g. layer_1_resnetb_0_conv2.npy (weights) / .ply (kernel points).
    -> {variable name: ndarray} for the files present."""
    from .ply import read_ply
    out = {}
    for fn in sorted(os.listdir(folder)):
        stem, ext = os.path.splitext(fn)
        parts = stem.split('_')
        # layer_L_<block words>_<i>[_convK|_shortcut]
        scope_tail = None
        if parts[-1] in ('conv1', 'conv2', 'conv3', 'shortcut'):
            scope_tail, parts = parts[-1], parts[:-1]
        head = '%s_%s' % (parts[0], parts[1])
        block = '_'.join(parts[2:])
        scope = head + '/' + block + ('/' + scope_tail if scope_tail else '')
        if ext == '.npy':
            out[scope + '/weights'] = np.load(os.path.join(folder, fn)).astype(np.float32)
        elif ext == '.ply':
            d = read_ply(os.path.join(folder, fn))
            out[scope + '/kernel_points'] = np.stack([d['x'], d['y'], d['z']], 1).astype(np.float32)
    return out
