"""TensorFlow checkpoint (V2 "tensor bundle") -> weight dict, without TensorFlow.

Replaces `load_weights_from_snapshot` (reference utils/general.py:614-651, used by eval2d.py:70,75, eval3d.py:74 with
`discard_list=['Adam', 'global_step', 'beta']` and by the training scripts with rename dicts): the reference reads the
snapshot through `pywrap_tensorflow.NewCheckpointReader`, drops every variable whose name contains a discard
substring, renames by substring replacement and assigns by name.  Here the same selection yields a
{name: ndarray} dict for `ColorHandPose3DNetwork.init_from_dict` / `hp3d_set_weight`.

Format (tensorflow/core/util/tensor_bundle, leveldb table format -- restated from the published layout, there is no
TensorFlow in this environment to cross-check against: verified by round trips through `write_bundle` only):
  <prefix>.index   sorted string table: key "" -> BundleHeaderProto, key <tensor name> -> BundleEntryProto
                   {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: fixed32 masked crc32c}
  <prefix>.data-0000k-of-0000n   raw little-endian tensor bytes at [offset, offset + size)
  table = data blocks + metaindex block + index block + 48-byte footer (two BlockHandles, magic 0xdb4775248b80fb57);
  block = prefix-compressed entries (varint shared, non_shared, value_len) + restart array + count; each block is
  followed by a 1-byte compression type (0 = none, the only one TF writes for bundles) and a masked crc32c.
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           19: np.float16}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- crc32c (Castagnoli), masked as leveldb / TF do --------------------------------------------------------
def _make_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return np.array(t, dtype=np.uint32)


_TABLE = _make_table()


_native = None


def _native_crc():
    """hp3d_crc32c from the engine library when it is built (slicing-by-8, ~1 GB/s); None otherwise."""
    global _native
    if _native is None:
        try:
            from .. import _lib
            lib = _lib.load()
            _native = lambda b: int(lib.hp3d_crc32c(b, len(b)))
        except Exception:
            _native = False
    return _native


def crc32c(data, force_python=False):
    data = bytes(data)
    nat = None if force_python else _native_crc()
    if nat and len(data) > 64:
        return nat(data)
    crc = 0xFFFFFFFF
    tab = _TABLE
    for b in data:
        crc = int(tab[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / minimal protobuf -------------------------------------------------------------------------------
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
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _proto_fields(buf):
    """[(field number, wire type, value)] of one message (varint / 64-bit / length-delimited / 32-bit)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for num, wt, v in _proto_fields(buf):
        if num == 1:
            e['dtype'] = v
        elif num == 2:
            for n2, _, v2 in _proto_fields(v):
                if n2 == 2:                              # TensorShapeProto.Dim
                    size = 0
                    for n3, _, v3 in _proto_fields(v2):
                        if n3 == 1:
                            size = _signed64(v3)
                    e['shape'].append(size)
                elif n2 == 3 and v2:
                    raise ValueError("tensor of unknown rank in checkpoint")
        elif num == 3:
            e['shard_id'] = v
        elif num == 4:
            e['offset'] = v
        elif num == 5:
            e['size'] = v
        elif num == 6:
            e['crc32c'] = struct.unpack('<I', v)[0]
        elif num == 7:
            e['slices'] += 1
    return e


# ---- table reader -----------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        want = struct.unpack('<I', data[offset + size + 1:offset + size + 5])[0]
        if _mask(crc32c(data[offset:offset + size + 1])) != want:
            raise ValueError("checkpoint index: block checksum mismatch at offset %d" % offset)
    if ctype != 0:
        raise NotImplementedError("compressed (type %d) table blocks are not supported" % ctype)
    return contents


def _block_entries(block):
    nrestarts = struct.unpack('<I', block[-4:])[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(index_path, verify=True):
    """{tensor name: entry dict} and the header's shard count of a <prefix>.index file."""
    data = open(index_path, 'rb').read()
    if len(data) < 48 or struct.unpack('<Q', data[-8:])[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow V2 checkpoint index (bad table magic)" % index_path)
    footer = data[-48:]
    _, p = _get_varint(footer, 0)            # metaindex handle (unused)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    entries, num_shards = {}, 1
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b'':
                for num, _, v in _proto_fields(value):
                    if num == 1:
                        num_shards = v
                    elif num == 2 and v != 0:
                        raise NotImplementedError("big-endian checkpoints are not supported")
            else:
                entries[key.decode('utf-8')] = _parse_entry(value)
    return entries, num_shards


def read_bundle(prefix, verify=True, keep=None):
    """Tensors of the checkpoint <prefix>(.index, .data-*) as {name: ndarray}; `keep(name) -> bool` selects which entries
    are read at all (the others are neither decoded nor checksummed, whatever their dtype)."""
    entries, num_shards = read_index(prefix + '.index', verify)
    shards, out = {}, {}
    for name, e in entries.items():
        if keep is not None and not keep(name):
            continue
        if e['slices']:
            raise NotImplementedError("partitioned variable %s (tensor slices) is not supported" % name)
        if e['dtype'] not in _DTYPES:
            raise NotImplementedError("tensor %s has unsupported dtype enum %d" % (name, e['dtype']))
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), dtype=np.uint8, mode='r')
        raw = bytes(shards[sid][e['offset']:e['offset'] + e['size']])
        if verify and e['crc32c'] is not None and _mask(crc32c(raw)) != e['crc32c']:
            raise ValueError("checkpoint data: checksum mismatch for %s" % name)
        dt = np.dtype(_DTYPES[e['dtype']])
        n = int(np.prod(e['shape'])) if e['shape'] else 1
        if n * dt.itemsize != e['size']:
            raise ValueError("tensor %s: %d bytes stored, shape %s needs %d" % (name, e['size'], e['shape'], n * dt.itemsize))
        out[name] = np.frombuffer(raw, dtype=dt.newbyteorder('<')).astype(dt).reshape(e['shape'])
    return out


def load_weights_from_snapshot(checkpoint_path, discard_list=None, rename_dict=None):
    """The selection / renaming of utils/general.py:614-651 on a V2 checkpoint prefix -> {name: ndarray}.
    A variable is dropped if ANY discard string occurs in its name; every rename key that occurs in the name is
    replaced (in dict order, on the progressively renamed name -- exactly the reference's loop)."""
    # only the surviving variables are touched, like the reference (reader.get_tensor runs after the discard loop,
    # :619-646): optimiser slots or non-float bookkeeping entries in the discard list are never decoded
    tensors = read_bundle(checkpoint_path, keep=lambda n: discard_list is None or not any(d in n for d in discard_list))
    out = {}
    for name, value in tensors.items():
        new_name = name
        if rename_dict is not None:
            for k, v in rename_dict.items():
                if k in name:
                    new_name = new_name.replace(k, v)
        out[new_name] = value
    return out


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint without TensorFlow: the prefix named by `model_checkpoint_path` in
    <dir>/checkpoint (a text-format CheckpointState), or None (eval2d.py:68-69 asserts on that)."""
    import re
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.exists(state):
        return None
    m = re.search(r'^\s*model_checkpoint_path:\s*"([^"]*)"', open(state).read(), re.M)
    if not m:
        return None
    path = m.group(1)
    if not os.path.isabs(path):
        path = os.path.join(checkpoint_dir, path)
    return path if os.path.exists(path + '.index') else None


# ---- writer (tests, exporting weight dicts in the reference's snapshot format) ------------------------------
def _build_block(items, restart_interval=16):
    buf, restarts, last = bytearray(), [], b''
    for i, (key, value) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(last), len(key)) and last[shared] == key[shared]:
                shared += 1
        buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack('<I', r)
    buf += struct.pack('<I', len(restarts))
    return bytes(buf)


def _emit_block(out, block):
    off = len(out)
    out += block + b'\x00'
    out += struct.pack('<I', _mask(crc32c(block + b'\x00')))
    return _put_varint(off) + _put_varint(len(block))


def write_bundle(prefix, tensors, entries_per_block=24):
    """Writes {name: ndarray} as a single-shard V2 checkpoint (<prefix>.index, <prefix>.data-00000-of-00001)."""
    names = sorted(tensors, key=lambda s: s.encode('utf-8'))
    data, items = bytearray(), []
    header = b'\x08\x01' + b'\x10\x00' + b'\x1a\x02\x08\x01'        # num_shards=1, little endian, version{producer=1}
    items.append((b'', header))
    for name in names:
        a = np.asarray(tensors[name], order='C')
        dt = a.dtype.newbyteorder('=')
        if np.dtype(dt) not in _DTYPE_IDS:
            raise NotImplementedError("dtype %s cannot be written" % a.dtype)
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        shape = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(s) for s in a.shape))
        entry = (b'\x08' + _put_varint(_DTYPE_IDS[np.dtype(dt)]) + b'\x12' + _put_varint(len(shape)) + shape +
                 b'\x20' + _put_varint(len(data)) + b'\x28' + _put_varint(len(raw)) + b'\x35' + struct.pack('<I', _mask(crc32c(raw))))
        items.append((name.encode('utf-8'), entry))
        data += raw
    table, index_items = bytearray(), []
    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        index_items.append((chunk[-1][0], _emit_block(table, _build_block(chunk))))
    meta_handle = _emit_block(table, _build_block([]))
    index_handle = _emit_block(table, _build_block(index_items, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    table += footer
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(table))
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), 'checkpoint'), 'w') as f:       # CheckpointState
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % ((os.path.basename(prefix),) * 2))
    return prefix
