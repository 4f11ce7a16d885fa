"""TF V2 checkpoint reader + load_weights_from_snapshot selection semantics
(reference utils/general.py:614-651).  No TensorFlow here: the format code is exercised by round trips, corruption
checks and the published crc32c test vectors."""
import struct

import numpy as np
import pytest

from hand3d_amd.utils import tf_checkpoint as C


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC32C (Castagnoli)
    assert C.crc32c(b'') == 0
    assert C.crc32c(bytes(32)) == 0x8A9136AA
    assert C.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert C.crc32c(bytes(range(32))) == 0x46DD794E
    assert C.crc32c(b'123456789') == 0xE3069283
    # the engine library's slicing-by-8 version agrees with the table loop (odd lengths, unaligned tails)
    rng = np.random.default_rng(1)
    for n in (65, 100, 1023, 4099):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert C.crc32c(b) == C.crc32c(b, force_python=True)


def test_bundle_round_trip_and_selection(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {'global_step': np.array(1234, np.int64), 'beta1_power': np.array(0.9, np.float32)}
    for i in range(40):                                   # > one table block, shared key prefixes
        base = 'PoseNet2D/conv%d_%d' % (i // 8 + 1, i % 8 + 1)
        tensors[base + '/weights'] = rng.standard_normal((3, 3, 4, 5)).astype(np.float32)
        tensors[base + '/biases'] = rng.standard_normal(5).astype(np.float32)
        tensors[base + '/weights/Adam'] = np.zeros((3, 3, 4, 5), np.float32)
        tensors[base + '/weights/Adam_1'] = np.zeros((3, 3, 4, 5), np.float32)
    tensors['CPM/PoseNet/conv1_1/weights'] = rng.standard_normal((3, 3, 3, 8)).astype(np.float32)
    tensors['half'] = rng.standard_normal((2, 3)).astype(np.float16)
    tensors['empty'] = np.zeros((0, 4), np.float32)
    prefix = C.write_bundle(str(tmp_path / 'model-30000'), tensors)
    back = C.read_bundle(prefix)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    # eval2d.py:70 / eval3d.py:74 selection
    sel = C.load_weights_from_snapshot(prefix, discard_list=['Adam', 'global_step', 'beta'])
    assert 'global_step' not in sel and 'beta1_power' not in sel and not any('Adam' in k for k in sel)
    assert len(sel) == 80 + 3
    # training_posenet.py:74-76 style rename
    ren = C.load_weights_from_snapshot(prefix, discard_list=['Adam', 'global_step', 'beta', 'PoseNet2D'],
                                       rename_dict={'CPM/PoseNet': 'PoseNet2D'})
    assert 'PoseNet2D/conv1_1/weights' in ren and np.array_equal(ren['PoseNet2D/conv1_1/weights'], tensors['CPM/PoseNet/conv1_1/weights'])


def test_corruption_is_detected(tmp_path):
    prefix = C.write_bundle(str(tmp_path / 'm'), {'a/weights': np.arange(12, dtype=np.float32).reshape(3, 4)})
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    raw[5] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        C.read_bundle(prefix)
    assert C.read_bundle(prefix, verify=False)['a/weights'].shape == (3, 4)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[3] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError):
        C.read_bundle(prefix)
    open(prefix + '.index', 'wb').write(b'not a table')
    with pytest.raises(ValueError):
        C.read_bundle(prefix)


def test_discarded_entries_are_never_decoded(tmp_path):
    """load_weights_from_snapshot touches only the tensors it keeps (utils/general.py:619-646): a damaged optimiser slot
    in the discard list must not stop the load."""
    t = {'PoseNet2D/conv1_1/weights': np.arange(24, dtype=np.float32).reshape(2, 3, 4),
         'PoseNet2D/conv1_1/weights/Adam': np.ones((2, 3, 4), np.float32)}
    prefix = C.write_bundle(str(tmp_path / 'm'), t)
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    pos = bytes(raw).find(np.ones(24, np.float32).tobytes())
    assert pos >= 0
    raw[pos + 7] ^= 0x10                                              # corrupt the Adam slot only
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        C.read_bundle(prefix)
    w = C.load_weights_from_snapshot(prefix, discard_list=['Adam', 'global_step', 'beta'])
    assert list(w) == ['PoseNet2D/conv1_1/weights'] and np.array_equal(w['PoseNet2D/conv1_1/weights'], t['PoseNet2D/conv1_1/weights'])


def test_snapshot_feeds_the_engine_loader(tmp_path, emu_engine, synth_weights):
    """A retrained snapshot (with optimizer slots) goes through the converter into init_from_dict."""
    from hand3d_amd import ColorHandPose3DNetwork
    snap = dict(synth_weights)
    snap['global_step'] = np.array(7, np.int64)
    for k in list(synth_weights)[:5]:
        snap[k + '/Adam'] = np.zeros_like(synth_weights[k])
    prefix = C.write_bundle(str(tmp_path / 'snapshots' / 'model-1'), snap)
    assert C.latest_checkpoint(str(tmp_path / 'snapshots')) == prefix and C.latest_checkpoint(str(tmp_path)) is None
    w = C.load_weights_from_snapshot(prefix, discard_list=['Adam', 'global_step', 'beta'])
    assert set(w) == set(synth_weights)
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(w)
    assert emu_engine.nets_mask() & 15 == 15


def _crc32c_bitwise(data):
    """An independent CRC-32C (reflected polynomial 0x82F63B78, bit by bit): not the module's table / native code."""
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def _masked(data):
    c = _crc32c_bitwise(data)
    return struct.pack('<I', ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF)


def test_reader_on_a_bundle_assembled_by_hand(tmp_path):
    """A two-tensor V2 checkpoint written out BYTE BY BYTE from the published layout (leveldb table format + BundleHeaderProto /
    BundleEntryProto of tensorflow/core/protobuf/tensor_bundle.proto), not by write_bundle: the reader is not only checked
    against its own writer.  Zero-valued proto fields (shard_id 0, offset 0) are omitted as proto3 writers do; the second key
    is prefix-compressed against the first ("a/" shared); checksums come from an independent bit-by-bit CRC-32C."""
    assert _crc32c_bitwise(b'123456789') == 0xE3069283
    biases = struct.pack('<2f', 1.0, -2.0)                 # "a/biases"  float32 [2]    at offset 0
    weights = struct.pack('<2f', 0.5, 3.0)                 # "a/weights" float32 [1, 2] at offset 8
    data = biases + weights
    header = bytes([0x08, 0x01,                            # BundleHeaderProto.num_shards = 1
                    0x10, 0x00,                            # .endianness = LITTLE
                    0x1A, 0x02, 0x08, 0x01])               # .version { producer: 1 }
    e_b = bytes([0x08, 0x01,                               # BundleEntryProto.dtype = DT_FLOAT
                 0x12, 0x04, 0x12, 0x02, 0x08, 0x02,       # .shape { dim { size: 2 } }
                 0x28, 0x08,                               # .size = 8            (shard_id = 0 and offset = 0: omitted)
                 0x35]) + _masked(biases)                  # .crc32c (fixed32, masked)
    e_w = bytes([0x08, 0x01,
                 0x12, 0x08, 0x12, 0x02, 0x08, 0x01, 0x12, 0x02, 0x08, 0x02,      # .shape { dim { size: 1 } dim { size: 2 } }
                 0x20, 0x08,                               # .offset = 8
                 0x28, 0x08,
                 0x35]) + _masked(weights)
    # data block: entries (shared, non_shared, value_len, key suffix, value), restart array [0], restart count 1
    block = (bytes([0, 0, len(header)]) + header +
             bytes([0, 8, len(e_b)]) + b'a/biases' + e_b +
             bytes([2, 7, len(e_w)]) + b'weights' + e_w +
             struct.pack('<II', 0, 1))
    table = block + b'\x00' + _masked(block + b'\x00')               # compression type 0 + masked crc of (block + type)
    meta = struct.pack('<II', 0, 1)                                   # empty metaindex block: one restart at 0
    meta_off = len(table)
    table += meta + b'\x00' + _masked(meta + b'\x00')
    handle = bytes([0x00, len(block)])                                # BlockHandle of the data block: offset 0, size (both < 128)
    assert len(block) < 128
    index = bytes([0, len(b'a/weights'), len(handle)]) + b'a/weights' + handle + struct.pack('<II', 0, 1)
    index_off = len(table)
    table += index + b'\x00' + _masked(index + b'\x00')
    footer = bytes([meta_off, len(meta), index_off, len(index)])      # two BlockHandles as varints (all < 128 here)
    assert max(footer) < 128
    table += footer + b'\x00' * (40 - len(footer)) + bytes.fromhex('57fb808b247547db')      # magic 0xdb4775248b80fb57, little endian
    prefix = str(tmp_path / 'model-1')
    open(prefix + '.index', 'wb').write(table)
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    got = C.read_bundle(prefix)
    assert sorted(got) == ['a/biases', 'a/weights']
    assert got['a/biases'].dtype == np.float32 and got['a/biases'].tolist() == [1.0, -2.0]
    assert got['a/weights'].shape == (1, 2) and got['a/weights'].tolist() == [[0.5, 3.0]]
    # and the writer produces a table the same reader accepts with identical content (two independent byte streams, one meaning)
    back = C.read_bundle(C.write_bundle(str(tmp_path / 'w' / 'model-1'), got))
    assert all(np.array_equal(back[k], got[k]) for k in got)
    # a flipped payload bit is caught by the hand-computed checksum
    bad = bytearray(data); bad[1] ^= 0x10
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(bad))
    with pytest.raises(ValueError):
        C.read_bundle(prefix)
