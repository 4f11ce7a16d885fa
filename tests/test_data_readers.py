"""dataset readers as NumPy iterators, checked on synthetic records written in the
reference's binary layout (create_binary_db.py:44-88, data/stb/write_binary_record.m)."""
import numpy as np
import pytest

from hand3d_amd.data import BinaryDbReader, BinaryDbReaderSTB
from hand3d_amd.data import binary_format as fmt
from oracle import general as G


def _rhd_sample(rng, left=True):
    img = rng.integers(0, 256, (320, 320, 3), dtype=np.uint8)
    mask = np.zeros((320, 320), np.uint8)
    mask[100:180, 90:200] = 5 if left else 20                 # parts 2..17 = left hand, 18.. = right
    mask[10:20, 10:30] = 20 if left else 5                    # a few pixels of the other hand
    xyz = rng.normal(0, 0.05, (42, 3)).astype(np.float32)
    uv = rng.uniform(60, 250, (42, 2)).astype(np.float32)
    vis = rng.uniform(size=42) > 0.2
    K = np.array([[283.1, 0, 160.0], [0, 283.1, 160.0], [0, 0, 1]], np.float32)
    return img, mask, xyz, uv, vis, K


def test_rhd_reader_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    samples = [_rhd_sample(rng, left=(i % 2 == 0)) for i in range(3)]
    path = tmp_path / 'rhd_evaluation.bin'
    with open(path, 'wb') as f:
        for s in samples:
            f.write(fmt.pack_rhd_record(*s))
    assert fmt.RHD_RECORD_BYTES == 2 + 4 * 126 + 4 * 84 + 36 + 307200 + 102400 + 42     # BinaryDbReader.py:103-124
    r = BinaryDbReader(mode='evaluation', shuffle=False, path_to_db=str(path))
    assert r.num_samples == 3
    for (img, mask, xyz, uv, vis, K), d in zip(samples, r.get()):
        assert d['image'].shape == (1, 320, 320, 3)
        assert np.array_equal(d['image'][0], img.astype(np.float32) / np.float32(255) - np.float32(0.5))
        left = int(((mask > 1) & (mask < 18)).sum()) > int((mask > 17).sum())
        assert d['hand_side'][0].tolist() == ([1.0, 0.0] if left else [0.0, 1.0])
        sel = slice(0, 21) if left else slice(21, 42)
        assert np.array_equal(d['keypoint_xyz21'][0], xyz[sel])
        assert np.array_equal(d['keypoint_uv21'][0], np.trunc(uv[sel]))                 # int32 cast (:147)
        assert np.array_equal(d['keypoint_vis21'][0], vis[sel])
        rel = xyz[sel] - xyz[sel][0]
        s = np.sqrt(np.sum((rel[12] - rel[11]) ** 2))
        assert abs(d['keypoint_scale'][0] - s) < 1e-7 and np.allclose(d['keypoint_xyz21_normed'][0], rel / s, atol=1e-5)
        assert np.array_equal(d['cam_mat'][0], K) and d['scoremap'].shape == (1, 320, 320, 21)
    # palm-centre keypoint 0 (use_wrist_coord=False, eval2d_gt_cropped.py:37)
    d = next(BinaryDbReader(mode='evaluation', shuffle=False, use_wrist_coord=False, path_to_db=str(path)).get())
    img, mask, xyz, uv, vis, K = samples[0]
    assert np.allclose(d['keypoint_xyz21'][0, 0], 0.5 * (xyz[0] + xyz[12]))
    with pytest.raises(NotImplementedError):
        BinaryDbReader(mode='evaluation', hue_aug=True, path_to_db=str(path))
    with pytest.raises(AssertionError, match="Could not find"):
        BinaryDbReader(mode='evaluation', path_to_db=str(tmp_path / 'nope.bin'))


def test_rhd_reader_hand_crop_and_scale_on_engine(tmp_path, emu_engine):
    rng = np.random.default_rng(1)
    s = _rhd_sample(rng)
    path = tmp_path / 'rhd_evaluation.bin'
    with open(path, 'wb') as f:
        f.write(fmt.pack_rhd_record(*s))
    d = next(BinaryDbReader(mode='evaluation', shuffle=False, hand_crop=True, use_wrist_coord=False,
                            path_to_db=str(path), engine=emu_engine).get())
    assert d['image_crop'].shape == (1, 256, 256, 3) and d['scoremap'].shape == (1, 256, 256, 21)
    img = (s[0].astype(np.float32) / np.float32(255) - np.float32(0.5))[None]
    base = next(BinaryDbReader(mode='evaluation', shuffle=False, use_wrist_coord=False, path_to_db=str(path)).get())
    center = base['keypoint_uv21'][0, 12, ::-1][None]
    ref = G.crop_image_from_xy(img, center, 256, np.array([d['crop_scale'][0]], np.float32))
    assert np.array_equal(d['image_crop'], ref)
    assert 1.0 <= d['crop_scale'][0] <= 10.0
    back = (d['keypoint_uv21'][0] - 128) / d['crop_scale'][0] + center[0, ::-1]           # uv mapping is invertible
    assert np.allclose(back, base['keypoint_uv21'][0], atol=1e-3)
    d2 = next(BinaryDbReader(mode='evaluation', shuffle=False, scale_to_size=True, use_wrist_coord=False, path_to_db=str(path),
                             engine=emu_engine).get())
    assert sorted(d2) == ['image', 'keypoint_uv21', 'keypoint_vis21'] and d2['image'].shape == (1, 240, 320, 3)
    assert np.array_equal(d2['image'], G.preprocess_u8(s[0][None], 240, 320))
    assert np.allclose(d2['keypoint_uv21'][0, :, 1], base['keypoint_uv21'][0, :, 1] * 0.75)


def test_stb_reader_and_gaussian_maps(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    xyz_mm = rng.normal(0, 50, (21, 3)).astype(np.float32)
    uvv = np.concatenate([rng.uniform(50, 400, (21, 2)), np.ones((21, 1))], 1).astype(np.float32)
    path = tmp_path / 'stb_eval.bin'
    with open(path, 'wb') as f:
        f.write(fmt.pack_stb_record(img, xyz_mm, uvv))
        f.write(fmt.pack_stb_record(img, xyz_mm, uvv))
    r = BinaryDbReaderSTB(mode='evaluation', shuffle=False, use_wrist_coord=False, batch_size=2, path_to_db=str(path))
    d = next(r.get())
    kp = BinaryDbReaderSTB._KP
    assert d['image'].shape == (2, 480, 640, 3) and d['hand_side'].tolist() == [[1.0, 0.0]] * 2
    assert np.allclose(d['keypoint_xyz21'][0], (xyz_mm / 1000.0)[kp]) and np.array_equal(d['keypoint_uv21'][0], uvv[kp, :2])
    dw = next(BinaryDbReaderSTB(mode='evaluation', shuffle=False, use_wrist_coord=True, path_to_db=str(path)).get())
    x = (xyz_mm / 1000.0)[kp]
    assert np.allclose(dw['keypoint_xyz21'][0, 0], x[16] + 2.0 * (x[0] - x[16]), atol=1e-6)
    m = fmt.create_multiple_gaussian_map(np.array([[10.0, 20.0], [0.0, 5.0], [30.7, 40.2]]), (64, 64), 25.0,
                                         valid_vec=[1, 1, 0])
    assert m.shape == (64, 64, 3) and m[10, 20, 0] == 1.0 and m[:, :, 1].max() == 0 and m[:, :, 2].max() == 0
    assert abs(m[10, 45, 0] - np.exp(-1.0)) < 1e-6
