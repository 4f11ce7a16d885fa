"""BinaryDbReader / BinaryDbReaderSTB as NumPy iterators.

Mirror data/BinaryDbReader.py:28-459 and data/BinaryDbReaderSTB.py:28-410 of the reference for the
EVALUATION use the inference scripts make of them (eval2d.py:44, eval2d_gt_cropped.py:37, eval3d.py:48,
eval_full.py:46): same constructor flags, same dictionary keys, same float32 arithmetic; `get()` returns
an iterator of batched dicts instead of TF queue tensors.  Training-time augmentations (hue, uv noise,
crop noise, random crops, scoremap dropout) raise NotImplementedError; the canonical-frame / local-frame
training labels (utils/canonical_trafo.py, bone_rel_trafo) are not produced.

Image-space resampling (GT hand crop = crop_image_from_xy, scale_to_size = tf.image.resize_images) runs on
the engine's kernels, so `engine=` is required for hand_crop / scale_to_size.
"""
import os

import numpy as np

from . import binary_format as fmt

F32 = np.float32


def _check_no_aug(**flags):
    on = [k for k, v in flags.items() if v]
    if on:
        raise NotImplementedError("training-time augmentation not available in the inference harness: %s" % on)


def _gt_hand_crop(d, image_u8, image_size, crop_size, engine):
    """data/BinaryDbReader.py:268-345 / data/BinaryDbReaderSTB.py:219-296 (no noise terms)."""
    assert engine is not None, "hand_crop=True needs engine= (crop_image_from_xy runs on the device)"
    uv21, vis21 = d['keypoint_uv21'], d['keypoint_vis21']
    crop_center = uv21[12, ::-1].astype(F32)
    if not np.all(np.isfinite(crop_center)):
        crop_center = np.array([0.0, 0.0], F32)
    hw = np.stack([uv21[vis21, 1], uv21[vis21, 0]], 1).astype(F32)
    with np.errstate(invalid='ignore'):
        if hw.shape[0]:
            min_coord = np.maximum(hw.min(0), F32(0.0))
            max_coord = np.minimum(hw.max(0), np.array(image_size, F32))
            best = F32(2) * np.maximum(max_coord - crop_center, crop_center - min_coord)
            best = np.minimum(np.maximum(best.max(), F32(50.0)), F32(500.0))
        else:
            best = F32(np.nan)
    if not np.isfinite(best):
        best = F32(200.0)
    scale = np.minimum(np.maximum(F32(crop_size) / best, F32(1.0)), F32(10.0)).astype(F32)
    d['crop_scale'] = scale
    img = (image_u8.astype(F32) / F32(255.0) - F32(0.5))[None]
    d['image_crop'] = engine.crop_and_resize(img, crop_center[None], np.array([scale], F32), crop_size)[0]
    half = crop_size // 2
    u = (uv21[:, 0] - crop_center[1]) * scale + half
    v = (uv21[:, 1] - crop_center[0]) * scale + half
    d['keypoint_uv21'] = np.stack([u, v], 1).astype(F32)
    scale_m = np.array([[scale, 0, 0], [0, scale, 0], [0, 0, 1]], F32)
    t1, t2 = crop_center[0] * scale - half, crop_center[1] * scale - half
    trans_m = np.array([[1, 0, -t2], [0, 1, -t1], [0, 0, 1]], F32)
    d['cam_mat'] = (trans_m @ (scale_m @ d['cam_mat'])).astype(F32)


def _xyz_items(d, xyz21):
    """root-relative, index-bone-normalised coordinates (data/BinaryDbReader.py:236-241)."""
    rel = (xyz21 - xyz21[0]).astype(F32)
    s = np.sqrt(np.sum(np.square(rel[12] - rel[11]), dtype=F32)).astype(F32)
    d['keypoint_scale'] = s
    d['keypoint_xyz21_normed'] = (rel / s).astype(F32)


class _ReaderBase(object):
    def _batches(self, parse):
        order = np.arange(self.num_records)
        if self.shuffle:
            np.random.default_rng(self.seed).shuffle(order)
        with open(self.path_to_db, 'rb') as f:
            buf = []
            for i in order:
                f.seek(int(i) * self.record_bytes)
                raw = f.read(self.record_bytes)
                assert len(raw) == self.record_bytes, "Doesnt add up."
                buf.append(parse(raw))
                if len(buf) == self.batch_size:
                    yield {k: np.stack([b[k] for b in buf], 0) for k in buf[0]}
                    buf = []


class BinaryDbReader(_ReaderBase):
    """ Reads data from a binary dataset created by create_binary_db.py (RHD). """

    def __init__(self, mode=None, batch_size=1, shuffle=True, use_wrist_coord=True, sigma=25.0, hand_crop=False,
                 random_crop_to_size=False, scale_to_size=False, hue_aug=False, coord_uv_noise=False,
                 crop_center_noise=False, crop_scale_noise=False, crop_offset_noise=False, scoremap_dropout=False,
                 path_to_db=None, engine=None, seed=0):
        _check_no_aug(random_crop_to_size=random_crop_to_size, hue_aug=hue_aug, coord_uv_noise=coord_uv_noise,
                      crop_center_noise=crop_center_noise, crop_scale_noise=crop_scale_noise,
                      crop_offset_noise=crop_offset_noise, scoremap_dropout=scoremap_dropout)
        if path_to_db is None:
            path_to_db = './data/bin/'
            if mode == 'training':
                path_to_db += 'rhd_training.bin'
            elif mode == 'evaluation':
                path_to_db += 'rhd_evaluation.bin'
            else:
                assert 0, "Unknown dataset mode."
        assert os.path.exists(path_to_db), "Could not find the binary data file!"
        self.path_to_db = path_to_db
        self.batch_size, self.sigma, self.shuffle, self.seed = batch_size, sigma, shuffle, seed
        self.use_wrist_coord, self.hand_crop, self.scale_to_size = use_wrist_coord, hand_crop, scale_to_size
        self.scale_target_size = (240, 320)
        self.image_size, self.crop_size, self.num_kp = fmt.RHD_IMAGE_SIZE, 256, 42
        self.record_bytes = fmt.RHD_RECORD_BYTES
        size = os.path.getsize(path_to_db)
        assert size % self.record_bytes == 0, "Doesnt add up."
        self.num_records = self.num_samples = size // self.record_bytes
        self.engine = engine

    def _parse(self, raw):
        d = dict()
        f32 = np.frombuffer(raw, '<f4', count=42 * 3 + 42 * 2 + 9)
        xyz = f32[:126].reshape(42, 3).astype(F32)
        uv = f32[126:210].reshape(42, 2).astype(np.int32).astype(F32)       # cast to int32 and back (:147-150)
        cam = f32[210:219].reshape(3, 3).astype(F32)
        off = 4 * 219 + 2
        u8 = np.frombuffer(raw, np.uint8)
        H, W = self.image_size
        image_u8 = u8[off:off + H * W * 3].reshape(H, W, 3)
        off += H * W * 3
        parts = u8[off:off + H * W].reshape(H, W).astype(np.int32)
        off += H * W
        vis = u8[off:off + 42].astype(bool)
        assert off + 42 == self.record_bytes, "Doesnt add up."
        if not self.use_wrist_coord:                                          # palm centre instead of wrist (:139-143)
            xyz = np.concatenate([0.5 * (xyz[0:1] + xyz[12:13]), xyz[1:21], 0.5 * (xyz[21:22] + xyz[33:34]), xyz[-20:]], 0)
            uv = np.concatenate([0.5 * (uv[0:1] + uv[12:13]), uv[1:21], 0.5 * (uv[21:22] + uv[33:34]), uv[-20:]], 0)
            vis = np.concatenate([vis[0:1] | vis[12:13], vis[1:21], vis[21:22] | vis[33:34], vis[-20:]], 0)
        d['keypoint_xyz'], d['keypoint_uv'], d['cam_mat'] = xyz.astype(F32), uv.astype(F32), cam
        d['image'] = (image_u8.astype(F32) / F32(255.0) - F32(0.5)).astype(F32)
        d['hand_parts'] = parts
        hand = parts > 1
        d['hand_mask'] = np.stack([~hand, hand], 2).astype(np.int32)
        d['keypoint_vis'] = vis
        # dominant hand from the part mask (:211-233): left parts 2..17, right parts 18..
        left = int(((parts > 1) & (parts < 18)).sum()) > int((parts > 17).sum())
        d['hand_side'] = np.array([1.0, 0.0] if left else [0.0, 1.0], F32)
        xyz21 = (xyz[:21] if left else xyz[-21:]).astype(F32)
        d['keypoint_xyz21'] = xyz21
        _xyz_items(d, xyz21)
        d['keypoint_vis21'] = vis[:21] if left else vis[-21:]
        d['keypoint_uv21'] = (uv[:21] if left else uv[-21:]).astype(F32)
        if self.hand_crop:
            _gt_hand_crop(d, image_u8, self.image_size, self.crop_size, self.engine)
        hw21 = np.stack([d['keypoint_uv21'][:, 1], d['keypoint_uv21'][:, 0]], -1)
        size = (self.crop_size, self.crop_size) if self.hand_crop else self.image_size
        d['scoremap'] = fmt.create_multiple_gaussian_map(hw21, size, self.sigma, valid_vec=d['keypoint_vis21'])
        if self.scale_to_size:                                                # :369-379
            assert self.engine is not None, "scale_to_size=True needs engine= (resize runs on the device)"
            th, tw = self.scale_target_size
            img = self.engine.preprocess_u8(image_u8[None], th, tw)[0]
            sc = (th / float(H), tw / float(W))
            uv21 = np.stack([d['keypoint_uv21'][:, 0] * F32(sc[1]), d['keypoint_uv21'][:, 1] * F32(sc[0])], 1).astype(F32)
            d = {'image': img, 'keypoint_uv21': uv21, 'keypoint_vis21': d['keypoint_vis21']}
        return d

    def get(self):
        """ Provides input data: an iterator of dicts of batched NumPy arrays. """
        return self._batches(self._parse)


class BinaryDbReaderSTB(_ReaderBase):
    """ Reads data from the STB binary dataset (data/BinaryDbReaderSTB.py). """

    # data/BinaryDbReaderSTB.py:396-410
    _KP = [0, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]

    def __init__(self, mode=None, batch_size=1, shuffle=True, use_wrist_coord=True, sigma=25.0, hand_crop=False,
                 random_crop_size=None, hue_aug=False, coord_uv_noise=False, crop_center_noise=False,
                 crop_scale_noise=False, crop_offset_noise=False, scoremap_dropout=False, path_to_db=None,
                 engine=None, seed=0):
        _check_no_aug(random_crop_size=random_crop_size, hue_aug=hue_aug, coord_uv_noise=coord_uv_noise,
                      crop_center_noise=crop_center_noise, crop_scale_noise=crop_scale_noise,
                      crop_offset_noise=crop_offset_noise, scoremap_dropout=scoremap_dropout)
        if path_to_db is None:
            path_to_db = './data/stb/'
            if mode == 'training':
                path_to_db += 'stb_train_shuffled.bin'
            elif mode == 'evaluation':
                path_to_db += 'stb_eval.bin'
            else:
                assert 0, "Unknown dataset mode."
        assert os.path.exists(path_to_db), "Could not find the binary data file!"
        self.path_to_db = path_to_db
        self.batch_size, self.sigma, self.shuffle, self.seed = batch_size, sigma, shuffle, seed
        self.use_wrist_coord, self.hand_crop = use_wrist_coord, hand_crop
        self.image_size, self.crop_size, self.num_kp = fmt.STB_IMAGE_SIZE, 256, 21
        self.record_bytes = fmt.STB_RECORD_BYTES
        size = os.path.getsize(path_to_db)
        assert size % self.record_bytes == 0, "Doesnt add up."
        self.num_records = self.num_samples = size // self.record_bytes
        self.engine = engine

    def _parse(self, raw):
        d = dict()
        f32 = np.frombuffer(raw, '<f4', count=126)
        xyz21 = (f32[:63].reshape(21, 3).astype(F32) / F32(1000.0))[self._KP]       # mm -> m, re-ordered
        uvv = f32[63:126].reshape(21, 3).astype(F32)[self._KP]
        uv21, vis21 = uvv[:, :2].copy(), uvv[:, 2] == 1.0
        if self.use_wrist_coord:                                                     # :131-153
            xyz21 = np.concatenate([(xyz21[16] + F32(2.0) * (xyz21[0] - xyz21[16]))[None], xyz21[1:]], 0)
            vis21 = np.concatenate([[vis21[16] | vis21[0]], vis21[1:]], 0)
            uv21 = np.concatenate([(uv21[16] + F32(2.0) * (uv21[0] - uv21[16]))[None], uv21[1:]], 0)
        H, W = self.image_size
        image_u8 = np.frombuffer(raw, np.uint8)[4 * 126:].reshape(H, W, 3)
        d['keypoint_xyz21'], d['keypoint_vis21'], d['keypoint_uv21'] = xyz21.astype(F32), vis21, uv21.astype(F32)
        d['image'] = (image_u8.astype(F32) / F32(255.0) - F32(0.5)).astype(F32)
        d['cam_mat'] = np.array([[822.79041, 0.0, 318.47345], [0.0, 822.79041, 250.31296], [0.0, 0.0, 1.0]], F32)
        d['hand_side'] = np.array([1.0, 0.0], F32)                                   # only left hands (:183)
        _xyz_items(d, d['keypoint_xyz21'])
        if self.hand_crop:
            _gt_hand_crop(d, image_u8, self.image_size, self.crop_size, self.engine)
        hw21 = np.stack([d['keypoint_uv21'][:, 1], d['keypoint_uv21'][:, 0]], -1)
        size = (self.crop_size, self.crop_size) if self.hand_crop else self.image_size
        d['scoremap'] = fmt.create_multiple_gaussian_map(hw21, size, self.sigma, valid_vec=d['keypoint_vis21'])
        return d

    def get(self):
        return self._batches(self._parse)
