"""Binary record layouts of the reference's datasets (and writers for them).

RHD  (create_binary_db.py:44-88, read back at data/BinaryDbReader.py:103-208):
    42x3 f32 xyz [m] | 42x2 f32 uv [px] | 3x3 f32 K | 2 pad bytes (255,255) |
    320x320x3 u8 image | 320x320 u8 part mask | 42 u8 visibility
STB  (data/stb/write_binary_record.m, read back at data/BinaryDbReaderSTB.py:99-190):
    21x3 f32 xyz [mm] | 21x3 f32 (u, v, vis) | 480x640x3 u8 image
All little-endian, no header, fixed-length records.
"""
import numpy as np

RHD_IMAGE_SIZE = (320, 320)
STB_IMAGE_SIZE = (480, 640)
RHD_RECORD_BYTES = 4 * 42 * 3 + 4 * 42 * 2 + 4 * 9 + 2 + 320 * 320 * 3 + 320 * 320 + 42
STB_RECORD_BYTES = 4 * 21 * 3 + 4 * 21 * 3 + 480 * 640 * 3


def pack_rhd_record(image, mask, kp_coord_xyz, kp_coord_uv, kp_visible, K_mat):
    """Bytes of one RHD record (create_binary_db.py:44-88)."""
    assert image.shape == (320, 320, 3) and mask.shape == (320, 320)
    parts = [np.asarray(kp_coord_xyz, '<f4').reshape(42, 3).tobytes(),
             np.asarray(kp_coord_uv, '<f4').reshape(42, 2).tobytes(),
             np.asarray(K_mat, '<f4').reshape(3, 3).tobytes(),
             bytes([255, 255]),
             np.asarray(image, np.uint8).tobytes(), np.asarray(mask, np.uint8).tobytes(),
             np.asarray(kp_visible).astype(np.uint8).reshape(42).tobytes()]
    rec = b''.join(parts)
    assert len(rec) == RHD_RECORD_BYTES
    return rec


def pack_stb_record(image, kp_xyz_mm, kp_uv_vis):
    assert image.shape == (480, 640, 3)
    rec = np.asarray(kp_xyz_mm, '<f4').reshape(21, 3).tobytes() + np.asarray(kp_uv_vis, '<f4').reshape(21, 3).tobytes() \
        + np.asarray(image, np.uint8).tobytes()
    assert len(rec) == STB_RECORD_BYTES
    return rec


def create_multiple_gaussian_map(coords_hw, output_size, sigma, valid_vec=None):
    """data/BinaryDbReader.py:412-459: exp(-d^2/sigma^2) per keypoint; a keypoint contributes only when
    valid and strictly inside (0, size-1) after truncation to int.  float32 [H,W,K]."""
    c = np.asarray(coords_hw, np.float32).astype(np.int32)
    K = c.shape[0]
    val = np.ones(K, bool) if valid_vec is None else (np.asarray(valid_vec, np.float32).reshape(-1) > 0.5)
    inside = (c[:, 0] < output_size[0] - 1) & (c[:, 0] > 0) & (c[:, 1] < output_size[1] - 1) & (c[:, 1] > 0)
    cond = (val & inside).astype(np.float32)
    cf = c.astype(np.float32)
    X = np.arange(output_size[0], dtype=np.float32)[:, None, None] - cf[None, None, :, 0]
    Y = np.arange(output_size[1], dtype=np.float32)[None, :, None] - cf[None, None, :, 1]
    dist = np.square(X) + np.square(Y)
    return (np.exp(-dist / np.square(np.float32(sigma))) * cond).astype(np.float32)
