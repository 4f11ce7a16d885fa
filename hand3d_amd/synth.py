"""Deterministic synthetic weights and inputs (no released weights/images on the box).

The reference's weight files (handsegnet-rhd.pickle, posenet3d-rhd-stb-slr-finetuned.pickle,
nets/ColorHandPose3DNetwork.py:48) come from a separate download; this module writes pickles in
the same format -- dict[str -> float32 ndarray] keyed by TF variable name --
with fan-in-scaled normal weights so activations stay O(1) through the 33 layers.
"""
import pickle

import numpy as np

from . import arch


def make_weights(seed=42, bottleneck=False, seg_bias=-0.62):
    """All variables of HandSegNet + PoseNet2D + PosePrior + ViewpointNet.
    `seg_bias` shifts the fg logit of HandSegNet/conv6_2 so random-weight masks are blobs
    rather than half the image."""
    rng = np.random.default_rng(seed)
    w = {}
    for l in arch.all_layers(bottleneck=bottleneck):
        base = '%s/%s' % (l.scope, l.name)
        if isinstance(l, arch.Conv):
            fan_in = l.k * l.k * l.cin
            shape = (l.k, l.k, l.cin, l.cout)
        else:
            fan_in = l.cin
            shape = (l.cin, l.cout)
        gain = np.sqrt(2.0 / (1.0 + 0.01 ** 2)) if l.relu else 1.0
        w[base + '/weights'] = (rng.standard_normal(shape) * (gain / np.sqrt(fan_in))).astype(np.float32)
        w[base + '/biases'] = (rng.standard_normal((l.cout,)) * 1e-2).astype(np.float32)
    w['HandSegNet/conv6_2/biases'] = np.array([0.0, seg_bias], dtype=np.float32)
    return w


def split_weight_files(weights):
    """The reference ships HandSegNet and PoseNet2D+PosePrior+ViewpointNet separately."""
    seg = {k: v for k, v in weights.items() if k.startswith('HandSegNet/')}
    pose = {k: v for k, v in weights.items() if not k.startswith('HandSegNet/')}
    return seg, pose


def write_weight_files(dirname, weights, seg_name='handsegnet-rhd.pickle',
                       pose_name='posenet3d-rhd-stb-slr-finetuned.pickle'):
    import os
    os.makedirs(dirname, exist_ok=True)
    seg, pose = split_weight_files(weights)
    paths = [os.path.join(dirname, seg_name), os.path.join(dirname, pose_name)]
    for p, d in zip(paths, (seg, pose)):
        with open(p, 'wb') as f:
            pickle.dump(d, f, protocol=2)
    return paths


def make_image(seed, H=240, W=320, blob=True):
    """Synthetic `image/255 - 0.5` input: low-passed noise + one bright
    hand-sized blob.  float32 [H,W,3] in [-0.5,0.5]."""
    rng = np.random.default_rng(seed)
    img = rng.uniform(-0.5, 0.5, size=(H // 8 + 2, W // 8 + 2, 3))
    img = np.kron(img, np.ones((8, 8, 1)))[4:4 + H, 4:4 + W]
    img = 0.6 * img + 0.15 * rng.uniform(-0.5, 0.5, size=(H, W, 3))
    if blob:
        cy, cx = rng.uniform(0.3, 0.7) * H, rng.uniform(0.3, 0.7) * W
        ry, rx = rng.uniform(0.12, 0.22) * H, rng.uniform(0.10, 0.18) * W
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.exp(-(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2))
        img = img * (1 - m[..., None]) + m[..., None] * np.array([0.45, 0.3, 0.2])
    return np.clip(img, -0.5, 0.5).astype(np.float32)


def make_batch(seed0, B, H=240, W=320):
    return np.stack([make_image(seed0 + i, H, W) for i in range(B)], 0)


def hand_sides(B):
    """Alternating left/right one-hot [B,2] (col 0 = left)."""
    hs = np.zeros((B, 2), dtype=np.float32)
    hs[np.arange(B), np.arange(B) % 2] = 1.0
    return hs


MASK_CASES = {      # engineered segmentation score maps for the mask / bounding-box stage (120 x 160)
    'one_blob': [(30, 70, 40, 90, 1.0)],
    # the 21x21 dilation of single_obj_scoremap bridges 10 background pixels but not 11
    'two_blobs_gap10': [(30, 60, 20, 50, 1.2), (30, 60, 60, 90, 1.0)],
    'two_blobs_gap11': [(30, 60, 20, 50, 1.2), (30, 60, 61, 90, 1.0)],
    'empty': [], 'full': [(0, 120, 0, 160, 1.0)], 'border': [(0, 12, 150, 160, 1.0)],
}


def blob_scoremap(case, H=120, W=160, strength=4.0, seed=0):
    """[1,H,W,2] hand score map: background logit 1, rectangles (y0,y1,x0,x1,s) with fg logit s*strength, tiny noise."""
    rng = np.random.default_rng(seed)
    sm = np.zeros((1, H, W, 2), np.float32)
    sm[..., 0] = 1.0
    for (y0, y1, x0, x1, s) in MASK_CASES[case]:
        sm[0, y0:y1, x0:x1, 1] = s * strength
    sm += (rng.standard_normal(sm.shape) * 0.01).astype(np.float32)
    return sm


def lifting_scoremaps(seed, B, size=256):
    """Seeded [B,size,size,21] score maps for the PosePriorNetwork entry point."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((B, size, size, 21)) * 0.3).astype(np.float32)
