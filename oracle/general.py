"""Graph-side and host-side glue of the reference restated in NumPy (oracle; test
infrastructure only).  Follows utils/general.py of lmb-freiburg/hand3d:
  single_obj_scoremap :233-268, find_max_location :199-230, calc_center_bb :271-328,
  crop_image_from_xy :163-196, detect_keypoints :331-344, trafo_coords :347-357,
  EvalUtil :522-611, calc_auc :654-659.
"""
import numpy as np

from . import tf_ops as T

F32 = np.float32
FLT_MAX = np.finfo(np.float32).max

# How tf.reduce_min/max behave on an EMPTY float tensor (only reached for an empty hand
# mask).  'inf'   : +inf / -inf  (newer Eigen) -> NaN centre -> the
#                   reference's fallbacks centre=(160,160), crop_size=100.
#         'fltmax': +FLT_MAX / -FLT_MAX (the Eigen-3.3 reducer identities NumTraits::highest /
#                   lowest that shipped with TF 1.3) -> centre = 0.5*(-FLT_MAX+FLT_MAX) = (0,0)
#                   is finite, size = -inf -> only the crop_size fallback (100) fires.
# The engine implements both (hp3d_set_option "empty_reduce"); default follows SURVEY.md.
EMPTY_REDUCE = 'inf'


def find_max_location(scoremap):
    """utils/general.py:199-230.  scoremap [B,H,W] -> int32 [B,2] (row, col) of the first
    maximal element of the row-major flattened map."""
    scoremap = np.asarray(scoremap)
    B, H, W = scoremap.shape
    assert B < H and B < W, "Scoremap must be [Batch, Width, Height]"  # :210
    idx = T.argmax_first_flat(scoremap.reshape(B, -1))
    return np.stack([idx // W, idx % W], axis=1).astype(np.int32)


def fg_and_detmap(scoremap):
    """utils/general.py:240-242: softmax over the 2 classes, fg = softmax[...,1:].max(-1),
    detmap = tf.round(fg) (half-to-even)."""
    sm = T.softmax_last(scoremap)
    fg = sm[:, :, :, 1:].max(axis=3)
    return fg, T.round_half_even(fg)


def grow_objectmap(det, seed, num_passes=None, filter_size=21, early_exit=False, naive=False):
    """utils/general.py:247-262 for one image.  det [H,W] in {0,1}; seed (row, col).
    objectmap_0 = one-hot(seed); repeat num_passes = max(H,W)//(filter_size//2) times:
    objectmap = round(det * dilation2d(objectmap, ones/441))."""
    H, W = det.shape
    if num_passes is None:
        num_passes = max(H, W) // (filter_size // 2)
    obj = np.zeros((H, W), dtype=F32)
    obj[seed[0], seed[1]] = 1.0  # sparse_to_dense(..., 1.0)  :253
    fv = F32(1.0) / F32(filter_size * filter_size)
    filt = np.full((filter_size, filter_size), fv, dtype=F32)
    passes_done = 0
    for _ in range(num_passes):
        dil = T.dilation2d_naive(obj, filt) if naive else T.dilation2d_flat(obj, filter_size, fv)
        new = T.round_half_even(det * dil)
        passes_done += 1
        if early_exit and np.array_equal(new, obj):
            obj = new
            break
        obj = new
    return obj, passes_done


def single_obj_scoremap(scoremap, early_exit=False):
    """utils/general.py:233-268.  scoremap [B,H,W,2] -> objectmap [B,H,W,1] in {0,1}."""
    scoremap = np.asarray(scoremap, dtype=F32)
    assert scoremap.ndim == 4, "Scoremap must be 4D."
    B, H, W, _ = scoremap.shape
    fg, det = fg_and_detmap(scoremap)
    max_loc = find_max_location(fg)
    out = np.zeros((B, H, W, 1), dtype=F32)
    for i in range(B):
        obj, _ = grow_objectmap(det[i], max_loc[i], early_exit=early_exit)
        out[i, :, :, 0] = obj
    return out


def calc_center_bb(binary_class_mask):
    """utils/general.py:271-328.  mask [B,H,W,1] (or [B,H,W]) -> center [B,2] (row, col),
    bb [B,2,2], crop_size [B,1].  'x' is the ROW index, 'y' the COLUMN index (:285-288)."""
    m = np.asarray(binary_class_mask).astype(np.int32) == 1
    if m.ndim == 4:
        m = m[:, :, :, 0]
    B, H, W = m.shape
    assert B < H and B < W, "binary_class_mask must be [Batch, Width, Height]"  # :282
    centers = np.zeros((B, 2), dtype=F32)
    bbs = np.zeros((B, 2, 2), dtype=F32)
    sizes = np.zeros((B, 1), dtype=F32)
    for i in range(B):
        rows, cols = np.nonzero(m[i])
        if rows.size == 0:
            if EMPTY_REDUCE == 'inf':
                x_min = y_min = F32(np.inf)
                x_max = y_max = F32(-np.inf)
            else:
                x_min = y_min = F32(FLT_MAX)
                x_max = y_max = F32(-FLT_MAX)
        else:
            x_min, x_max = F32(rows.min()), F32(rows.max())
            y_min, y_max = F32(cols.min()), F32(cols.max())
        with np.errstate(invalid='ignore', over='ignore'):
            bbs[i] = np.array([[x_min, x_max], [y_min, y_max]], dtype=F32)  # stack([start,end],1) :303
            center = np.array([F32(0.5) * (x_max + x_min), F32(0.5) * (y_max + y_min)], dtype=F32)
            if not np.all(np.isfinite(center)):
                center = np.array([160.0, 160.0], dtype=F32)  # :311-312
            size = np.maximum(F32(x_max - x_min), F32(y_max - y_min))
            if not np.isfinite(size):
                size = F32(100.0)  # :319-320
        centers[i] = center
        sizes[i, 0] = size
    return centers, bbs, sizes


def crop_boxes(center, crop_size, scale, H, W):
    """The box arithmetic of crop_image_from_xy (utils/general.py:180-191), float32."""
    scale = np.asarray(scale, dtype=F32).reshape(-1)
    loc = np.asarray(center, dtype=F32).reshape(-1, 2)
    cs = F32(crop_size) / scale
    half = np.floor(cs / F32(2.0)).astype(F32)  # float floordiv  (App. B.10)
    y1 = loc[:, 0] - half
    y2 = y1 + cs
    x1 = loc[:, 1] - half
    x2 = x1 + cs
    y1 = y1 / F32(H)
    y2 = y2 / F32(H)
    x1 = x1 / F32(W)
    x2 = x2 / F32(W)
    return np.stack([y1, x1, y2, x2], axis=-1).astype(F32)


def crop_image_from_xy(image, crop_location, crop_size, scale=1.0):
    """utils/general.py:163-196."""
    image = np.asarray(image, dtype=F32)
    assert image.ndim == 4, "Image needs to be of shape [batch, width, height, channel]"
    B, H, W, _ = image.shape
    scale = np.broadcast_to(np.asarray(scale, dtype=F32).reshape(-1), (B,)) if np.ndim(scale) == 0 \
        else np.asarray(scale, dtype=F32).reshape(-1)
    boxes = crop_boxes(crop_location, crop_size, scale, H, W)
    return T.crop_and_resize(image, boxes, int(crop_size), int(crop_size))


def scale_from_crop_size(crop_size_best, crop_size=256):
    """nets/ColorHandPose3DNetwork.py:84-85: size*=1.25; clip(256/size, 0.25, 5.0)."""
    s = np.asarray(crop_size_best, dtype=F32) * F32(1.25)
    with np.errstate(divide='ignore'):
        sc = F32(crop_size) / s
    return np.minimum(np.maximum(sc, F32(0.25)), F32(5.0)).astype(F32)


def preprocess_u8(image_u8, out_h, out_w):
    """The scripts' input pre-processing: `image/255.0 - 0.5` in float32 (data/BinaryDbReader.py:182,
    run.py:59) then tf.image.resize_images(image, (240, 320)) (eval_full.py:50, eval2d.py:53)."""
    x = (np.asarray(image_u8).astype(F32) / F32(255.0) - F32(0.5)).astype(F32)
    return T.resize_bilinear_legacy(x, out_h, out_w)


def detect_keypoints(scoremaps):
    """utils/general.py:331-344.  [H,W,C] -> float64 [C,2] (v=row, u=col)."""
    if len(scoremaps.shape) == 4:
        scoremaps = np.squeeze(scoremaps)
    s = scoremaps.shape
    assert len(s) == 3, "This function was only designed for 3D Scoremaps."
    assert (s[2] < s[1]) and (s[2] < s[0]), "Probably the input is not correct, because [H, W, C] is expected."
    kp = np.zeros((s[2], 2))
    for i in range(s[2]):
        v, u = np.unravel_index(np.argmax(scoremaps[:, :, i]), (s[0], s[1]))
        kp[i, 0] = v
        kp[i, 1] = u
    return kp


def trafo_coords(keypoints_crop_coords, centers, scale, crop_size):
    """utils/general.py:347-357."""
    k = np.copy(keypoints_crop_coords)
    k -= crop_size // 2
    k /= scale
    k += centers
    return k


class EvalUtil:
    """utils/general.py:522-611."""

    def __init__(self, num_kp=21):
        self.data = [list() for _ in range(num_kp)]
        self.num_kp = num_kp

    def feed(self, keypoint_gt, keypoint_vis, keypoint_pred):
        keypoint_gt = np.squeeze(keypoint_gt)
        keypoint_pred = np.squeeze(keypoint_pred)
        keypoint_vis = np.squeeze(keypoint_vis).astype('bool')
        assert len(keypoint_gt.shape) == 2
        assert len(keypoint_pred.shape) == 2
        assert len(keypoint_vis.shape) == 1
        diff = keypoint_gt - keypoint_pred
        dist = np.sqrt(np.sum(np.square(diff), axis=1))
        for i in range(keypoint_gt.shape[0]):
            if keypoint_vis[i]:
                self.data[i].append(dist[i])

    def _get_pck(self, kp_id, threshold):
        if len(self.data[kp_id]) == 0:
            return None
        data = np.array(self.data[kp_id])
        return np.mean((data <= threshold).astype('float'))

    def _get_epe(self, kp_id):
        if len(self.data[kp_id]) == 0:
            return None, None
        data = np.array(self.data[kp_id])
        return np.mean(data), np.median(data)

    def get_measures(self, val_min, val_max, steps):
        thresholds = np.array(np.linspace(val_min, val_max, steps))
        norm_factor = np.trapezoid(np.ones_like(thresholds), thresholds)
        epe_mean_all, epe_median_all, auc_all, pck_curve_all = [], [], [], []
        for part_id in range(self.num_kp):
            mean, median = self._get_epe(part_id)
            if mean is None:
                continue
            epe_mean_all.append(mean)
            epe_median_all.append(median)
            pck_curve = np.array([self._get_pck(part_id, t) for t in thresholds])
            pck_curve_all.append(pck_curve)
            auc_all.append(np.trapezoid(pck_curve, thresholds) / norm_factor)
        return (np.mean(np.array(epe_mean_all)), np.mean(np.array(epe_median_all)),
                np.mean(np.array(auc_all)), np.mean(np.array(pck_curve_all), 0), thresholds)


def calc_auc(x, y):
    """utils/general.py:654-659."""
    return np.trapezoid(y, x) / np.trapezoid(np.ones_like(y), x)
