"""Host-side helpers with the reference's signatures and semantics (utils/general.py of
lmb-freiburg/hand3d): detect_keypoints :331-344, trafo_coords :347-357, EvalUtil :522-611,
calc_auc :654-659.  Written from the behaviour (vectorised NumPy), not from the reference text:
  * detect_keypoints: per channel, (row, col) of the first maximum of an [H,W,C] score map, float64 [C,2];
  * trafo_coords: crop coordinates -> image coordinates, (kp - crop_size//2) / scale + centre;
  * EvalUtil: per-keypoint Euclidean errors of the visible keypoints; mean / median end-point error
    averaged over the keypoints that received data; PCK over linspace thresholds; AUC = trapz / width.
"""
import numpy as np

_trapz = getattr(np, 'trapezoid', None) or np.trapz          # NumPy >= 2.0 renamed trapz (the reference calls np.trapz)


def detect_keypoints(scoremaps):
    """[H,W,C] (or [1,H,W,C]) score maps -> float64 [C,2] with (v=row, u=col) of each channel's first maximum."""
    sm = np.squeeze(scoremaps) if scoremaps.ndim == 4 else scoremaps
    assert sm.ndim == 3, "This function was only designed for 3D Scoremaps."
    h, w, c = sm.shape
    assert c < w and c < h, "Probably the input is not correct, because [H, W, C] is expected."
    flat = np.argmax(sm.reshape(h * w, c), axis=0)          # first maximum in row-major order
    return np.stack([flat // w, flat % w], axis=1).astype(np.float64)


def trafo_coords(keypoints_crop_coords, centers, scale, crop_size):
    """Maps keypoints found in the crop back into the frame the crop was taken from."""
    return (np.asarray(keypoints_crop_coords, dtype=np.float64) - (crop_size // 2)) / scale + centers


class EvalUtil:
    """Accumulates per-keypoint errors over a dataset and reports EPE / PCK / AUC (reference semantics)."""

    def __init__(self, num_kp=21):
        self.num_kp = num_kp
        self.data = [[] for _ in range(num_kp)]

    def feed(self, keypoint_gt, keypoint_vis, keypoint_pred):
        gt, pred = np.squeeze(keypoint_gt), np.squeeze(keypoint_pred)
        vis = np.squeeze(keypoint_vis).astype(bool)
        assert gt.ndim == 2 and pred.ndim == 2 and vis.ndim == 1
        err = np.linalg.norm(gt - pred, axis=1)
        for k in np.flatnonzero(vis[:gt.shape[0]]):
            self.data[k].append(err[k])

    def _errors(self):
        return [np.asarray(d, dtype=np.float64) for d in self.data if len(d) > 0]

    def _get_pck(self, kp_id, threshold):
        d = self.data[kp_id]
        return None if len(d) == 0 else float(np.mean(np.asarray(d) <= threshold))

    def _get_epe(self, kp_id):
        d = self.data[kp_id]
        return (None, None) if len(d) == 0 else (float(np.mean(d)), float(np.median(d)))

    def get_measures(self, val_min, val_max, steps):
        """(mean EPE, median EPE, AUC, PCK curve, thresholds): each averaged over keypoints that have data."""
        thresholds = np.linspace(val_min, val_max, steps)
        width = _trapz(np.ones_like(thresholds), thresholds)
        errs = self._errors()
        pck = np.array([[np.mean(e <= t) for t in thresholds] for e in errs])       # [keypoints with data, steps]
        auc = np.array([_trapz(row, thresholds) / width for row in pck])
        return (np.mean([e.mean() for e in errs]), np.mean([np.median(e) for e in errs]), np.mean(auc),
                pck.mean(axis=0), thresholds)


def calc_auc(x, y):
    """Normalised area under the curve y(x)."""
    return _trapz(y, x) / _trapz(np.ones_like(y), x)
