"""Host-side helpers of the reference with unchanged signatures (utils/general.py of
lmb-freiburg/hand3d): detect_keypoints :331-344, trafo_coords :347-357, EvalUtil :522-611,
calc_auc :654-659.  Pure NumPy post-processing, exactly as the reference runs them on the host.
"""
import numpy as np


def detect_keypoints(scoremaps):
    """ Performs detection per scoremap for the hands keypoints. """
    if len(scoremaps.shape) == 4:
        scoremaps = np.squeeze(scoremaps)
    s = scoremaps.shape
    assert len(s) == 3, "This function was only designed for 3D Scoremaps."
    assert (s[2] < s[1]) and (s[2] < s[0]), "Probably the input is not correct, because [H, W, C] is expected."
    keypoint_coords = np.zeros((s[2], 2))
    for i in range(s[2]):
        v, u = np.unravel_index(np.argmax(scoremaps[:, :, i]), (s[0], s[1]))
        keypoint_coords[i, 0] = v
        keypoint_coords[i, 1] = u
    return keypoint_coords


def trafo_coords(keypoints_crop_coords, centers, scale, crop_size):
    """ Transforms coords into global image coordinates. """
    keypoints_coords = np.copy(keypoints_crop_coords)
    keypoints_coords -= crop_size // 2
    keypoints_coords /= scale
    keypoints_coords += centers
    return keypoints_coords


class EvalUtil:
    """ Util class for evaluation networks. """

    def __init__(self, num_kp=21):
        self.data = list()
        self.num_kp = num_kp
        for _ in range(num_kp):
            self.data.append(list())

    def feed(self, keypoint_gt, keypoint_vis, keypoint_pred):
        """ Stores the euclidean distance between gt and pred, when it is visible. """
        keypoint_gt = np.squeeze(keypoint_gt)
        keypoint_pred = np.squeeze(keypoint_pred)
        keypoint_vis = np.squeeze(keypoint_vis).astype('bool')
        assert len(keypoint_gt.shape) == 2
        assert len(keypoint_pred.shape) == 2
        assert len(keypoint_vis.shape) == 1
        diff = keypoint_gt - keypoint_pred
        euclidean_dist = np.sqrt(np.sum(np.square(diff), axis=1))
        num_kp = keypoint_gt.shape[0]
        for i in range(num_kp):
            if keypoint_vis[i]:
                self.data[i].append(euclidean_dist[i])

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
        """ Outputs the average mean and median error as well as the pck score. """
        thresholds = np.array(np.linspace(val_min, val_max, steps))
        norm_factor = np.trapezoid(np.ones_like(thresholds), thresholds)
        epe_mean_all, epe_median_all, auc_all, pck_curve_all = list(), list(), list(), list()
        for part_id in range(self.num_kp):
            mean, median = self._get_epe(part_id)
            if mean is None:
                continue
            epe_mean_all.append(mean)
            epe_median_all.append(median)
            pck_curve = np.array([self._get_pck(part_id, t) for t in thresholds])
            pck_curve_all.append(pck_curve)
            auc_all.append(np.trapezoid(pck_curve, thresholds) / norm_factor)
        epe_mean_all = np.mean(np.array(epe_mean_all))
        epe_median_all = np.mean(np.array(epe_median_all))
        auc_all = np.mean(np.array(auc_all))
        pck_curve_all = np.mean(np.array(pck_curve_all), 0)
        return epe_mean_all, epe_median_all, auc_all, pck_curve_all, thresholds


def calc_auc(x, y):
    """ Given x and y values it calculates the approx. integral and normalizes it: area under curve"""
    return np.trapezoid(y, x) / np.trapezoid(np.ones_like(y), x)
