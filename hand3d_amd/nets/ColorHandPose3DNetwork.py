"""ColorHandPose3DNetwork -- the reference's call surface on the MI355X engine.

Mirrors nets/ColorHandPose3DNetwork.py:28-99,101-219 of lmb-freiburg/hand3d: same method
names, argument order, return-tuple order, NHWC float32 shapes.  Where the reference built
TF-graph tensors to be evaluated later by `sess.run`, these methods evaluate eagerly on NumPy
arrays through libhp3d.so (hand-written HIP, gfx950).  There is no TensorFlow, no torch and no
CPU fallback behind them.

A script written against the reference changes three lines:
    net = ColorHandPose3DNetwork()
    net.init(None)                                   # was: net.init(sess) after tf.Session()
    outs = net.inference(image_v, hand_side_v, True) # was: sess.run([...6 tensors...], feed_dict)
"""
from __future__ import print_function, unicode_literals

import os
import pickle

import numpy as np

from .._lib import Engine


def read_weight_file(file_name):
    """One weight file -> dict[tf variable name -> ndarray].  `.pickle` is the reference's format (a pickled dict, :50-53);
    `.npz` holds the same keys and arrays (BASELINE.json's wording "loads the existing .npz weights"):
    written by `export_npz` / `pickle_to_npz` below, read without unpickling anything."""
    assert os.path.exists(file_name), "File not found."
    if file_name.endswith('.npz'):
        with np.load(file_name, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    with open(file_name, 'rb') as fi:
        return pickle.load(fi, encoding='latin1')


def load_weight_files(engine, weight_files, exclude_var_list=None, verbose=True):
    """The loading loop of ColorHandPose3DNetwork.init (:50-59) / PosePriorNetwork.init (:47-57):
    read dict[str -> ndarray] (pickle or .npz), drop keys containing any exclude substring, assign by name.
    Returns the merged dict of what was assigned (later files win, like repeated assign ops)."""
    if exclude_var_list is None:
        exclude_var_list = list()
    loaded = dict()
    for file_name in weight_files:
        weight_dict = read_weight_file(file_name)
        weight_dict = {k: v for k, v in weight_dict.items() if not any([x in k for x in exclude_var_list])}
        if len(weight_dict) > 0:
            engine.load_weight_dict(weight_dict)
            loaded.update(weight_dict)
            if verbose:
                print('Loaded %d variables from %s' % (len(weight_dict), file_name))
    engine.finalize_weights()
    return loaded


def save_npz(weight_dict, npz_path):
    """dict[tf variable name -> ndarray] -> one uncompressed .npz with the same keys (C order, the arrays' own dtype: the
    reference's pickles hold float32, and nothing here narrows a file that holds something else)."""
    np.savez(npz_path, **{k: np.ascontiguousarray(v) for k, v in weight_dict.items()})


def pickle_to_npz(weight_files, npz_path, exclude_var_list=None):
    """Converts the reference's weight pickles (one or several, merged in order) into one .npz that `init` accepts."""
    merged = dict()
    for file_name in weight_files:
        d = read_weight_file(file_name)
        merged.update({k: v for k, v in d.items() if not any([x in k for x in (exclude_var_list or [])])})
    save_npz(merged, npz_path)
    return sorted(merged)


class ColorHandPose3DNetwork(object):
    """ Network performing 3D pose estimation of a human hand from a single color image. """

    def __init__(self, device=0, engine=None, keep_weights=False):
        """`keep_weights=True` keeps a host reference to every array `init` assigns, for `export_npz` (off by default: the engine
        owns a packed device copy, a second 140 MB host copy for the lifetime of the object serves nothing else;
        `pickle_to_npz` converts files without a network object)."""
        self.crop_size = 256
        self.num_kp = 21
        self.engine = engine if engine is not None else Engine(device)
        self.keep_weights = bool(keep_weights)
        self.weight_dict = dict()

    def init(self, session=None, weight_files=None, exclude_var_list=None):
        """ Initializes weights from pickled python dictionaries (reference :34-59) or from `.npz` files with the same keys.
            `session` is accepted for call compatibility and ignored. """
        if weight_files is None:
            weight_files = ['./weights/handsegnet-rhd.pickle', './weights/posenet3d-rhd-stb-slr-finetuned.pickle']
        loaded = load_weight_files(self.engine, weight_files, exclude_var_list)
        if self.keep_weights:
            self.weight_dict.update(loaded)

    def init_from_dict(self, weight_dict, dtype=0):
        """Convenience for synthetic weights: the merged content of the weight files.
        dtype='f16' selects the half-precision trunks (BASELINE config 5)."""
        self.engine.load_weight_dict(weight_dict)
        self.engine.finalize_weights(dtype)
        if self.keep_weights:
            self.weight_dict.update(weight_dict)

    def export_npz(self, npz_path):
        """ Writes every variable assigned so far into one `.npz` (keys = TF variable names) that `init` reads back.
            Needs `keep_weights=True` at construction (or use `pickle_to_npz` on the files). """
        assert self.keep_weights, "export_npz needs ColorHandPose3DNetwork(keep_weights=True) (or use pickle_to_npz on the weight files)"
        save_npz(self.weight_dict, npz_path)

    @staticmethod
    def _check_eval(evaluation):
        if not bool(evaluation):
            raise NotImplementedError("inference engine: evaluation=False (dropout active) is a training path")

    def inference(self, image, hand_side, evaluation):
        """ Full pipeline: HandSegNet + PoseNet + PosePrior (reference :61-99).
            Returns hand_scoremap [B,H,W,2], image_crop [B,256,256,3], scale_crop [B,1],
            center [B,2] (row, col), keypoints_scoremap [B,256,256,21], keypoint_coord3d [B,21,3]. """
        self._check_eval(evaluation)
        o = self.engine.infer_full(image, hand_side)
        return o['scoremap'], o['crop'], o['scale'], o['center'], o['kpmap'], o['coord3d']

    def inference_from_uint8(self, image_u8, hand_side, evaluation, net_size=(240, 320)):
        """ Not in the reference class: the scripts' pre-processing (`x/255 - 0.5`, run.py:59 /
            data/BinaryDbReader.py:182, then resize to 240x320, eval_full.py:50) fused in front of
            inference() on the device.  Same 6-tuple as inference(). """
        self._check_eval(evaluation)
        o = self.engine.infer_full_u8(image_u8, hand_side, net_size[0], net_size[1])
        return o['scoremap'], o['crop'], o['scale'], o['center'], o['kpmap'], o['coord3d']

    def inference_keypoints(self, image, hand_side, evaluation):
        """ Not in the reference class: inference() followed by the scripts' host post-processing
            (run.py:72-73: detect_keypoints + trafo_coords, utils/general.py:331-357) evaluated on the device, so
            neither the 5.5 MB/image heat-maps nor the crop travel to the host.  Returns
            keypoint_coord3d [B,21,3] float32, keypoint_hw [B,21,2] float64 (row, col in the input image),
            keypoint_hw_crop [B,21,2] float64 (row, col in the 256x256 crop), scale_crop [B,1], center [B,2]. """
        self._check_eval(evaluation)
        o = self.engine.infer_full(image, hand_side, outputs=('coord3d', 'kp_crop', 'kp_hw', 'scale', 'center'))
        return o['coord3d'], o['kp_hw'], o['kp_crop'].astype(np.float64), o['scale'], o['center']

    def inference2d_keypoints(self, image):
        """ inference2d() + detect_keypoints + trafo_coords on the device (eval2d.py:58,93-94): keypoint_hw [B,21,2]
            float64 in the input image, keypoint_hw_crop [B,21,2] float64, scale_crop, center. """
        kpc, kph, scale, center = self.engine.infer_2d_keypoints(image)
        return kph, kpc.astype(np.float64), scale, center

    def inference2d(self, image):
        """ Only 2D part of the pipeline: HandSegNet + PoseNet (reference :101-129).
            Returns keypoints_scoremap, image_crop, scale_crop, center -- note the order. """
        return self.engine.infer_2d(image)

    def inference_detection(self, image, train=False):
        """ HandSegNet (reference :131-168).  Returns a list (len 1) of [B,H,W,2] score maps. """
        if train:
            raise NotImplementedError("inference engine: train=True is not supported")
        return [self.engine.handsegnet(image)]

    def inference_pose2d(self, image_crop, train=False):
        """ PoseNet (reference :170-219).  Returns the list of 3 [B,h/8,w/8,21] score maps. """
        if train:
            raise NotImplementedError("inference engine: train=True is not supported")
        return self.engine.posenet2d(image_crop)

    def _inference_pose3d(self, keypoints_scoremap, hand_side, evaluation, train=False):
        """ PosePrior + Viewpoint on a [B,32,32,21] score map (reference :221-247). """
        self._check_eval(evaluation)
        return self.engine.pose3d(keypoints_scoremap, hand_side)[0]
