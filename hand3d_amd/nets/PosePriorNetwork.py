"""PosePriorNetwork -- nets/PosePriorNetwork.py:30-95 of the reference on the MI355X engine.

All five variants run on the engine: 'direct', 'bottleneck', 'proposed', and 'local' /
'local_w_xyz_loss' (PosePrior net + bone_rel_trafo_inv, utils/relative_trafo.py:243-295).
"""
from __future__ import print_function, unicode_literals

from .._lib import Engine
from .ColorHandPose3DNetwork import load_weight_files, save_npz


class PosePriorNetwork(object):
    """ Network containing different variants for lifting 2D predictions into 3D. """

    def __init__(self, variant, device=0, engine=None, keep_weights=False):
        self.num_kp = 21
        self.variant = variant
        self.engine = engine if engine is not None else Engine(device)
        self.keep_weights = bool(keep_weights)       # host references for export_npz only (see ColorHandPose3DNetwork)
        self.weight_dict = dict()

    def init(self, session=None, weight_files=None, exclude_var_list=None):
        """ reference :36-57 -- weight_files is required there (no default); `.pickle` or `.npz`. """
        assert weight_files is not None, "weight_files is required"
        loaded = load_weight_files(self.engine, weight_files, exclude_var_list)
        if self.keep_weights:
            self.weight_dict.update(loaded)

    def init_from_dict(self, weight_dict):
        self.engine.load_weight_dict(weight_dict)
        self.engine.finalize_weights()
        if self.keep_weights:
            self.weight_dict.update(weight_dict)

    def export_npz(self, npz_path):
        assert self.keep_weights, "export_npz needs PosePriorNetwork(..., keep_weights=True) (or use pickle_to_npz on the weight files)"
        save_npz(self.weight_dict, npz_path)

    def inference(self, scoremap, hand_side, evaluation):
        """ Infere 3D coordinates from 2D scoremaps (reference :59-95).
            Returns (coord_xyz_rel_normed, coord3d, R); R is None for direct/bottleneck. """
        if not bool(evaluation):
            raise NotImplementedError("inference engine: evaluation=False (dropout active) is a training path")
        assert self.variant in ('direct', 'bottleneck', 'proposed', 'local', 'local_w_xyz_loss'), "Unknown variant."
        return self.engine.poseprior(self.variant, scoremap, hand_side)
