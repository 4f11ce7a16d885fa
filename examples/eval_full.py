#!/usr/bin/env python
"""eval_full.py of the reference (eval_full.py:44-101): full pipeline on STB-e (or RHD-e), mean / median EPE
and AUC via EvalUtil.  --synthetic writes a tiny STB-format file with random records (plumbing check)."""
import os
import tempfile

import numpy as np

from common import parser, print_result, synthetic_stb_db, synthetic_weight_files

if __name__ == '__main__':
    ap = parser(__doc__)
    ap.add_argument('--db', default=None, help='path to stb_eval.bin (default ./data/stb/stb_eval.bin)')
    a = ap.parse_args()
    from hand3d_amd.data import BinaryDbReaderSTB, binary_format as fmt
    from hand3d_amd.nets.ColorHandPose3DNetwork import ColorHandPose3DNetwork
    from hand3d_amd.utils.general import EvalUtil

    net = ColorHandPose3DNetwork(device=a.device)
    if a.synthetic:
        tmp = tempfile.mkdtemp()
        files = synthetic_weight_files(tmp)
        a.db = synthetic_stb_db(os.path.join(tmp, 'stb_eval.bin'), a.limit or 4)
    else:
        files = ['%s/handsegnet-rhd.pickle' % a.weights_dir, '%s/posenet3d-rhd-stb.pickle' % a.weights_dir]   # :66-67
    net.init(None, weight_files=files)
    dataset = BinaryDbReaderSTB(mode='evaluation', shuffle=False, use_wrist_coord=False, path_to_db=a.db,
                                engine=net.engine)                                                   # :46
    util = EvalUtil()
    for i, data in enumerate(dataset.get()):
        if a.limit and i >= a.limit:
            break
        u8 = np.rint((data['image'] + 0.5) * 255.0).astype(np.uint8)
        # eval_full.py:50: tf.image.resize_images(data['image'], (240, 320)) -- fused with x/255-0.5 on device
        _, _, _, _, _, coord3d_pred_v = net.inference_from_uint8(u8, data['hand_side'], True, net_size=(240, 320))
        keypoint_xyz21 = np.squeeze(data['keypoint_xyz21'])
        coord3d_pred_v = np.squeeze(coord3d_pred_v) * np.squeeze(data['keypoint_scale'])             # :81
        keypoint_xyz21 = keypoint_xyz21 - keypoint_xyz21[0, :]                                        # :84
        util.feed(keypoint_xyz21, np.ones_like(np.squeeze(data['keypoint_vis21'])), coord3d_pred_v)   # :86
    mean, median, auc, pck_curve_all, threshs = util.get_measures(0.0, 0.050, 20)                     # :92
    print('Evaluation results:')
    print('Average mean EPE: %.3f mm' % (mean * 1000))
    print('Average median EPE: %.3f mm' % (median * 1000))
    print('Area under curve between 0mm - 50mm: %.3f' % auc)
    print_result(mean, median, auc)
