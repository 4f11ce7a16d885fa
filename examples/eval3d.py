#!/usr/bin/env python
"""eval3d.py of the reference (eval3d.py:42-105): PosePriorNetwork(VARIANT) on GT score maps of RHD-e."""
import os
import tempfile

import numpy as np

from common import parser, print_result, synthetic_rhd_db

if __name__ == '__main__':
    ap = parser(__doc__)
    ap.add_argument('--db', default=None)
    ap.add_argument('--variant', default='proposed',
                    choices=['direct', 'bottleneck', 'local', 'local_w_xyz_loss', 'proposed'])          # eval3d.py:43-47
    a = ap.parse_args()
    from hand3d_amd import synth
    from hand3d_amd.data import BinaryDbReader, binary_format as fmt
    from hand3d_amd.nets.PosePriorNetwork import PosePriorNetwork
    from hand3d_amd.utils.general import EvalUtil

    net = PosePriorNetwork(a.variant, device=a.device)                                                # :56
    if a.synthetic:
        tmp = tempfile.mkdtemp()
        w = synth.make_weights(bottleneck=(a.variant == 'bottleneck'))
        files = synth.write_weight_files(tmp, {k: v for k, v in w.items() if not k.startswith(('HandSegNet', 'PoseNet2D'))})[1:]
        a.db = synthetic_rhd_db(os.path.join(tmp, 'rhd_evaluation.bin'), a.limit or 4)
    else:
        files = ['%s/lifting-%s.pickle' % (a.weights_dir, a.variant)]                                  # :76
    net.init(None, weight_files=files)
    dataset = BinaryDbReader(mode='evaluation', shuffle=False, hand_crop=True, use_wrist_coord=False,
                             path_to_db=a.db, engine=net.engine)                                     # :48
    util = EvalUtil()
    for i, data in enumerate(dataset.get()):
        if a.limit and i >= a.limit:
            break
        coord3d_pred, _, _ = net.inference(data['scoremap'], data['hand_side'], True)                # :60
        keypoint_xyz21 = np.squeeze(data['keypoint_xyz21'])
        coord3d_pred_v = np.squeeze(coord3d_pred) * np.squeeze(data['keypoint_scale'])               # :89
        keypoint_xyz21 = keypoint_xyz21 - keypoint_xyz21[0, :]                                        # :92
        util.feed(keypoint_xyz21, np.ones_like(np.squeeze(data['keypoint_vis21'])), coord3d_pred_v)   # :95
    mean, median, auc, _, _ = util.get_measures(0.0, 0.050, 20)
    print('Evaluation results for %s:' % a.variant)
    print('Average mean EPE: %.3f mm' % (mean * 1000))
    print('Average median EPE: %.3f mm' % (median * 1000))
    print('Area under curve: %.3f' % auc)
    print_result(mean, median, auc)
