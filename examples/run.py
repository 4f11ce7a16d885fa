#!/usr/bin/env python
"""run.py of the reference (run.py:30-92) on the MI355X engine: five images -> 2-D / 3-D keypoints.

    python examples/run.py img1.png img2.png ...          (needs ./weights/*.pickle, like the reference)
    python examples/run.py --synthetic                    (seeded synthetic weights + images)
"""
import json
import tempfile

import numpy as np

from common import parser, synthetic_weight_files

if __name__ == '__main__':
    ap = parser(__doc__)
    ap.add_argument('images', nargs='*')
    a = ap.parse_args()
    from hand3d_amd import synth
    from hand3d_amd.nets.ColorHandPose3DNetwork import ColorHandPose3DNetwork
    from hand3d_amd.utils.general import detect_keypoints, trafo_coords

    net = ColorHandPose3DNetwork(device=a.device)                    # run.py:44
    if a.synthetic:
        net.init(None, weight_files=synthetic_weight_files(tempfile.mkdtemp()))
        frames = [(synth.make_image(i) + 0.5) * 255.0 for i in range(5)]
    else:
        net.init(None, weight_files=['%s/handsegnet-rhd.pickle' % a.weights_dir,
                                     '%s/posenet3d-rhd-stb-slr-finetuned.pickle' % a.weights_dir])   # run.py:53
        from PIL import Image
        frames = [np.asarray(Image.open(p).convert('RGB').resize((320, 240), Image.BILINEAR), np.float32) for p in a.images]
    hand_side_v = np.array([[1.0, 0.0]], np.float32)                 # run.py:40: left hand
    for i, image_raw in enumerate(frames):
        image_v = np.expand_dims((image_raw.astype('float') / 255.0) - 0.5, 0)                       # run.py:59
        hand_scoremap_v, image_crop_v, scale_v, center_v, keypoints_scoremap_v, keypoint_coord3d_v = \
            net.inference(image_v, hand_side_v, True)                                                  # run.py:61-64
        keypoint_coord3d_v = np.squeeze(keypoint_coord3d_v)
        coord_hw_crop = detect_keypoints(np.squeeze(keypoints_scoremap_v))                             # run.py:72
        coord_hw = trafo_coords(coord_hw_crop, center_v, scale_v, 256)                                 # run.py:73
        print(json.dumps({'image': i, 'center': center_v.tolist(), 'scale': float(scale_v[0, 0]),
                          'wrist_hw': coord_hw[0].tolist(), 'wrist_xyz': keypoint_coord3d_v[0].tolist()}))
