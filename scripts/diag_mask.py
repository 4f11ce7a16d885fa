"""Are mask differences between kernel choices knife-edge pixels or real errors?  B = 32, 320x320 (the bench batch of the test)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand3d_amd import ColorHandPose3DNetwork, synth
w = synth.make_weights()
net = ColorHandPose3DNetwork(device=0)
net.init_from_dict(w)
img = synth.make_batch(3000, 32, 320, 320)
hs = synth.hand_sides(32)
outs = {}
for mode in ('0', 'auto', '1'):
    net.engine.set_option('wino2', mode)
    outs[mode] = net.engine.infer_full(img, hs, want_mask=True)
a = outs['0']
for mode in ('auto', '1'):
    b = outs[mode]
    d = np.abs(a['scoremap'] - b['scoremap'])
    print('wino2=%s vs 0: scoremap max diff %.3e, kpmap %.3e, coord3d %.3e' % (mode, d.max(), np.abs(a['kpmap'] - b['kpmap']).max(), np.abs(a['coord3d'] - b['coord3d']).max()))
    for i in range(32):
        if not np.array_equal(a['mask'][i], b['mask'][i]):
            da = a['scoremap'][i, ..., 1] > a['scoremap'][i, ..., 0]
            db = b['scoremap'][i, ..., 1] > b['scoremap'][i, ..., 0]
            fl = np.argwhere(da != db)
            marg = [float(a['scoremap'][i, y, x, 1] - a['scoremap'][i, y, x, 0]) for y, x in fl[:5]]
            print('  image %d: mask differs in %d px; det flips %d, margins there %s, max scoremap diff on this image %.3e' % (
                i, int((a['mask'][i] != b['mask'][i]).sum()), len(fl), ['%.2e' % m for m in marg], d[i].max()))
