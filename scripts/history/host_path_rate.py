"""PCIe-inclusive rate of the host-buffer entry points (hp3d_infer_full / hp3d_infer_full_u8): NumPy arrays in, all six
outputs back in NumPy arrays, B=32, 320x320 (never bench.py's `value`, which keeps inputs resident in HBM)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hand3d_amd import Engine, synth

eng = Engine(0)
eng.load_weight_dict(synth.make_weights())
eng.finalize_weights(0)
B, H, W = 32, 320, 320
img = synth.make_batch(1, B, H, W)
hs = synth.hand_sides(B)
u8 = np.clip((img + 0.5) * 255.0, 0, 255).astype(np.uint8)
for name, fn in (('infer_full (f32 host in, 6 outputs out)', lambda: eng.infer_full(img, hs)),
                 ('infer_full (coord3d only out)', lambda: eng.infer_full(img, hs, outputs=('coord3d',))),
                 ('infer_full_u8 (uint8 host in, 6 outputs out)', lambda: eng.infer_full_u8(u8, hs, H, W))):
    fn(); fn()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    print('%-48s %7.1f ms/batch  %7.1f images/s' % (name, dt * 1e3, B / dt))
