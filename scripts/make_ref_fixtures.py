"""Writes tests/golden/ref_*.npz by EXECUTING THE REFERENCE'S OWN CODE (from /root/reference, unmodified) on top
of the NumPy TensorFlow stand-in (oracle/tfshim, oracle/refrun.py) with the seeded synthetic weights / inputs
of hand3d_amd/synth.py.  /root/reference does not exist on the GPU box; these files are what travels there:
tests/test_gpu_reference_fixtures.py compares the HIP path with them, tests/test_golden.py the oracle and the
CPU-interpreter build of the kernels.

What the numbers pin: the reference's Python (composition, glue arithmetic, host post-processing).  The TF kernels
underneath are oracle/tf_ops.py (restated TF 1.3 semantics).  The dump itself is scripts/make_tf_fixtures.py -- the
script a box with real TensorFlow 1.x runs on the same inputs (`--export-inputs DIR` writes them) to produce
tests/golden/tf13_*.npz.

    python scripts/make_ref_fixtures.py                         (build container, from the repo root)
    python scripts/make_ref_fixtures.py --export-inputs DIR     (inputs for the TensorFlow box; ~150 MB of weights)
"""
import argparse
import os
import pickle
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from hand3d_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
C1_SEEDS = (0, 1, 2, 3, 4)               # BASELINE config 1: 5 images, 240 x 320, B = 1
C3_SEED0 = 200
C4_SEED0 = 300                           # BASELINE config 4's per-GPU shard in small: one batch of 8, 240 x 320


def export_inputs(d, with_c4=True):
    """Everything scripts/make_tf_fixtures.py reads: plain .npy / .npz / protocol-2 pickles (old NumPy / Python 2 safe)."""
    os.makedirs(d, exist_ok=True)
    w = synth.make_weights(seed=42)
    synth.write_weight_files(d, w)                 # handsegnet-rhd.pickle + posenet3d-rhd-stb-slr-finetuned.pickle
    for name, ww in (('lifting.pickle', w), ('lifting-bottleneck.pickle', synth.make_weights(bottleneck=True))):
        with open(os.path.join(d, name), 'wb') as f:
            pickle.dump({k: v for k, v in ww.items() if k.startswith(('PosePrior', 'ViewpointNet'))}, f, protocol=2)
    np.save(os.path.join(d, 'c1_seeds.npy'), np.array(C1_SEEDS))
    np.save(os.path.join(d, 'c1_images.npy'), np.concatenate([synth.make_batch(s, 1, 240, 320) for s in C1_SEEDS]))
    np.save(os.path.join(d, 'c1_hand_sides.npy'),
            np.array([[0.0, 1.0] if s % 2 else [1.0, 0.0] for s in C1_SEEDS], np.float32))
    np.save(os.path.join(d, 'c3_seed0.npy'), np.array(C3_SEED0))
    np.save(os.path.join(d, 'c3_images.npy'), synth.make_batch(C3_SEED0, 2, 320, 320))
    if with_c4:
        np.save(os.path.join(d, 'c4_seed0.npy'), np.array(C4_SEED0))
        np.save(os.path.join(d, 'c4_images.npy'), synth.make_batch(C4_SEED0, 8, 240, 320))
        np.save(os.path.join(d, 'c4_hand_sides.npy'), synth.hand_sides(8))
    np.savez(os.path.join(d, 'mask_cases.npz'), **{c: synth.blob_scoremap(c) for c in synth.MASK_CASES})
    np.save(os.path.join(d, 'lifting_scoremaps.npy'), synth.lifting_scoremaps(5, 2))
    np.save(os.path.join(d, 'lifting_hand_sides.npy'), synth.hand_sides(2))
    rng = np.random.default_rng(3)
    gt = rng.normal(0, 10, (50, 21, 2))
    pred = gt + rng.normal(0, 4, (50, 21, 2))
    vis = rng.uniform(size=(50, 21)) > 0.3
    vis[:, 7] = False                                   # a keypoint that never receives data
    np.savez(os.path.join(d, 'evalutil_feeds.npz'), gt=gt, vis=vis, pred=pred)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--export-inputs', default=None)
    a = ap.parse_args()
    if a.export_inputs:
        export_inputs(a.export_inputs)
        print('inputs written to', a.export_inputs)
        return
    from oracle import general as G
    from oracle import refrun
    import make_tf_fixtures
    ref = refrun.load()
    assert ref is not None, "needs the reference tree at %s" % refrun.REFERENCE
    with tempfile.TemporaryDirectory() as d:
        export_inputs(d)
        make_tf_fixtures.main(['--inputs', d, '--out', OUT, '--prefix', 'ref_'], tf=ref.tf, eager=True,
                              mods=(ref.ColorHandPose3DNetwork, ref.PosePriorNetwork, ref.general),
                              set_empty_reduce=lambda rid: setattr(G, 'EMPTY_REDUCE', rid))


if __name__ == '__main__':
    main()
