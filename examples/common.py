"""Shared bits of the example harnesses (mirrors of the reference's run.py / eval*.py call sites)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parser(desc):
    ap = argparse.ArgumentParser(description=desc)
    ap.add_argument('--weights-dir', default='./weights', help='directory with the reference weight pickles')
    ap.add_argument('--synthetic', action='store_true',
                    help='no released weights/data on this box: seeded synthetic weights and inputs (plumbing check)')
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--limit', type=int, default=0, help='stop after N samples (0 = all)')
    return ap


def synthetic_weight_files(tmpdir, bottleneck=False):
    from hand3d_amd import synth
    return synth.write_weight_files(tmpdir, synth.make_weights(bottleneck=bottleneck))


def synthetic_rhd_db(path, n, seed=0):
    """n seeded records in the RHD binary layout (create_binary_db.py:44-88): random frame, one left-hand blob in the
    part mask, all 42 keypoints visible inside it."""
    import numpy as np
    from hand3d_amd.data import binary_format as fmt
    rng = np.random.default_rng(seed)
    with open(path, 'wb') as f:
        for _ in range(n):
            mask = np.zeros((320, 320), np.uint8)
            mask[100:200, 80:220] = 5
            f.write(fmt.pack_rhd_record(rng.integers(0, 256, (320, 320, 3), dtype=np.uint8), mask, rng.normal(0, .05, (42, 3)),
                                        rng.uniform(90, 210, (42, 2)), np.ones(42), np.eye(3)))
    return path


def synthetic_stb_db(path, n, seed=0):
    """n seeded records in the STB binary layout (data/stb/write_binary_record.m)."""
    import numpy as np
    from hand3d_amd.data import binary_format as fmt
    rng = np.random.default_rng(seed)
    with open(path, 'wb') as f:
        for _ in range(n):
            uvv = np.concatenate([rng.uniform(100, 400, (21, 2)), np.ones((21, 1))], 1)
            f.write(fmt.pack_stb_record(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8), rng.normal(0, 40, (21, 3)), uvv))
    return path


def print_result(mean, median, auc):
    """Machine-readable copy of the three reported numbers (full precision) for the tests."""
    import json
    print('RESULT ' + json.dumps({'mean': float(mean), 'median': float(median), 'auc': float(auc)}))
