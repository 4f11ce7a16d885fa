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
