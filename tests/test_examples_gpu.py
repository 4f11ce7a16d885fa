"""The example harnesses (mirrors of run.py / eval*.py) run end to end on the GPU with synthetic data."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cmd", [
    ['run.py', '--synthetic'],
    ['eval_full.py', '--synthetic', '--limit', '3'],
    ['eval2d.py', '--synthetic', '--limit', '3'],
    ['eval2d.py', '--synthetic', '--limit', '3', '--gt-cropped'],
    ['eval3d.py', '--synthetic', '--limit', '3', '--variant', 'proposed'],
    ['eval3d.py', '--synthetic', '--limit', '3', '--variant', 'local'],
    ['eval3d.py', '--synthetic', '--limit', '3', '--variant', 'bottleneck'],
], ids=lambda c: '_'.join(c))
def test_example_harness(cmd):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', cmd[0])] + cmd[1:], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert ('EPE' in out.stdout) or ('wrist_xyz' in out.stdout)
