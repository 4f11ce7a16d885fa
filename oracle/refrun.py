"""Runs the reference's OWN Python modules (from /root/reference, unmodified, never copied) on top of the
NumPy TensorFlow stand-in in oracle/tfshim -- TEST INFRASTRUCTURE ONLY.

    ref = refrun.load()                       # None when /root/reference is absent (e.g. on the GPU box)
    net = ref.ColorHandPose3DNetwork()
    ref.init(net, weight_dict)                # the reference's init() through a temporary pickle file
    outs = net.inference(image, hand_side, True)

Used by tests/test_reference_pin.py (oracle == reference code, here) and by scripts/make_ref_fixtures.py, which
writes tests/golden/ref_*.npz: those fixtures are what travels to the GPU box, where the HIP path is compared
with them (tests/test_gpu_parity.py::test_reference_fixtures_*).
"""
import importlib
import os
import pickle
import sys
import tempfile
import types

REFERENCE = os.environ.get('HP3D_REFERENCE', '/root/reference')
SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tfshim')
_REF_TOP = ('nets', 'utils', 'data')          # the reference's top-level packages (namespace packages: no __init__.py)


def available():
    return os.path.isfile(os.path.join(REFERENCE, 'nets', 'ColorHandPose3DNetwork.py'))


def _purge():
    for name in list(sys.modules):
        if name.split('.')[0] in _REF_TOP or name == 'tensorflow' or name.startswith('tensorflow.'):
            mod = sys.modules[name]
            f = getattr(mod, '__file__', None) or ''
            p = [str(x) for x in getattr(mod, '__path__', [])]
            if f.startswith((REFERENCE, SHIM)) or any(x.startswith((REFERENCE, SHIM)) for x in p) or not f:
                del sys.modules[name]


def load():
    """Imports the reference modules with the shim as `tensorflow`; returns a namespace or None."""
    if not available():
        return None
    saved = list(sys.path)
    _purge()
    sys.path[:0] = [SHIM, REFERENCE]
    try:
        tf = importlib.import_module('tensorflow')
        assert tf.__file__.startswith(SHIM), tf.__file__
        ns = types.SimpleNamespace(tf=tf)
        chp = importlib.import_module('nets.ColorHandPose3DNetwork')
        ppn = importlib.import_module('nets.PosePriorNetwork')
        ns.general = importlib.import_module('utils.general')
        ns.relative_trafo = importlib.import_module('utils.relative_trafo')
        ns.canonical_trafo = importlib.import_module('utils.canonical_trafo')
        assert chp.__file__.startswith(REFERENCE) and ns.general.__file__.startswith(REFERENCE)
        ns.ColorHandPose3DNetwork = chp.ColorHandPose3DNetwork
        ns.PosePriorNetwork = ppn.PosePriorNetwork
        try:
            ns.BinaryDbReader = importlib.import_module('data.BinaryDbReader').BinaryDbReader
            ns.BinaryDbReaderSTB = importlib.import_module('data.BinaryDbReaderSTB').BinaryDbReaderSTB
        except Exception as e:     # the readers pull in more of TF than the nets; report, don't hide
            ns.reader_import_error = e
    finally:
        sys.path[:] = saved
    ns.init = _init
    ns.reset = tf.reset_default_graph
    return ns


def _init(net, weight_dict, exclude_var_list=None, split=None):
    """Calls the reference's net.init(session, weight_files, exclude_var_list) with pickle files written from
    `weight_dict` (split: optional list of key-prefix tuples -> one file each, like the released weight sets)."""
    import tensorflow as tf          # the shim (load() ran first)
    groups = [weight_dict] if not split else [{k: v for k, v in weight_dict.items() if k.startswith(tuple(p))} for p in split]
    files = []
    try:
        for g in groups:
            fd, fn = tempfile.mkstemp(suffix='.pickle')
            with os.fdopen(fd, 'wb') as f:
                pickle.dump(g, f)
            files.append(fn)
        net.init(tf.Session(), weight_files=files, exclude_var_list=exclude_var_list)
    finally:
        for fn in files:
            os.unlink(fn)
