"""Parity pinned to the reference's OWN code, executed here.

TensorFlow 1.3 cannot be installed in this container, but every line the reference itself wrote is Python.
`oracle/refrun.py` imports the reference's modules UNMODIFIED from /root/reference with `oracle/tfshim` standing
in for `tensorflow` (an eager NumPy implementation of the ~70 TF calls the reference makes, whose kernels are
oracle/tf_ops.py).  These tests then check, on the same seeded inputs,

  * oracle/nets.py, oracle/general.py, oracle/relative_trafo.py  ==  the reference's nets / glue / trafo code;
  * hand3d_amd.utils.general (product host helpers)              ==  the reference's TF-free NumPy functions,
    run as they are (detect_keypoints, trafo_coords, EvalUtil, calc_auc: utils/general.py:331-357,522-611,654-659);
  * hand3d_amd.data readers + record packers                      ==  the reference's writer
    (create_binary_db.py:44-88, ast-extracted: the module itself is a script) and its BinaryDbReader /
    BinaryDbReaderSTB (data/*.py) reading the same bytes.

What this does NOT pin: the arithmetic inside TensorFlow's kernels (restated in oracle/tf_ops.py; a box with
TF 1.x closes that with scripts/make_tf_fixtures.py -> tests/test_tf13_fixtures.py).

/root/reference is only present in the build container: without it the whole module is skipped (the fixtures
scripts/make_ref_fixtures.py wrote from the same runs travel instead: tests/golden/ref_*.npz).
"""
import ast
import os

import numpy as np
import pytest

from hand3d_amd import synth
from hand3d_amd.utils import general as PG          # the product's host helpers
from oracle import general as G
from oracle import nets as N
from oracle import refrun
from oracle import relative_trafo as RT

pytestmark = pytest.mark.skipif(not refrun.available(),
                                reason="reference tree (%s) not present on this box: fixtures in tests/golden/ref_*.npz "
                                       "carry these runs instead" % refrun.REFERENCE)


@pytest.fixture(scope='module')
def ref():
    r = refrun.load()
    assert r is not None
    assert not hasattr(r, 'reader_import_error'), r.reader_import_error
    return r


@pytest.fixture(scope='module')
def weights():
    return synth.make_weights()


@pytest.fixture(scope='module')
def ref_net(ref, weights):
    ref.reset()
    net = ref.ColorHandPose3DNetwork()
    # the released weights come as two pickles (handsegnet-*, posenet3d-*): load them the same way
    ref.init(net, weights, split=[('HandSegNet',), ('PoseNet2D', 'PosePrior', 'ViewpointNet')])
    return net


def _np(x):
    return np.asarray(x).view(np.ndarray)


# ----------------------------------------------------------------------------------------------- networks
@pytest.mark.parametrize('seed', [0, 3])
def test_inference_reference_code_equals_oracle(ref, ref_net, weights, seed):
    """ColorHandPose3DNetwork.inference (CHP3D.py:61-99), BASELINE config 1 images."""
    tf = ref.tf
    img = synth.make_batch(seed, 1, 240, 320)
    hs = np.array([[0.0, 1.0]], np.float32) if seed % 2 else np.array([[1.0, 0.0]], np.float32)
    out = [_np(o) for o in ref_net.inference(tf.constant(img), tf.constant(hs), tf.constant(True))]
    exp = N.inference(weights, img, hs, True)
    names = ['hand_scoremap', 'image_crop', 'scale_crop', 'center', 'keypoints_scoremap', 'keypoint_coord3d']
    for n, a, b in zip(names, out, exp):
        assert a.shape == b.shape and a.dtype == np.float32, n
    for i in (0, 1, 2, 3, 4):           # same kernels, same composition: bit-identical
        assert np.array_equal(out[i], exp[i]), names[i]
    assert np.abs(out[5] - exp[5]).max() <= 1e-6


def test_inference_batch_and_inference2d_order(ref, ref_net, weights):
    """B = 2 (left + right hand) and inference2d's different return order (CHP3D.py:101-129)."""
    tf = ref.tf
    img = synth.make_batch(10, 2, 240, 320)
    hs = synth.hand_sides(2)
    out = [_np(o) for o in ref_net.inference(tf.constant(img), tf.constant(hs), tf.constant(True))]
    exp = N.inference(weights, img, hs, True)
    for a, b in zip(out, exp):
        assert np.abs(a - b).max() <= 1e-6
    o2 = [_np(o) for o in ref_net.inference2d(tf.constant(img))]
    e2 = N.inference2d(weights, img)
    assert [a.shape for a in o2] == [(2, 256, 256, 21), (2, 256, 256, 3), (2, 1), (2, 2)]
    for a, b in zip(o2, e2):
        assert np.array_equal(a, b)
    assert np.array_equal(o2[0], out[4]) and np.array_equal(o2[1], out[1])


def test_subnet_entry_points(ref, ref_net, weights):
    """inference_detection returns a list of 1, inference_pose2d a list of 3 (CHP3D.py:131-219)."""
    tf = ref.tf
    img = synth.make_batch(20, 1, 64, 96)
    det = ref_net.inference_detection(tf.constant(img))
    assert len(det) == 1 and np.array_equal(_np(det[0]), N.handsegnet(weights, img)[1][0])
    crop = synth.make_batch(21, 1, 64, 64)
    maps = ref_net.inference_pose2d(tf.constant(crop))
    exp = N.posenet2d(weights, crop)
    assert len(maps) == 3 and all(np.array_equal(_np(a), b) for a, b in zip(maps, exp))


@pytest.mark.parametrize('variant', ['direct', 'bottleneck', 'local', 'local_w_xyz_loss', 'proposed'])
def test_poseprior_network_variants(ref, variant):
    """PosePriorNetwork(variant).inference (nets/PosePriorNetwork.py:59-95), incl. bone_rel_trafo_inv."""
    tf = ref.tf
    w = synth.make_weights(bottleneck=(variant == 'bottleneck'))
    w = {k: v for k, v in w.items() if k.startswith(('PosePrior', 'ViewpointNet'))}
    ref.reset()
    net = ref.PosePriorNetwork(variant)
    ref.init(net, w)
    rng = np.random.default_rng(5)
    sm = (rng.standard_normal((2, 256, 256, 21)) * 0.3).astype(np.float32)
    hs = synth.hand_sides(2)
    rel, c3d, R = net.inference(tf.constant(sm), tf.constant(hs), tf.constant(True))
    erel, ec3d, eR = N.poseprior_network(w, variant, sm, hs)
    assert np.abs(_np(rel) - erel).max() <= 2e-6 and np.abs(_np(c3d) - ec3d).max() <= 1e-6
    assert (R is None) == (eR is None)
    if R is not None:
        assert np.abs(_np(R) - eR).max() <= 1e-6


def test_unknown_variant_and_missing_file(ref):
    tf = ref.tf
    with pytest.raises(AssertionError, match="Unknown variant."):
        ref.PosePriorNetwork('nope').inference(tf.constant(np.zeros((1, 256, 256, 21), np.float32)),
                                               tf.constant(np.array([[1., 0.]], np.float32)), tf.constant(True))
    with pytest.raises(AssertionError, match="File not found."):
        ref.ColorHandPose3DNetwork().init(tf.Session(), weight_files=['/nonexistent.pickle'])
    from hand3d_amd.nets.ColorHandPose3DNetwork import load_weight_files
    with pytest.raises(AssertionError, match="File not found."):
        load_weight_files(None, ['/nonexistent.pickle'])


# ----------------------------------------------------------------------------------------------- mask / crop glue
def _blob_scoremap(H, W, blobs, strength=4.0, seed=0):
    rng = np.random.default_rng(seed)
    sm = np.zeros((1, H, W, 2), np.float32)
    sm[..., 0] = 1.0
    for (y0, y1, x0, x1, s) in blobs:
        sm[0, y0:y1, x0:x1, 1] = s * strength
    sm += (rng.standard_normal(sm.shape) * 0.01).astype(np.float32)
    return sm


@pytest.mark.parametrize('case', ['one_blob', 'two_blobs_gap10', 'two_blobs_gap11', 'empty', 'full', 'border'])
@pytest.mark.parametrize('reduce_id', ['inf', 'fltmax'])
def test_mask_glue_reference_code_equals_oracle(ref, case, reduce_id, monkeypatch):
    """single_obj_scoremap + calc_center_bb (utils/general.py:233-328) on engineered score maps."""
    tf = ref.tf
    H, W = 120, 160
    blobs = {'one_blob': [(30, 70, 40, 90, 1.0)],
             # the 21x21 dilation bridges a gap of 10 background pixels but not 11
             'two_blobs_gap10': [(30, 60, 20, 50, 1.2), (30, 60, 60, 90, 1.0)],
             'two_blobs_gap11': [(30, 60, 20, 50, 1.2), (30, 60, 61, 90, 1.0)],
             'empty': [], 'full': [(0, H, 0, W, 1.0)], 'border': [(0, 12, 150, W, 1.0)]}[case]
    sm = _blob_scoremap(H, W, blobs)
    monkeypatch.setattr(G, 'EMPTY_REDUCE', reduce_id)
    mask = ref.general.single_obj_scoremap(tf.constant(sm))
    center, bb, size = ref.general.calc_center_bb(mask)
    emask = G.single_obj_scoremap(sm)
    ecenter, ebb, esize = G.calc_center_bb(emask)
    assert np.array_equal(_np(mask), emask)
    assert np.array_equal(_np(center), ecenter) and np.array_equal(_np(size), esize)
    if case != 'empty':
        assert np.array_equal(_np(bb), ebb)
        assert emask.sum() > 0
    else:
        assert emask.sum() == 0 and float(esize[0, 0]) == 100.0
        assert ecenter[0].tolist() == ([160.0, 160.0] if reduce_id == 'inf' else [0.0, 0.0])
    if case == 'two_blobs_gap11':       # the weaker blob must not be reached
        assert emask[0, 30:60, 61:90, 0].sum() == 0


def test_crop_image_from_xy_and_scale(ref):
    """crop_image_from_xy (utils/general.py:163-196) + the scale clip of CHP3D.py:84-85."""
    tf = ref.tf
    rng = np.random.default_rng(7)
    img = rng.uniform(-0.5, 0.5, (4, 60, 80, 3)).astype(np.float32)
    center = np.array([[30, 40], [0, 0], [59.5, 79.5], [10.25, 70.75]], np.float32)
    size = np.array([[40.0], [0.0], [500.0], [17.3]], np.float32)
    cs = size * np.float32(1.25)
    with np.errstate(divide='ignore'):
        scale = _np(tf.minimum(tf.maximum(256 / tf.constant(cs), 0.25), 5.0))
    assert np.array_equal(scale, G.scale_from_crop_size(size))
    out = ref.general.crop_image_from_xy(tf.constant(img), tf.constant(center), 256, scale=tf.constant(scale))
    assert np.array_equal(_np(out), G.crop_image_from_xy(img, center, 256, scale))


def test_bone_rel_trafo_roundtrip(ref):
    """utils/relative_trafo.py:243-295 (inverse, PosePriorNetwork `local*`) and :148-240 (forward)."""
    tf = ref.tf
    rng = np.random.default_rng(11)
    xyz = rng.normal(0, 1, (3, 21, 3)).astype(np.float32)
    xyz[:, 0] = 0
    rel = _np(ref.relative_trafo.bone_rel_trafo(tf.constant(xyz)))
    assert np.abs(rel - RT.bone_rel_trafo(xyz)).max() <= 2e-5
    back = _np(ref.relative_trafo.bone_rel_trafo_inv(tf.constant(rel)))
    assert np.abs(back - RT.bone_rel_trafo_inv(rel)).max() <= 2e-5
    assert np.abs(back - xyz).max() <= 1e-3


# ----------------------------------------------------------------------------------------------- host helpers (product)
def test_product_host_helpers_equal_reference_functions(ref):
    """The product's detect_keypoints / trafo_coords / EvalUtil / calc_auc against the reference's NumPy functions."""
    R = ref.general
    rng = np.random.default_rng(3)
    for trial in range(4):
        sm = rng.standard_normal((40, 56, 21)).astype(np.float32)
        if trial == 1:          # ties: first maximum in row-major order wins
            sm[5, 7, :] = sm[20, 3, :] = 9.0
        if trial == 2:
            sm[:] = 0.0
        a = R.detect_keypoints(sm[None] if trial == 3 else sm)
        b = PG.detect_keypoints(sm[None] if trial == 3 else sm)
        assert a.dtype == b.dtype == np.float64 and np.array_equal(a, b)
        c, s = np.array([[120.5, 160.25]]), np.array([[1.7]])
        assert np.array_equal(R.trafo_coords(a, c, s, 256), PG.trafo_coords(b, c, s, 256))
    with pytest.raises(AssertionError):
        PG.detect_keypoints(np.zeros((21, 40, 56), np.float32))
    ru, pu = R.EvalUtil(), PG.EvalUtil()
    for i in range(50):
        gt = rng.normal(0, 10, (21, 2))
        pr = gt + rng.normal(0, 4, (21, 2))
        vis = rng.uniform(size=21) > 0.3
        vis[7] = False                  # a keypoint that never receives data
        ru.feed(gt[None], vis[None], pr)
        pu.feed(gt[None], vis[None], pr)
    with pytest.warns(DeprecationWarning) if hasattr(np, 'trapz') and np.lib.NumpyVersion(np.__version__) >= '2.0.0' \
            else _nullcontext():
        rm = ru.get_measures(0.0, 30.0, 20)
    pm = pu.get_measures(0.0, 30.0, 20)
    for a, b in zip(rm, pm):
        assert np.allclose(a, b, rtol=1e-13, atol=0)
    for k in (0, 7):
        assert ru._get_pck(k, 5.0) == pu._get_pck(k, 5.0)
        assert ru._get_epe(k) == pu._get_epe(k)
    x = np.linspace(0, 1, 9)
    y = rng.uniform(size=9)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        assert np.isclose(R.calc_auc(x, y), PG.calc_auc(x, y), rtol=1e-14)


class _nullcontext(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


# ----------------------------------------------------------------------------------------------- record writer / readers
def _reference_write_to_binary():
    """create_binary_db.py is a script (it opens the dataset at import): take the function definition only."""
    path = os.path.join(refrun.REFERENCE, 'create_binary_db.py')
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'write_to_binary']
    assert len(fn) == 1
    ns = {}
    exec(compile(ast.Module(body=[ast.parse('import struct').body[0]] + fn, type_ignores=[]), path, 'exec'), ns)
    return ns['write_to_binary']


def _rhd_sample(rng, left):
    img = rng.integers(0, 256, (320, 320, 3), dtype=np.uint8)
    mask = np.zeros((320, 320), np.uint8)
    mask[100:180, 90:200] = 5 if left else 20
    mask[10:20, 10:30] = 20 if left else 5
    xyz = rng.normal(0, 0.05, (42, 3)).astype(np.float32)
    uv = rng.uniform(60, 250, (42, 2)).astype(np.float32)
    vis = rng.uniform(size=42) > 0.2
    K = np.array([[283.1, 0, 160.0], [0, 283.1, 160.0], [0, 0, 1]], np.float32)
    return img, mask, xyz, uv, vis, K


@pytest.fixture()
def rhd_db(tmp_path, monkeypatch):
    """Three RHD records written by the REFERENCE's writer into ./data/bin/rhd_evaluation.bin of a scratch cwd."""
    rng = np.random.default_rng(0)
    samples = [_rhd_sample(rng, left=(i % 2 == 0)) for i in range(3)]
    os.makedirs(tmp_path / 'data' / 'bin')
    path = tmp_path / 'data' / 'bin' / 'rhd_evaluation.bin'
    write = _reference_write_to_binary()
    with open(path, 'wb') as f:
        for s in samples:
            write(f, *s)
    monkeypatch.chdir(tmp_path)
    return samples, str(path)


def test_record_bytes_equal_reference_writer(rhd_db):
    from hand3d_amd.data import binary_format as fmt
    samples, path = rhd_db
    raw = open(path, 'rb').read()
    assert len(raw) == 3 * fmt.RHD_RECORD_BYTES
    assert raw == b''.join(fmt.pack_rhd_record(*s) for s in samples)


@pytest.mark.parametrize('flags', [dict(use_wrist_coord=True), dict(use_wrist_coord=False),
                                   dict(use_wrist_coord=False, hand_crop=True),
                                   dict(use_wrist_coord=True, scale_to_size=True)])
def test_rhd_reader_equals_reference_reader(ref, rhd_db, emu_engine, flags):
    """Product reader vs the reference's BinaryDbReader (data/BinaryDbReader.py) on bytes the reference wrote; the call
    sites are eval2d.py:44, eval2d_gt_cropped.py:37, eval3d.py:48."""
    from hand3d_amd.data import BinaryDbReader
    samples, path = rhd_db
    ref.tf.FixedLengthRecordReader.rewind()
    rr = ref.BinaryDbReader(mode='evaluation', shuffle=False, **flags)
    pr = BinaryDbReader(mode='evaluation', shuffle=False, path_to_db=path, engine=emu_engine, **flags)
    it = pr.get()
    for _ in samples:
        rd = {k: _np(v) for k, v in rr.get().items()}
        pd = next(it)
        # everything the inference scripts read; training labels (canonical / local frames) are out of scope
        skip = {'keypoint_xyz21_local', 'keypoint_xyz21_can', 'rot_mat'}
        for k, v in rd.items():
            if k in skip:
                continue
            assert k in pd, k
            p = np.asarray(pd[k])
            assert p.shape == v.shape, (k, p.shape, v.shape)
            if v.dtype == np.bool_ or np.issubdtype(v.dtype, np.integer):
                assert np.array_equal(p, v), k
            elif k in ('scoremap',):
                assert np.abs(p - v).max() <= 2e-7, k        # exp(): libm vs NumPy SIMD
            elif k in ('image_crop',):
                assert np.abs(p - v).max() <= 1e-6, k
            else:
                assert np.allclose(p, v, rtol=1e-6, atol=1e-6), k


def test_stb_reader_equals_reference_reader(ref, tmp_path, monkeypatch):
    """BinaryDbReaderSTB (data/BinaryDbReaderSTB.py; eval_full.py:46): xyz in mm, uv+vis triplets, keypoint re-ordering."""
    from hand3d_amd.data import BinaryDbReaderSTB
    from hand3d_amd.data import binary_format as fmt
    rng = np.random.default_rng(2)
    os.makedirs(tmp_path / 'data' / 'stb')
    path = tmp_path / 'data' / 'stb' / 'stb_eval.bin'
    recs = []
    with open(path, 'wb') as f:
        for _ in range(2):
            img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
            xyz = rng.normal(0, 80, (21, 3)).astype(np.float32) + np.array([0, 0, 600], np.float32)
            uvv = np.concatenate([rng.uniform(100, 400, (21, 2)), (rng.uniform(size=(21, 1)) > 0.2)], 1).astype(np.float32)
            recs.append((img, xyz, uvv))
            f.write(fmt.pack_stb_record(img, xyz, uvv))
    monkeypatch.chdir(tmp_path)
    for wrist in (True, False):
        ref.tf.FixedLengthRecordReader.rewind()
        rr = ref.BinaryDbReaderSTB(mode='evaluation', shuffle=False, use_wrist_coord=wrist)
        it = BinaryDbReaderSTB(mode='evaluation', shuffle=False, use_wrist_coord=wrist, path_to_db=str(path)).get()
        for _ in recs:
            rd = {k: _np(v) for k, v in rr.get().items()}
            pd = next(it)
            for k in ('image', 'keypoint_xyz21', 'keypoint_uv21', 'keypoint_vis21', 'cam_mat', 'hand_side',
                      'keypoint_scale', 'keypoint_xyz21_normed', 'scoremap'):
                p, v = np.asarray(pd[k]), rd[k]
                assert p.shape == v.shape, (k, p.shape, v.shape)
                if v.dtype == np.bool_:
                    assert np.array_equal(p, v), k
                else:
                    assert np.allclose(p, v, rtol=1e-6, atol=2e-7), k


def test_real_tensorflow_branch_of_the_fixture_dump_runs_on_the_graph_facade(ref, tmp_path, monkeypatch):
    """scripts/make_tf_fixtures.py has two branches: the eager one that wrote tests/golden/ref_*.npz, and the one a TensorFlow box
    would run -- graph on `tf.placeholder`s, `sess.run(tf.global_variables_initializer())`, `net.init(sess, ...)` AFTER the graph is
    built, `sess.run(fetches, feed_dict)` ("its graph-mode branch has never executed anywhere").  oracle/tfshim's graph
    facade (trace on stand-in data, variable nodes, both `tf.cond` branches, replay on the fed values) executes exactly that branch
    over the reference's unmodified modules, and every array of every file it writes equals the committed eager fixture bit for bit."""
    import glob
    import sys
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts')
    monkeypatch.syspath_prepend(scripts)
    import make_ref_fixtures
    import make_tf_fixtures
    inputs, out = tmp_path / 'inputs', tmp_path / 'out'
    os.makedirs(out)
    make_ref_fixtures.export_inputs(str(inputs), with_c4=False)      # (the batch-of-8 file is the same code path as c1: left out for time)
    monkeypatch.setattr(G, 'EMPTY_REDUCE', G.EMPTY_REDUCE)          # (the dump switches it per case; restored afterwards)
    try:
        make_tf_fixtures.main(['--inputs', str(inputs), '--out', str(out), '--prefix', 'graph_'], tf=ref.tf, eager=False,
                              mods=(ref.ColorHandPose3DNetwork, ref.PosePriorNetwork, ref.general),
                              set_empty_reduce=lambda rid: setattr(G, 'EMPTY_REDUCE', rid))
    finally:
        ref.tf.reset_default_graph()
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    files = sorted(glob.glob(str(out / 'graph_*.npz')))
    assert len(files) == 5
    for f in files:
        a, b = np.load(f, allow_pickle=True), np.load(os.path.join(golden, 'ref_' + os.path.basename(f)[6:]), allow_pickle=True)
        assert sorted(a.files) == sorted(b.files), os.path.basename(f)
        for k in a.files:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (f, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == 'f'), (os.path.basename(f), k)
    assert 'make_tf_fixtures' in sys.modules
