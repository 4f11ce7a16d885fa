"""The Python call surface mirrors the reference's (names, argument order, return order, errors):
nets/ColorHandPose3DNetwork.py:28-219, nets/PosePriorNetwork.py:30-95, weight loading :34-59."""
import inspect
import os
import pickle

import numpy as np
import pytest

from hand3d_amd import ColorHandPose3DNetwork, PosePriorNetwork, arch, synth


def test_method_surface_matches_reference():
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(ColorHandPose3DNetwork.init) == ['self', 'session', 'weight_files', 'exclude_var_list']
    assert sig(ColorHandPose3DNetwork.inference) == ['self', 'image', 'hand_side', 'evaluation']
    assert sig(ColorHandPose3DNetwork.inference2d) == ['self', 'image']
    assert sig(ColorHandPose3DNetwork.inference_detection) == ['self', 'image', 'train']
    assert sig(ColorHandPose3DNetwork.inference_pose2d) == ['self', 'image_crop', 'train']
    assert sig(PosePriorNetwork.__init__)[:2] == ['self', 'variant']
    assert sig(PosePriorNetwork.init) == ['self', 'session', 'weight_files', 'exclude_var_list']
    assert sig(PosePriorNetwork.inference) == ['self', 'scoremap', 'hand_side', 'evaluation']


def test_arch_tables_match_survey():
    shapes = arch.var_shapes(arch.all_layers())
    assert sum(int(np.prod(s)) for s in shapes.values()) == 34996515          # SURVEY.md App. A grand total
    assert shapes['PoseNet2D/conv6_1/weights'] == (7, 7, 149, 128)
    assert shapes['PosePrior/fc_rel0/weights'] == (2050, 512) and shapes['ViewpointNet/fc_vp0/weights'] == (4098, 256)
    assert shapes['HandSegNet/conv6_2/weights'] == (1, 1, 512, 2)
    f = arch.pipeline_flops(240, 320)
    assert abs(f['total'] / 1e9 - 121.83) < 0.01 and abs(arch.pipeline_flops(320, 320)['total'] / 1e9 - 142.26) < 0.01


def test_weight_files_loading_and_exclusion(tmp_path, emu_engine, synth_weights):
    paths = synth.write_weight_files(str(tmp_path), synth_weights)
    with open(paths[1], 'rb') as f:
        d = pickle.load(f)
    assert 'PoseNet2D/conv1_1/weights' in d and d['PosePrior/fc_xyz/weights'].dtype == np.float32
    net = ColorHandPose3DNetwork(engine=emu_engine)
    assert net.crop_size == 256 and net.num_kp == 21
    net.init(None, weight_files=paths)
    assert emu_engine.nets_mask() & 15 == 15
    # eval2d.py:78-79: posenet weights without the lifting nets
    from hand3d_amd import Engine
    e2 = Engine(0, path=os.environ.get('HP3D_LIB') or emu_engine.lib._name)
    net2 = ColorHandPose3DNetwork(engine=e2)
    net2.init(None, weight_files=paths, exclude_var_list=['PosePrior', 'ViewpointNet'])
    assert e2.nets_mask() & 15 == 3
    with pytest.raises(Exception):
        net2.inference(np.zeros((1, 16, 24, 3), np.float32), np.array([[1., 0.]], np.float32), True)  # lifting nets absent
    with pytest.raises(AssertionError, match="File not found."):
        net2.init(None, weight_files=[str(tmp_path / 'missing.pickle')])
    e2.close()


def test_npz_weights_round_trip(tmp_path, emu_engine, synth_weights):
    """BASELINE.json's north star says ".npz weights"; the reference's loader is pickle (nets/ColorHandPose3DNetwork.py:45-59).
    Both are accepted: pickle -> export_npz / pickle_to_npz -> init(.npz) gives the same engine state and the same outputs."""
    from hand3d_amd import Engine
    from hand3d_amd.nets.ColorHandPose3DNetwork import pickle_to_npz, read_weight_file
    paths = synth.write_weight_files(str(tmp_path), synth_weights)
    lean = ColorHandPose3DNetwork(engine=Engine(0, path=emu_engine.lib._name))
    lean.init(None, weight_files=paths, exclude_var_list=['HandSegNet', 'PoseNet2D'])      # (the two small nets: every init packs what it loads)
    assert lean.weight_dict == {}, "weights are only retained on request"
    with pytest.raises(AssertionError, match="keep_weights"):
        lean.export_npz(str(tmp_path / 'no.npz'))
    lean.engine.close()            # (one engine at a time beside the session's: each holds a 1.3 GB weight blob)
    net = ColorHandPose3DNetwork(engine=emu_engine, keep_weights=True)
    net.init(None, weight_files=paths)
    npz = str(tmp_path / 'all.npz')
    net.export_npz(npz)
    back = read_weight_file(npz)
    assert sorted(back) == sorted(synth_weights)
    for k, v in synth_weights.items():
        assert back[k].dtype == np.float32 and back[k].shape == v.shape and np.array_equal(back[k], v), k
    # the converter on the files gives the same archive content; exclusion applies to .npz files like to pickles
    npz2 = str(tmp_path / 'conv.npz')
    keys = pickle_to_npz(paths, npz2)
    back2 = read_weight_file(npz2)
    assert keys == sorted(synth_weights) and all(np.array_equal(back2[k], back[k]) for k in keys)
    e2 = Engine(0, path=emu_engine.lib._name)
    net2 = ColorHandPose3DNetwork(engine=e2)
    net2.init(None, weight_files=[npz])
    assert e2.nets_mask() & 15 == 15
    # the same engine state: every net gives the same bits (stage by stage on small inputs -- the whole path would put a 256 x 256 crop
    # through the interpreter twice: two minutes)
    rng = np.random.RandomState(5)
    img = (rng.rand(1, 16, 24, 3).astype(np.float32) - 0.5)
    hs = np.array([[1., 0.]], np.float32)
    sm = (rng.randn(1, 32, 32, 21) * 0.3).astype(np.float32)
    for ea, eb, na, nb in ((emu_engine, e2, net, net2),):
        for x, y in zip(ea.handsegnet(img, want_small=True), eb.handsegnet(img, want_small=True)):
            assert np.array_equal(x, y)
        for x, y in zip(na.inference_pose2d(img[:, :16, :16]), nb.inference_pose2d(img[:, :16, :16])):
            assert np.array_equal(x, y)
        for x, y in zip(ea.pose3d(sm, hs), eb.pose3d(sm, hs)):
            assert np.array_equal(x, y)
    e2.close()
    e3 = Engine(0, path=emu_engine.lib._name)
    net3 = ColorHandPose3DNetwork(engine=e3)
    net3.init(None, weight_files=[npz], exclude_var_list=['HandSegNet', 'PoseNet2D'])
    assert e3.nets_mask() & 15 == 12
    with pytest.raises(AssertionError, match="File not found."):
        net3.init(None, weight_files=[str(tmp_path / 'missing.npz')])
    e3.close()
    pp = PosePriorNetwork('direct', engine=Engine(0, path=emu_engine.lib._name))
    pp.init(None, weight_files=[npz], exclude_var_list=['HandSegNet', 'PoseNet2D', 'ViewpointNet'])
    assert pp.engine.nets_mask() & 15 == 4
    pp.engine.close()


def test_error_behaviour(emu_engine, synth_weights):
    net = ColorHandPose3DNetwork(engine=emu_engine)
    net.init_from_dict(synth_weights)
    img = np.zeros((1, 16, 24, 3), np.float32)
    with pytest.raises(NotImplementedError):
        net.inference(img, np.array([[1., 0.]], np.float32), False)      # dropout path = training
    with pytest.raises(NotImplementedError):
        net.inference_pose2d(img, train=True)
    with pytest.raises(AssertionError):
        net.inference_detection(np.zeros((1, 12, 24, 3), np.float32))     # H < 16
    with pytest.raises(AssertionError):
        emu_engine.set_weight('HandSegNet/conv1_1/weights', np.zeros((3, 3, 3, 63), np.float32))   # bad shape
    with pytest.raises(AssertionError):
        emu_engine.set_weight('HandSegNet/nope/weights', np.zeros((1,), np.float32))
    with pytest.raises(AssertionError, match="Unknown variant."):
        PosePriorNetwork('bogus', engine=emu_engine).inference(np.zeros((1, 256, 256, 21), np.float32),
                                                                np.array([[1., 0.]], np.float32), True)


def test_incomplete_network_is_rejected(emu_engine, synth_weights):
    from hand3d_amd import Engine, Hp3dError
    e2 = Engine(0, path=emu_engine.lib._name)
    part = {k: v for k, v in synth_weights.items() if k.startswith('HandSegNet/conv1')}
    e2.load_weight_dict(part)
    with pytest.raises(Hp3dError):
        e2.finalize_weights()
    e2.close()


def test_timing_and_comm_entry_points_on_interpreter(emu_engine, synth_weights):
    """hp3d_get_timing needs no GPU; the RCCL entry points exist in the CPU interpreter build but refuse to run
    (HP3D_ERR_UNSUPPORTED -> NotImplementedError), they never pretend."""
    t = emu_engine.get_timing()
    assert set(t) == {'HandSegNet', 'mask_crop', 'PoseNet2D', 'lifting', 'total'} and t['total'] == 0.0
    with pytest.raises(NotImplementedError):
        emu_engine.comm_init(0, 1, b'\0' * 128)
    with pytest.raises(NotImplementedError):
        emu_engine.bcast_weights(0)
    with pytest.raises(AssertionError):
        emu_engine.comm_init(0, 1, b'short')
    emu_engine.set_option('micro_batch', '0')
    emu_engine.set_option('micro_batch', '32')
    emu_engine.set_option('micro_batch', 'auto')
    for bad in ('-1', 'x', '3.5', ''):
        with pytest.raises(AssertionError):
            emu_engine.set_option('micro_batch', bad)
