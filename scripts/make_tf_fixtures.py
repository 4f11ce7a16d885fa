"""Dumps per-stage outputs of the REFERENCE (lmb-freiburg/hand3d, run unmodified) for the seeded synthetic inputs.

Two ways to run it -- the SAME code path either way:

  * on a box that has real TensorFlow 1.x (>= 1.3, the reference's pin; Python 2.7 / 3.5 / 3.6; this file needs only
    numpy + pickle + tensorflow and is written for that vintage):

        python scripts/make_ref_fixtures.py --export-inputs /tmp/hp3d_inputs       # in the build container (NumPy 2)
        python scripts/make_tf_fixtures.py --reference /path/to/hand3d --inputs /tmp/hp3d_inputs \
               --out tests/golden --prefix tf13_                                   # on the TensorFlow box

    -> tests/golden/tf13_*.npz, consumed by tests/test_tf13_fixtures.py (CPU: oracle vs TF; GPU: HIP vs TF).  That closes
    the one gap left in the parity chain: the arithmetic inside TensorFlow's kernels.

  * in the build container, through scripts/make_ref_fixtures.py, with oracle/tfshim standing in for tensorflow (eager
    NumPy) -> tests/golden/ref_*.npz.  This is also what keeps this script exercised where no TensorFlow exists -- both of its
    branches: `eager=False` over the stand-in's graph facade runs the placeholder / sess.run code below unchanged and must reproduce
    the eager files bit for bit (tests/test_reference_pin.py).

With real TF the graph is built on placeholders, `net.init(sess)` assigns the pickled weights, and `sess.run` evaluates;
with the eager stand-in `init` runs first and the network methods are called on the arrays directly.
"""
from __future__ import print_function

import argparse
import os
import pickle
import sys
import warnings

import numpy as np

VARIANTS = ('direct', 'bottleneck', 'local', 'local_w_xyz_loss', 'proposed')


class Runner(object):
    """Evaluates `build(**tensors) -> list of tensors` of a reference network on NumPy feeds."""

    def __init__(self, tf, eager):
        self.tf, self.eager = tf, eager

    def run(self, make_net, weight_files, build, feeds, exclude=None):
        tf = self.tf
        tf.reset_default_graph()
        net = make_net()
        if self.eager:
            net.init(tf.Session(), weight_files=weight_files, exclude_var_list=exclude)
            outs = build(net, **dict((k, tf.constant(v)) for k, v in feeds.items()))
            return [None if o is None else np.asarray(o).view(np.ndarray) for o in outs]
        phs = dict((k, tf.placeholder(tf.bool if v.dtype == np.bool_ else tf.float32, v.shape)) for k, v in feeds.items())
        outs = build(net, **phs)
        keep = [i for i, o in enumerate(outs) if o is not None]
        with tf.Session() as sess:
            sess.run(tf.global_variables_initializer())
            net.init(sess, weight_files=weight_files, exclude_var_list=exclude)
            vals = sess.run([outs[i] for i in keep], dict((phs[k], v) for k, v in feeds.items()))
        res = [None] * len(outs)
        for i, v in zip(keep, vals):
            res[i] = v
        return res


KNIFE_EDGE = 4e-6          # |p_fg - 1/2| below which a det pixel is a coin toss of the float32 summation order (SURVEY section 7 "Hard parts")


def knife_edge_pixels(hand_scoremap):
    """(min |p_fg - 1/2| per image, [[image, row, col], ...] of the pixels below KNIFE_EDGE) of a [B,H,W,2] score map -- the hand mask is a
    THRESHOLD of it (utils/general.py:240-245), so a kernel with another summation order may legitimately land such a pixel on the other side.
    Stored with the fixtures so that the GPU tests can tell a knife-edge flip from a defect."""
    s = hand_scoremap.astype(np.float64)
    fg = 1.0 / (1.0 + np.exp(s[..., 0] - s[..., 1]))
    m = np.abs(fg - 0.5)
    return m.reshape(m.shape[0], -1).min(1), np.argwhere(m < KNIFE_EDGE).astype(np.int32).reshape(-1, 3)


def main(argv=None, tf=None, eager=False, mods=None, set_empty_reduce=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference', help='checkout of lmb-freiburg/hand3d')
    ap.add_argument('--inputs', required=True, help='directory written by make_ref_fixtures.py --export-inputs')
    ap.add_argument('--out', required=True)
    ap.add_argument('--prefix', default='tf13_')
    a = ap.parse_args(argv)
    if tf is None:
        sys.path.insert(0, a.reference)
        import tensorflow as tf            # noqa: F811  (the real one)
        from nets.ColorHandPose3DNetwork import ColorHandPose3DNetwork
        from nets.PosePriorNetwork import PosePriorNetwork
        import utils.general as general
        print('TensorFlow', tf.__version__)
    else:
        ColorHandPose3DNetwork, PosePriorNetwork, general = mods
    run = Runner(tf, eager).run
    inp = lambda *p: os.path.join(a.inputs, *p)        # noqa: E731
    out = lambda n: os.path.join(a.out, a.prefix + n)  # noqa: E731
    true = np.array(True)
    wfiles = [inp('handsegnet-rhd.pickle'), inp('posenet3d-rhd-stb-slr-finetuned.pickle')]

    # ---- full pipeline (nets/ColorHandPose3DNetwork.py:61-99) + run.py:72-74 post-processing, B = 1, 240 x 320 ----------
    imgs, sides, seeds = np.load(inp('c1_images.npy')), np.load(inp('c1_hand_sides.npy')), np.load(inp('c1_seeds.npy'))
    d = {'seeds': seeds, 'shape': np.array(imgs.shape[1:3])}

    def full(net, image, hand_side, evaluation):
        o = list(net.inference(image, hand_side, evaluation))
        return o + [general.single_obj_scoremap(o[0])]
    for s, img, hs in zip(seeds, imgs, sides):
        hand_scoremap, image_crop, scale_crop, center, kp_scoremap, coord3d, mask = run(
            ColorHandPose3DNetwork, wfiles, full, dict(image=img[None], hand_side=hs[None], evaluation=true))
        kp = general.detect_keypoints(np.squeeze(kp_scoremap))
        uv = general.trafo_coords(kp, center, scale_crop, 256)
        k = 's%d_' % s
        d[k + 'hand_side'] = hs[None]
        d[k + 'hand_scoremap_sub'] = hand_scoremap[0, ::8, ::8, :]     # = the 30x40 net output (legacy resize keeps sources)
        d[k + 'mask_packed'] = np.packbits(mask[0, :, :, 0].astype(np.uint8))
        d[k + 'min_margin'], d[k + 'knife_iyx'] = knife_edge_pixels(hand_scoremap)
        d[k + 'center'], d[k + 'scale_crop'] = center, scale_crop
        d[k + 'image_crop_sub'] = image_crop[0, ::8, ::8, :]
        d[k + 'scoremap32'] = kp_scoremap[0, ::8, ::8, :]              # = PoseNet2D's last 32x32x21 map
        d[k + 'scoremap256_rows'] = kp_scoremap[0, 101:104, :, :]      # three interpolated rows of the 256x256 map
        d[k + 'scoremap256_sum'] = kp_scoremap[0].sum(axis=(0, 1), dtype=np.float64)
        d[k + 'keypoint_coord3d'] = coord3d
        d[k + 'kp_crop'], d[k + 'kp_uv'] = kp, uv
    np.savez_compressed(out('c1_inference.npz'), **d)

    # ---- inference2d on raw 320 x 320 frames (BASELINE config 3 shape), B = 2; weights without the lifting nets as in
    #      eval2d.py:78-79 ---------------------------------------------------------------------------------------------
    img = np.load(inp('c3_images.npy'))
    sm256, crop, scale, center = run(ColorHandPose3DNetwork, wfiles, lambda net, image: list(net.inference2d(image)),
                                     dict(image=img), exclude=['PosePrior', 'ViewpointNet'])
    np.savez_compressed(out('c3_inference2d.npz'), seed0=np.load(inp('c3_seed0.npy')), scoremap32=sm256[:, ::8, ::8, :],
                        image_crop_sub=crop[:, ::8, ::8, :], scale_crop=scale, center=center,
                        kp_crop=np.stack([general.detect_keypoints(sm256[i]) for i in range(len(sm256))]))

    # ---- the full pipeline on a BATCH of 8 (nets/ColorHandPose3DNetwork.py:61-99; utils/general.py:210 only needs B < H, W):
    #      BASELINE config 4's per-GPU shard in small -- the batch size from which the engine's F(4x4,3x3) kernel takes the trunk
    #      layers, so this file holds the headline kernel directly to the reference's code.  Optional input.
    if os.path.exists(inp('c4_images.npy')):
        img, sides = np.load(inp('c4_images.npy')), np.load(inp('c4_hand_sides.npy'))
        hand_scoremap, image_crop, scale_crop, center, kp_scoremap, coord3d, mask = run(
            ColorHandPose3DNetwork, wfiles, full, dict(image=img, hand_side=sides, evaluation=true))
        kps = np.stack([general.detect_keypoints(kp_scoremap[i]) for i in range(len(img))])
        np.savez_compressed(out('c4_b8_inference.npz'), seed0=np.load(inp('c4_seed0.npy')), shape=np.array(img.shape[1:3]), hand_side=sides,
                            hand_scoremap_sub=hand_scoremap[:, ::8, ::8, :],
                            mask_packed=np.stack([np.packbits(mask[i, :, :, 0].astype(np.uint8)) for i in range(len(img))]),
                            min_margin=knife_edge_pixels(hand_scoremap)[0], knife_iyx=knife_edge_pixels(hand_scoremap)[1],
                            center=center, scale_crop=scale_crop, image_crop_sub=image_crop[:, ::8, ::8, :],
                            scoremap32=kp_scoremap[:, ::8, ::8, :], scoremap256_sum=kp_scoremap.sum(axis=(1, 2), dtype=np.float64),
                            keypoint_coord3d=coord3d, kp_crop=kps,
                            kp_uv=np.stack([general.trafo_coords(kps[i], center[i:i + 1], scale_crop[i:i + 1], 256) for i in range(len(img))]))

    # ---- mask / bounding-box stage (utils/general.py:233-328) on engineered score maps --------------------------------
    cases = np.load(inp('mask_cases.npz'))
    d = {}

    def glue(net, scoremap):
        m = general.single_obj_scoremap(scoremap)
        c, bb, s = general.calc_center_bb(m)
        return [m, c, s]

    class _NoNet(object):
        def init(self, *a, **k):
            pass
    for case in sorted(cases.files):
        # real TF has ONE behaviour for reduce_min/max of an empty tensor (key suffix 'tf'); the stand-in is run with both
        # candidate reducer identities
        for rid in (('inf', 'fltmax') if set_empty_reduce else ('tf',)):
            if set_empty_reduce:
                set_empty_reduce(rid)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore', RuntimeWarning)
                m, c, s = run(_NoNet, [], glue, dict(scoremap=cases[case]))
            d['%s_%s_mask_packed' % (case, rid)] = np.packbits(m[0, :, :, 0].astype(np.uint8))
            d['%s_%s_center' % (case, rid)], d['%s_%s_size' % (case, rid)] = c, s
    if set_empty_reduce:
        set_empty_reduce('inf')
    np.savez_compressed(out('mask_cases.npz'), **d)

    # ---- PosePriorNetwork, all five variants (nets/PosePriorNetwork.py:59-95) -------------------------------------------
    d = {}
    sm, hs = np.load(inp('lifting_scoremaps.npy')), np.load(inp('lifting_hand_sides.npy'))
    for v in VARIANTS:
        wf = [inp('lifting-bottleneck.pickle' if v == 'bottleneck' else 'lifting.pickle')]
        rel, c3d, R = run(lambda: PosePriorNetwork(v), wf,
                          lambda net, scoremap, hand_side, evaluation: list(net.inference(scoremap, hand_side, evaluation)),
                          dict(scoremap=sm, hand_side=hs, evaluation=true))
        d[v + '_rel'], d[v + '_coord3d'] = rel, c3d
        if R is not None:
            d[v + '_R'] = R
    np.savez_compressed(out('poseprior_variants.npz'), **d)

    # ---- the metric: EvalUtil (utils/general.py:522-611) on seeded errors -------------------------------------------------
    e = np.load(inp('evalutil_feeds.npz'))
    util = general.EvalUtil()
    for gt, vis, pr in zip(e['gt'], e['vis'], e['pred']):
        util.feed(gt, vis, pr)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        mean, median, auc, pck, thr = util.get_measures(0.0, 30.0, 20)
    np.savez_compressed(out('evalutil.npz'), gt=e['gt'], vis=e['vis'], pred=e['pred'],
                        mean=mean, median=median, auc=auc, pck=pck, thresholds=thr)
    for f in sorted(os.listdir(a.out)):
        if f.startswith(a.prefix):
            print(f, os.path.getsize(os.path.join(a.out, f)))


if __name__ == '__main__':
    main()
