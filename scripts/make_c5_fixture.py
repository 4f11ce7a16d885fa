"""BASELINE config 5 at its OWN size and precision, from the oracle: tests/golden/c5_f16_480x640.npz.

Config 5 = 640x480 RGB frames, half-precision HandSegNet / PoseNet2D trunks (float32 accumulation, heads, mask stage and
lifting nets; include/hp3d.h `hp3d_finalize_weights(ctx, 1)`).  The oracle needs minutes per image at this size, so it is
run ONCE here on the CPU box (float64 accumulation, rounding to half exactly where the engine stores halves:
oracle/nets.py `f16=True`) and the GPU test only reads the fixture.  Layer lists followed:
nets/ColorHandPose3DNetwork.py:144-161,183-214 (via oracle/nets.py).

What is kept per image (2 images of synth.make_batch(1000, 2, 480, 640), alternating hand sides):
  * `seg_small`   [2,60,80,2]  HandSegNet logits (conv6_2), f16-rounding oracle; `seg_small_f32`: the float32 oracle's
  * `det`, `mask` packed bits   round(softmax)[...,1] and the grown hand mask of the f16 oracle (utils/general.py:233-268)
  * `margin_q`    [2,480,640] uint8   min(|logit1 - logit0| of the up-sampled map, 0.0255) * 1e4: lets the GPU test say
                  WHERE a det pixel may legitimately differ (half-precision trunks move the logits by ~1e-3)
  * `center`, `scale_crop`, `seed`
  * `sm32`        [3][2,32,32,21] the three PoseNet2D score maps on the oracle's own crop; `sm32_f32`: float32 oracle's last
  * `coord3d`     [2,21,3]; `coord3d_f32`

    python scripts/make_c5_fixture.py            (~10-20 min on 8 cores)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hand3d_amd import synth  # noqa: E402
from oracle import general as G  # noqa: E402
from oracle import nets as N  # noqa: E402
from oracle import tf_ops as T  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'c5_f16_480x640.npz')
SEED0, NIMG, H, W = 1000, 2, 480, 640


def one_pass(w, img, hs, f16):
    """oracle.nets.inference with the intermediate results the fixture keeps (same calls, same order)."""
    small, large = N.handsegnet(w, img, acc=np.float64, f16=f16)
    hand_scoremap = large[-1]
    fg, det = G.fg_and_detmap(hand_scoremap)
    seed = G.find_max_location(fg)
    mask = G.single_obj_scoremap(hand_scoremap, early_exit=True)
    center, _, best = G.calc_center_bb(mask)
    scale = G.scale_from_crop_size(best, 256)
    crop = G.crop_image_from_xy(img, center, 256, scale=scale)
    sms = N.posenet2d(w, crop, acc=np.float64, f16=f16)
    coord3d, _, _ = N.pose3d(w, sms[-1], hs, acc=np.float64)
    return dict(small=small, large=hand_scoremap, det=det, mask=mask, seed=seed, center=center, scale=scale, sms=sms,
                coord3d=coord3d)


def main():
    w = synth.make_weights(seed=42)
    img = synth.make_batch(SEED0, NIMG, H, W)
    hs = synth.hand_sides(NIMG)
    t0 = time.time()
    r16 = [one_pass(w, img[i:i + 1], hs[i:i + 1], True) for i in range(NIMG)]
    print('f16-rounding oracle: %.0f s' % (time.time() - t0), flush=True)
    t0 = time.time()
    r32 = [one_pass(w, img[i:i + 1], hs[i:i + 1], False) for i in range(NIMG)]
    print('float32 oracle: %.0f s' % (time.time() - t0), flush=True)

    def cat(rs, key):
        return np.concatenate([r[key] for r in rs], 0)
    large = cat(r16, 'large')
    margin = np.abs(large[..., 1] - large[..., 0])
    out = dict(
        seed0=np.int32(SEED0), seg_small=cat(r16, 'small'), seg_small_f32=cat(r32, 'small'),
        det=np.packbits(cat(r16, 'det').reshape(NIMG, -1).astype(np.uint8), axis=1),
        mask=np.packbits(cat(r16, 'mask').reshape(NIMG, -1).astype(np.uint8), axis=1),
        mask_f32=np.packbits(cat(r32, 'mask').reshape(NIMG, -1).astype(np.uint8), axis=1),
        margin_q=np.minimum(np.floor(margin * 1e4), 255).astype(np.uint8),
        center=cat(r16, 'center'), scale_crop=cat(r16, 'scale'), seed=cat(r16, 'seed'),
        center_f32=cat(r32, 'center'), scale_crop_f32=cat(r32, 'scale'),
        sm32=np.stack([np.concatenate([r['sms'][k] for r in r16], 0) for k in range(3)], 0),
        sm32_f32=np.concatenate([r['sms'][2] for r in r32], 0),
        coord3d=cat(r16, 'coord3d'), coord3d_f32=cat(r32, 'coord3d'))
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), 'bytes')
    for k, v in out.items():
        print('  %-16s %s %s' % (k, v.dtype, v.shape))
    print('f16 vs f32 oracle: logits %.2e  sm32 %.2e  coord3d %.2e;  masks equal: %s;  centres %s / %s' % (
        np.abs(out['seg_small'] - out['seg_small_f32']).max(), np.abs(out['sm32'][2] - out['sm32_f32']).max(),
        np.abs(out['coord3d'] - out['coord3d_f32']).max(), np.array_equal(out['mask'], out['mask_f32']),
        out['center'].tolist(), out['center_f32'].tolist()))


if __name__ == '__main__':
    main()
