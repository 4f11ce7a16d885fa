"""The reference's networks restated on top of oracle.tf_ops (oracle; test
infrastructure only).  `weights` is the reference's weight-file content merged into one
dict: TF variable name -> float32 ndarray, e.g.
'HandSegNet/conv1_1/weights' [3,3,3,64] HWIO, 'PosePrior/fc_rel0/weights' [2050,512].
"""
import numpy as np

from . import tf_ops as T
from . import general as G

F32 = np.float32


class _Ops:
    """NetworkOps of utils/general.py:26-65,112-148 bound to a weight dict and a scope."""

    def __init__(self, weights, scope, acc=np.float32, taps=None, f16=False):
        self.w, self.scope, self.acc, self.taps, self.f16 = weights, scope, acc, taps, f16

    @staticmethod
    def _h(x):
        """round-trip through float16: what a half-precision tensor in HBM holds"""
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)

    def _tap(self, name, x):
        if self.taps is not None:
            self.taps['%s/%s' % (self.scope, name)] = x

    def conv(self, x, name, kernel_size, stride, out_chan):
        w = self.w['%s/%s/weights' % (self.scope, name)]
        b = self.w['%s/%s/biases' % (self.scope, name)]
        assert w.shape == (kernel_size, kernel_size, x.shape[3], out_chan), (name, w.shape, x.shape)
        if self.f16:      # engine dtype 1: half operands, float32 accumulate + bias (include/hp3d.h)
            x, w = self._h(x), self._h(w)
        y = T.bias_add(T.conv2d_same(x, w, stride, acc=self.acc), b)
        return y

    def conv_relu(self, x, name, kernel_size, stride, out_chan):
        y = T.leaky_relu(self.conv(x, name, kernel_size, stride, out_chan))
        if self.f16:
            y = self._h(y)          # activations are stored as halves; heads (conv_lin) stay float32
        self._tap(name, y)
        return y

    def conv_lin(self, x, name, kernel_size, stride, out_chan):
        y = self.conv(x, name, kernel_size, stride, out_chan)
        self._tap(name, y)
        return y

    def max_pool(self, x, name):
        y = T.max_pool_2x2(x)
        self._tap(name, y)
        return y

    def fc(self, x, name, out_chan):
        w = self.w['%s/%s/weights' % (self.scope, name)]
        b = self.w['%s/%s/biases' % (self.scope, name)]
        assert x.ndim == 2 and w.shape == (x.shape[1], out_chan), (name, w.shape, x.shape)
        y = T.fully_connected(x, w, b, acc=self.acc)
        self._tap(name, y)
        return y

    def fc_relu(self, x, name, out_chan):
        y = T.leaky_relu(self.fc(x, name, out_chan))
        self._tap(name + '/relu', y)
        return y


def handsegnet(weights, image, acc=np.float32, taps=None, f16=False):
    """ColorHandPose3DNetwork.inference_detection -- nets/ColorHandPose3DNetwork.py:131-168.
    Returns (scoremap_small [B,H/8,W/8,2], [scoremap_large [B,H,W,2]])."""
    ops = _Ops(weights, 'HandSegNet', acc, taps, f16)
    x = np.asarray(image, dtype=F32)
    for block_id, (n, c, pool) in enumerate(zip([2, 2, 4, 4], [64, 128, 256, 512], [True, True, True, False]), 1):
        for layer_id in range(n):
            x = ops.conv_relu(x, 'conv%d_%d' % (block_id, layer_id + 1), 3, 1, c)
        if pool:
            x = ops.max_pool(x, 'pool%d' % block_id)
    x = ops.conv_relu(x, 'conv5_1', 3, 1, 512)
    enc = ops.conv_relu(x, 'conv5_2', 3, 1, 128)
    x = ops.conv_relu(enc, 'conv6_1', 1, 1, 512)
    scoremap = ops.conv_lin(x, 'conv6_2', 1, 1, 2)
    H, W = image.shape[1], image.shape[2]
    return scoremap, [T.resize_bilinear_legacy(scoremap, H, W)]


def posenet2d(weights, image_crop, acc=np.float32, taps=None, num_kp=21, f16=False):
    """ColorHandPose3DNetwork.inference_pose2d -- nets/ColorHandPose3DNetwork.py:170-219.
    Returns the list of 3 scoremaps [B,h/8,w/8,21]."""
    ops = _Ops(weights, 'PoseNet2D', acc, taps, f16)
    x = np.asarray(image_crop, dtype=F32)
    for block_id, (n, c, pool) in enumerate(zip([2, 2, 4, 2], [64, 128, 256, 512], [True, True, True, False]), 1):
        for layer_id in range(n):
            x = ops.conv_relu(x, 'conv%d_%d' % (block_id, layer_id + 1), 3, 1, c)
        if pool:
            x = ops.max_pool(x, 'pool%d' % block_id)
    for name in ('conv4_3', 'conv4_4', 'conv4_5', 'conv4_6'):
        x = ops.conv_relu(x, name, 3, 1, 256)
    enc = ops.conv_relu(x, 'conv4_7', 3, 1, 128)
    x = ops.conv_relu(enc, 'conv5_1', 1, 1, 512)
    scoremaps = [ops.conv_lin(x, 'conv5_2', 1, 1, num_kp)]
    for pass_id in range(2):
        x = np.concatenate([scoremaps[-1], enc], axis=3)  # scoremap FIRST (:210)
        for rec_id in range(5):
            x = ops.conv_relu(x, 'conv%d_%d' % (pass_id + 6, rec_id + 1), 7, 1, 128)
        x = ops.conv_relu(x, 'conv%d_6' % (pass_id + 6), 1, 1, 128)
        scoremaps.append(ops.conv_lin(x, 'conv%d_7' % (pass_id + 6), 1, 1, num_kp))
    return scoremaps


def poseprior_can(weights, scoremap32, hand_side, acc=np.float32, taps=None, bottleneck=False, num_kp=21):
    """_inference_pose3d_can -- nets/ColorHandPose3DNetwork.py:249-272 (and
    PosePriorNetwork._inference_pose3d, nets/PosePriorNetwork.py:97-122 with `bottleneck`)."""
    ops = _Ops(weights, 'PosePrior', acc, taps)
    x = np.asarray(scoremap32, dtype=F32)
    B = x.shape[0]
    for i, c in enumerate([32, 64, 128]):
        x = ops.conv_relu(x, 'conv_pose_%d_1' % i, 3, 1, c)
        x = ops.conv_relu(x, 'conv_pose_%d_2' % i, 3, 2, c)
    x = x.reshape(B, -1)  # NHWC flatten (h,w,c)
    x = np.concatenate([x, np.asarray(hand_side, dtype=F32)], axis=1)
    for i, c in enumerate([512, 512]):
        x = ops.fc_relu(x, 'fc_rel%d' % i, c)  # dropout(keep=1.0) == identity when evaluating
    if bottleneck:
        x = ops.fc(x, 'fc_bottleneck', 30)
    x = ops.fc(x, 'fc_xyz', num_kp * 3)
    return x.reshape(B, num_kp, 3)


def viewpoint_uvec(weights, scoremap32, hand_side, acc=np.float32, taps=None):
    """_rotation_estimation -- nets/ColorHandPose3DNetwork.py:285-309."""
    ops = _Ops(weights, 'ViewpointNet', acc, taps)
    x = np.asarray(scoremap32, dtype=F32)
    B = x.shape[0]
    for i, c in enumerate([64, 128, 256]):
        x = ops.conv_relu(x, 'conv_vp_%d_1' % i, 3, 1, c)
        x = ops.conv_relu(x, 'conv_vp_%d_2' % i, 3, 2, c)
    x = x.reshape(B, -1)
    x = np.concatenate([x, np.asarray(hand_side, dtype=F32)], axis=1)
    for i, c in enumerate([256, 128]):
        x = ops.fc_relu(x, 'fc_vp%d' % i, c)
    ux = ops.fc(x, 'fc_vp_ux', 1)
    uy = ops.fc(x, 'fc_vp_uy', 1)
    uz = ops.fc(x, 'fc_vp_uz', 1)
    return ux, uy, uz


def get_rot_mat(ux_b, uy_b, uz_b):
    """_get_rot_mat + _stitch_mat_from_vecs -- nets/ColorHandPose3DNetwork.py:311-334,363-384.
    float32 throughout; returns [B,3,3] row-major (App. B.13)."""
    ux_b, uy_b, uz_b = (np.asarray(v, dtype=F32) for v in (ux_b, uy_b, uz_b))
    u_norm = np.sqrt(np.square(ux_b) + np.square(uy_b) + np.square(uz_b) + F32(1e-8)).astype(F32)
    theta = u_norm
    st = np.sin(theta).astype(F32)[:, 0]
    ct = np.cos(theta).astype(F32)[:, 0]
    one_ct = (F32(1.0) - np.cos(theta).astype(F32))[:, 0]
    norm_fac = (F32(1.0) / u_norm[:, 0]).astype(F32)
    ux = ux_b[:, 0] * norm_fac
    uy = uy_b[:, 0] * norm_fac
    uz = uz_b[:, 0] * norm_fac
    rows = [ct + ux * ux * one_ct, ux * uy * one_ct - uz * st, ux * uz * one_ct + uy * st,
            uy * ux * one_ct + uz * st, ct + uy * uy * one_ct, uy * uz * one_ct - ux * st,
            uz * ux * one_ct - uy * st, uz * uy * one_ct + ux * st, ct + uz * uz * one_ct]
    m = np.stack(rows, axis=0).astype(F32)           # [9, B]
    return m.reshape(3, 3, -1).transpose(2, 0, 1).copy()


def flip_right_hand(coords, hand_side):
    """_flip_right_hand -- nets/ColorHandPose3DNetwork.py:240-242,336-361: z -> -z where
    argmax(hand_side)==1."""
    right = np.argmax(np.asarray(hand_side), axis=1) == 1
    out = np.array(coords, dtype=F32, copy=True)
    out[right, :, 2] = -out[right, :, 2]
    return out


def pose3d(weights, scoremap32, hand_side, acc=np.float32, taps=None):
    """_inference_pose3d -- nets/ColorHandPose3DNetwork.py:221-247.  Returns
    (coord_xyz_rel_normed [B,21,3], coord_can [B,21,3], rot_mat [B,3,3])."""
    can = poseprior_can(weights, scoremap32, hand_side, acc, taps)
    ux, uy, uz = viewpoint_uvec(weights, scoremap32, hand_side, acc, taps)
    R = get_rot_mat(ux, uy, uz)
    flipped = flip_right_hand(can, hand_side)
    rel = np.einsum('bki,bij->bkj', flipped.astype(acc), R.astype(acc)).astype(F32)
    return rel, can, R


def inference(weights, image, hand_side, evaluation=True, acc=np.float32, taps=None, crop_size=256, f16=False):
    """ColorHandPose3DNetwork.inference -- nets/ColorHandPose3DNetwork.py:61-99.
    Returns (hand_scoremap, image_crop, scale_crop, center, keypoints_scoremap, keypoint_coord3d)."""
    assert evaluation, "the oracle restates the evaluation graph only (dropout == identity)"
    image = np.asarray(image, dtype=F32)
    _, large = handsegnet(weights, image, acc, taps, f16)
    hand_scoremap = large[-1]
    hand_mask = G.single_obj_scoremap(hand_scoremap, early_exit=True)
    center, _, crop_size_best = G.calc_center_bb(hand_mask)
    scale_crop = G.scale_from_crop_size(crop_size_best, crop_size)
    image_crop = G.crop_image_from_xy(image, center, crop_size, scale=scale_crop)
    if taps is not None:
        taps['hand_mask'] = hand_mask
    sm32 = posenet2d(weights, image_crop, acc, taps, f16=f16)[-1]
    coord3d, _, _ = pose3d(weights, sm32, hand_side, acc, taps)
    kp_scoremap = T.resize_bilinear_legacy(sm32, crop_size, crop_size)
    return hand_scoremap, image_crop, scale_crop, center, kp_scoremap, coord3d


def inference2d(weights, image, acc=np.float32, taps=None, crop_size=256):
    """ColorHandPose3DNetwork.inference2d -- nets/ColorHandPose3DNetwork.py:101-129.
    Returns (keypoints_scoremap, image_crop, scale_crop, center) -- note the order."""
    image = np.asarray(image, dtype=F32)
    _, large = handsegnet(weights, image, acc, taps)
    hand_mask = G.single_obj_scoremap(large[-1], early_exit=True)
    center, _, crop_size_best = G.calc_center_bb(hand_mask)
    scale_crop = G.scale_from_crop_size(crop_size_best, crop_size)
    image_crop = G.crop_image_from_xy(image, center, crop_size, scale=scale_crop)
    sm32 = posenet2d(weights, image_crop, acc, taps)[-1]
    return T.resize_bilinear_legacy(sm32, crop_size, crop_size), image_crop, scale_crop, center


def poseprior_network(weights, variant, scoremap256, hand_side, evaluation=True, acc=np.float32, taps=None):
    """PosePriorNetwork(variant).inference -- nets/PosePriorNetwork.py:59-95.
    Returns (coord_xyz_rel_normed, coord3d, R)."""
    assert evaluation
    pooled = T.avg_pool_8x8(np.asarray(scoremap256, dtype=F32))
    if variant == 'direct':
        c = poseprior_can(weights, pooled, hand_side, acc, taps)
        return c, c, None
    if variant == 'bottleneck':
        c = poseprior_can(weights, pooled, hand_side, acc, taps, bottleneck=True)
        return c, c, None
    if variant == 'proposed':
        rel, can, R = pose3d(weights, pooled, hand_side, acc, taps)
        return rel, can, R
    if variant in ('local', 'local_w_xyz_loss'):
        from .relative_trafo import bone_rel_trafo_inv
        loc = poseprior_can(weights, pooled, hand_side, acc, taps)
        return bone_rel_trafo_inv(loc), loc, None
    assert 0, "Unknown variant."
