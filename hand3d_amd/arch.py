"""Layer tables of the ColorHandPose3D networks (variable names, shapes, layouts).

This is the key space of the reference's weight pickles: every entry
is `<scope>/<layer>/weights` (+ `/biases`).  Conv weights are HWIO [k,k,Cin,Cout]
(utils/general.py:41-50 of the reference), FC weights [in,out] (:117-126).
Layer lists follow nets/ColorHandPose3DNetwork.py:144-161 (HandSegNet), :183-214
(PoseNet2D), :255-267 (PosePrior), :291-307 (ViewpointNet) and
nets/PosePriorNetwork.py:115-116 (fc_bottleneck).
"""
from collections import namedtuple

Conv = namedtuple('Conv', 'scope name k cin cout stride relu')
FC = namedtuple('FC', 'scope name cin cout relu')


def handsegnet_layers():
    L, cin = [], 3
    for block_id, (n, c) in enumerate(zip([2, 2, 4, 4], [64, 128, 256, 512]), 1):
        for layer_id in range(n):
            L.append(Conv('HandSegNet', 'conv%d_%d' % (block_id, layer_id + 1), 3, cin, c, 1, True))
            cin = c
    L.append(Conv('HandSegNet', 'conv5_1', 3, 512, 512, 1, True))
    L.append(Conv('HandSegNet', 'conv5_2', 3, 512, 128, 1, True))
    L.append(Conv('HandSegNet', 'conv6_1', 1, 128, 512, 1, True))
    L.append(Conv('HandSegNet', 'conv6_2', 1, 512, 2, 1, False))
    return L


def posenet2d_layers(num_kp=21):
    L, cin = [], 3
    for block_id, (n, c) in enumerate(zip([2, 2, 4, 2], [64, 128, 256, 512]), 1):
        for layer_id in range(n):
            L.append(Conv('PoseNet2D', 'conv%d_%d' % (block_id, layer_id + 1), 3, cin, c, 1, True))
            cin = c
    for name, ci, co in (('conv4_3', 512, 256), ('conv4_4', 256, 256), ('conv4_5', 256, 256),
                         ('conv4_6', 256, 256), ('conv4_7', 256, 128)):
        L.append(Conv('PoseNet2D', name, 3, ci, co, 1, True))
    L.append(Conv('PoseNet2D', 'conv5_1', 1, 128, 512, 1, True))
    L.append(Conv('PoseNet2D', 'conv5_2', 1, 512, num_kp, 1, False))
    for p in (6, 7):
        cin = num_kp + 128
        for r in range(1, 6):
            L.append(Conv('PoseNet2D', 'conv%d_%d' % (p, r), 7, cin, 128, 1, True))
            cin = 128
        L.append(Conv('PoseNet2D', 'conv%d_6' % p, 1, 128, 128, 1, True))
        L.append(Conv('PoseNet2D', 'conv%d_7' % p, 1, 128, num_kp, 1, False))
    return L


def poseprior_layers(num_kp=21, bottleneck=False):
    L, cin = [], num_kp
    for i, c in enumerate([32, 64, 128]):
        L.append(Conv('PosePrior', 'conv_pose_%d_1' % i, 3, cin, c, 1, True))
        L.append(Conv('PosePrior', 'conv_pose_%d_2' % i, 3, c, c, 2, True))
        cin = c
    L.append(FC('PosePrior', 'fc_rel0', 4 * 4 * 128 + 2, 512, True))
    L.append(FC('PosePrior', 'fc_rel1', 512, 512, True))
    if bottleneck:
        L.append(FC('PosePrior', 'fc_bottleneck', 512, 30, False))
        L.append(FC('PosePrior', 'fc_xyz', 30, num_kp * 3, False))
    else:
        L.append(FC('PosePrior', 'fc_xyz', 512, num_kp * 3, False))
    return L


def viewpoint_layers(num_kp=21):
    L, cin = [], num_kp
    for i, c in enumerate([64, 128, 256]):
        L.append(Conv('ViewpointNet', 'conv_vp_%d_1' % i, 3, cin, c, 1, True))
        L.append(Conv('ViewpointNet', 'conv_vp_%d_2' % i, 3, c, c, 2, True))
        cin = c
    L.append(FC('ViewpointNet', 'fc_vp0', 4 * 4 * 256 + 2, 256, True))
    L.append(FC('ViewpointNet', 'fc_vp1', 256, 128, True))
    for a in ('ux', 'uy', 'uz'):
        L.append(FC('ViewpointNet', 'fc_vp_%s' % a, 128, 1, False))
    return L


def all_layers(bottleneck=False):
    return handsegnet_layers() + posenet2d_layers() + poseprior_layers(bottleneck=bottleneck) + viewpoint_layers()


def var_shapes(layers):
    """{tf variable name: shape} for a layer list."""
    out = {}
    for l in layers:
        base = '%s/%s' % (l.scope, l.name)
        if isinstance(l, Conv):
            out[base + '/weights'] = (l.k, l.k, l.cin, l.cout)
        else:
            out[base + '/weights'] = (l.cin, l.cout)
        out[base + '/biases'] = (l.cout,)
    return out


def conv_flops(l, ho, wo):
    """2*MAC of one conv layer at output size ho x wo (SURVEY.md App. A convention)."""
    return 2.0 * l.k * l.k * l.cin * l.cout * ho * wo


def pipeline_flops(H, W):
    """Algorithmic FLOPs per image of the full pipeline at input H x W:
    returns dict(handsegnet=..., posenet=..., lifting=..., total=...)."""
    def stack(layers, h, w, pools):
        f = 0.0
        for l in layers:
            if isinstance(l, FC):
                f += 2.0 * l.cin * l.cout
                continue
            if l.stride == 2:
                h, w = (h + 1) // 2, (w + 1) // 2
            f += conv_flops(l, h, w)
            if l.name in pools:
                h, w = h // 2, w // 2
        return f
    pools = ('conv1_2', 'conv2_2', 'conv3_4')
    seg = stack(handsegnet_layers(), H, W, pools)
    pose = stack(posenet2d_layers(), 256, 256, pools)
    lift = stack(poseprior_layers(), 32, 32, ()) + stack(viewpoint_layers(), 32, 32, ())
    return dict(handsegnet=seg, posenet=pose, lifting=lift, total=seg + pose + lift)
