"""The hot path as a BATCHED torch-CPU (oneDNN) program -- the "CPU path timed beside" the GPU engine (bench.py's
`cpu_baseline`, kind "port"; SURVEY.md 8d: TensorFlow 1.3 cannot run here, so the baseline is a CPU restatement).

Test / measurement infrastructure, like everything under oracle/: only tests/ and bench.py's cpu_baseline leg import it.  It
follows the same reference lines as oracle/nets.py and oracle/general.py (cited per function) but is written for SPEED on many
cores: NCHW tensors stay in torch across layers, the whole batch goes through every layer in one call, and the glue (legacy
bilinear resize as two small matrix products, mask growth as separable max-pools over the whole batch, crop as one
grid_sample) is vectorised -- oracle/nets.py calls NumPy glue per image (0.4 s/image at 320x320) and copies NHWC <-> NCHW per
layer, which under-feeds a big host by an order of magnitude.  tests/test_oracle_nets_torch.py holds it to the oracle.
"""
import numpy as np
import torch
import torch.nn.functional as Fn


def _same_pads(n, k, s):
    """TF 'SAME' (SURVEY.md App. B.1): the remainder of the padding goes AFTER."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


class TorchPort(object):
    """weights: the reference's variable dictionary (TF name -> HWIO / [in,out] float32 arrays)."""

    def __init__(self, weights):
        self.w, self.b = {}, {}
        for name, v in weights.items():
            layer, kind = name.rsplit('/', 1)
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
            if kind == 'weights':
                self.w[layer] = t.permute(3, 2, 0, 1).contiguous() if t.ndim == 4 else t      # HWIO -> OIHW
            else:
                self.b[layer] = t
        self._rs = {}

    # ---- NetworkOps (utils/general.py:26-65,112-136) -------------------------------------------------------------------
    def conv(self, x, layer, stride=1, relu=True):
        w = self.w[layer]
        k = w.shape[2]
        pt, pb = _same_pads(x.shape[2], k, stride)
        pl, pr = _same_pads(x.shape[3], k, stride)
        if pt == pb and pl == pr:
            y = Fn.conv2d(x, w, self.b[layer], stride=stride, padding=(pt, pl))
        else:
            y = Fn.conv2d(Fn.pad(x, (pl, pr, pt, pb)), w, self.b[layer], stride=stride)
        return torch.maximum(y, 0.01 * y) if relu else y

    def fc(self, x, layer, relu):
        y = torch.addmm(self.b[layer], x, self.w[layer])
        return torch.maximum(y, 0.01 * y) if relu else y

    def resize_legacy(self, x, oh, ow):
        """tf.image.resize_images (bilinear, align_corners=False, no half-pixel offset: src = dst * in/out) as R_y x R_x^T."""
        def mat(n_in, n_out):
            key = (n_in, n_out)
            if key not in self._rs:
                src = np.arange(n_out, dtype=np.float32) * np.float32(n_in / n_out)
                lo = np.floor(src).astype(np.int64)
                hi = np.minimum(lo + 1, n_in - 1)
                fr = (src - lo).astype(np.float32)
                m = np.zeros((n_out, n_in), np.float32)
                np.add.at(m, (np.arange(n_out), lo), 1.0 - fr)
                np.add.at(m, (np.arange(n_out), hi), fr)
                self._rs[key] = torch.from_numpy(m)
            return self._rs[key]
        return torch.matmul(torch.matmul(mat(x.shape[2], oh), x), mat(x.shape[3], ow).t())

    # ---- the networks (nets/ColorHandPose3DNetwork.py:131-168, 170-219, 249-309) ----------------------------------------
    def trunk(self, x, scope, n4):
        for blk, (n, pool) in enumerate(zip([2, 2, 4, n4], [True, True, True, False]), 1):
            for i in range(n):
                x = self.conv(x, '%s/conv%d_%d' % (scope, blk, i + 1))
            if pool:
                x = Fn.max_pool2d(x, 2)
        return x

    def handsegnet(self, x):
        x = self.trunk(x, 'HandSegNet', 4)
        x = self.conv(self.conv(x, 'HandSegNet/conv5_1'), 'HandSegNet/conv5_2')
        return self.conv(self.conv(x, 'HandSegNet/conv6_1'), 'HandSegNet/conv6_2', relu=False)

    def posenet2d(self, x):
        x = self.trunk(x, 'PoseNet2D', 2)
        for nm in ('conv4_3', 'conv4_4', 'conv4_5', 'conv4_6'):
            x = self.conv(x, 'PoseNet2D/' + nm)
        enc = self.conv(x, 'PoseNet2D/conv4_7')
        sm = self.conv(self.conv(enc, 'PoseNet2D/conv5_1'), 'PoseNet2D/conv5_2', relu=False)
        for p in (6, 7):
            x = torch.cat([sm, enc], 1)                      # score map FIRST (:210)
            for r in range(1, 7):
                x = self.conv(x, 'PoseNet2D/conv%d_%d' % (p, r))
            sm = self.conv(x, 'PoseNet2D/conv%d_7' % p, relu=False)
        return sm

    def lift(self, sm, hs):
        def tower(scope, fmt):
            x = sm
            for i in range(3):
                x = self.conv(self.conv(x, fmt % (scope, i, 1)), fmt % (scope, i, 2), stride=2)
            return torch.cat([x.permute(0, 2, 3, 1).reshape(x.shape[0], -1), hs], 1)       # flatten (h, w, c)
        x = tower('PosePrior', '%s/conv_pose_%d_%d')
        can = self.fc(self.fc(self.fc(x, 'PosePrior/fc_rel0', True), 'PosePrior/fc_rel1', True), 'PosePrior/fc_xyz', False).reshape(-1, 21, 3)
        x = tower('ViewpointNet', '%s/conv_vp_%d_%d')
        x = self.fc(self.fc(x, 'ViewpointNet/fc_vp0', True), 'ViewpointNet/fc_vp1', True)
        u = torch.cat([self.fc(x, 'ViewpointNet/fc_vp_u' + a, False) for a in 'xyz'], 1)
        # _get_rot_mat (:311-334)
        th = torch.sqrt((u * u).sum(1) + 1e-8)
        st, ct = torch.sin(th), torch.cos(th)
        oc = 1.0 - ct
        ux, uy, uz = (u / th[:, None]).unbind(1)
        R = torch.stack([ct + ux * ux * oc, ux * uy * oc - uz * st, ux * uz * oc + uy * st,
                         uy * ux * oc + uz * st, ct + uy * uy * oc, uy * uz * oc - ux * st,
                         uz * ux * oc - uy * st, uz * uy * oc + ux * st, ct + uz * uz * oc], 1).reshape(-1, 3, 3)
        right = hs.argmax(1) == 1                         # _flip_right_hand (:336-361)
        flip = can.clone()
        flip[right, :, 2] = -flip[right, :, 2]
        return torch.bmm(flip, R)

    # ---- glue (utils/general.py:163-328) ---------------------------------------------------------------------------------
    def mask_center_scale(self, logits_large):
        B, _, H, W = logits_large.shape
        fg = torch.softmax(logits_large, 1)[:, 1]
        det = torch.round(fg)
        seed = fg.reshape(B, -1).argmax(1)
        obj = torch.zeros(B, H * W)
        obj[torch.arange(B), seed] = 1.0
        obj = obj.reshape(B, 1, H, W)
        det = det[:, None]
        for _ in range(max(H, W) // 10):                  # O_{j+1} = det AND dilate_21x21(O_j); a fix-point may stop early
            nxt = det * Fn.max_pool2d(Fn.max_pool2d(obj, (1, 21), 1, (0, 10)), (21, 1), 1, (10, 0))
            if torch.equal(nxt, obj):
                break
            obj = nxt
        m = obj[:, 0] > 0.5
        rows, cols = m.any(2), m.any(1)
        ar, ac = torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32)
        inf = torch.tensor(float('inf'))
        rmin = torch.where(rows, ar, inf).amin(1); rmax = torch.where(rows, ar, -inf).amax(1)
        cmin = torch.where(cols, ac, inf).amin(1); cmax = torch.where(cols, ac, -inf).amax(1)
        center = torch.stack([0.5 * (rmax + rmin), 0.5 * (cmax + cmin)], 1)
        center = torch.where(torch.isfinite(center).all(1, keepdim=True), center, torch.full_like(center, 160.0))
        size = torch.maximum(rmax - rmin, cmax - cmin)
        size = torch.where(torch.isfinite(size), size, torch.full_like(size, 100.0))
        scale = torch.clamp(256.0 / (size * 1.25), 0.25, 5.0)
        return m, center, scale

    def crop(self, img, center, scale, crop=256):
        """crop_image_from_xy (:163-196) + tf.image.crop_and_resize ((H-1) mapping, bilinear, 0 outside the image)."""
        B, _, H, W = img.shape
        cs = crop / scale
        half = torch.floor(cs / 2.0)
        y1 = (center[:, 0] - half) / H; y2 = y1 + cs / H
        x1 = (center[:, 1] - half) / W; x2 = x1 + cs / W
        t = torch.arange(crop, dtype=torch.float32) / (crop - 1)
        iy = (y1[:, None] + t[None] * (y2 - y1)[:, None]) * (H - 1)             # [B, crop] source rows
        ix = (x1[:, None] + t[None] * (x2 - x1)[:, None]) * (W - 1)
        ok = ((iy >= 0) & (iy <= H - 1))[:, :, None] & ((ix >= 0) & (ix <= W - 1))[:, None, :]
        grid = torch.stack([(ix * (2.0 / (W - 1)) - 1.0)[:, None, :].expand(B, crop, crop),
                            (iy * (2.0 / (H - 1)) - 1.0)[:, :, None].expand(B, crop, crop)], 3)
        out = Fn.grid_sample(img, grid, mode='bilinear', padding_mode='border', align_corners=True)
        return out * ok[:, None].to(out.dtype)

    # ---- ColorHandPose3DNetwork.inference (:61-99) + detect_keypoints (utils/general.py:331-344) ------------------------
    def inference(self, image_nhwc, hand_side):
        with torch.no_grad():
            img = torch.from_numpy(np.ascontiguousarray(image_nhwc, dtype=np.float32)).permute(0, 3, 1, 2).contiguous()
            hs = torch.from_numpy(np.ascontiguousarray(hand_side, dtype=np.float32))
            H, W = img.shape[2], img.shape[3]
            hand_scoremap = self.resize_legacy(self.handsegnet(img), H, W)
            _, center, scale = self.mask_center_scale(hand_scoremap)
            image_crop = self.crop(img, center, scale)
            sm32 = self.posenet2d(image_crop)
            coord3d = self.lift(sm32, hs)
            kpmap = self.resize_legacy(sm32, 256, 256)
            kp = kpmap.reshape(kpmap.shape[0], 21, -1).argmax(2)
            kp_crop = torch.stack([kp // 256, kp % 256], 2)
            return dict(hand_scoremap=hand_scoremap.permute(0, 2, 3, 1).numpy(), image_crop=image_crop.permute(0, 2, 3, 1).numpy(),
                        scale_crop=scale[:, None].numpy(), center=center.numpy(), sm32=sm32.permute(0, 2, 3, 1).numpy(),
                        keypoints_scoremap=kpmap, keypoint_coord3d=coord3d.numpy(), kp_crop=kp_crop.numpy())

    def pose2d(self, crop_nhwc):
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(crop_nhwc, dtype=np.float32)).permute(0, 3, 1, 2).contiguous()
            return self.posenet2d(x).permute(0, 2, 3, 1).numpy()
