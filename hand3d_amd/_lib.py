"""ctypes binding of libhp3d.so (include/hp3d.h) -- the only bridge between the Python call
surface and the HIP engine.  No torch, no numpy fallback: if the library is missing or no GPU is
visible, the product path raises.

HP3D_LIB=<path> overrides the library (the CPU test-suite points it at tests/emu/libhp3d_emu.so,
an interpreter of the same kernel sources; see tests/emu/hp3d_emu.h).
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, 'libhp3d.so')

VARIANTS = {'direct': 0, 'bottleneck': 1, 'proposed': 2, 'local': 3, 'local_w_xyz_loss': 3}
NET_SEG, NET_POSE, NET_PRIOR, NET_VP, NET_BOTTLENECK = 1, 2, 4, 8, 16

_f = C.POINTER(C.c_float)
_i32 = C.POINTER(C.c_int32)
_ctx = C.c_void_p

_SIGNATURES = {
    'hp3d_abi_version': (C.c_int, []),
    'hp3d_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'hp3d_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'hp3d_create': (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    'hp3d_destroy': (C.c_int, [_ctx]),
    'hp3d_last_error': (C.c_char_p, [_ctx]),
    'hp3d_stream': (C.c_void_p, [_ctx]),
    'hp3d_sync': (C.c_int, [_ctx]),
    'hp3d_set_option': (C.c_int, [_ctx, C.c_char_p, C.c_char_p]),
    'hp3d_set_weight': (C.c_int, [_ctx, C.c_char_p, _f, C.POINTER(C.c_int64), C.c_int]),
    'hp3d_finalize_weights': (C.c_int, [_ctx, C.c_int]),
    'hp3d_weights_blob_bytes': (C.c_int, [_ctx, C.POINTER(C.c_size_t)]),
    'hp3d_weights_blob_export': (C.c_int, [_ctx, C.c_void_p]),
    'hp3d_weights_blob_import': (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    'hp3d_nets_mask': (C.c_int, [_ctx]),
    'hp3d_infer_full': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9),
    'hp3d_infer_full_dev': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9),
    'hp3d_infer_full_kp': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 11),
    'hp3d_infer_full_kp_dev': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 11),
    'hp3d_infer_full_kp_u8': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10),
    'hp3d_infer_full_u8': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8),
    'hp3d_preprocess_u8': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_infer_2d': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    'hp3d_infer_2d_kp': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7),
    'hp3d_detect_keypoints': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_handsegnet': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3),
    'hp3d_posenet2d': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4),
    'hp3d_posenet2d_dev': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4),
    'hp3d_poseprior': (C.c_int, [_ctx, C.c_int, C.c_int] + [C.c_void_p] * 5),
    'hp3d_pose3d': (C.c_int, [_ctx, C.c_int] + [C.c_void_p] * 5),
    'hp3d_conv2d': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_maxpool2': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_avgpool8': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_resize_bilinear': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_crop_and_resize': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_void_p]),
    'hp3d_mask_from_scoremap': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    'hp3d_fc': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_argmax2d': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_dev_alloc': (C.c_int, [_ctx, C.c_size_t, C.POINTER(C.c_void_p)]),
    'hp3d_dev_free': (C.c_int, [_ctx, C.c_void_p]),
    'hp3d_host_alloc': (C.c_int, [_ctx, C.c_size_t, C.POINTER(C.c_void_p)]),
    'hp3d_host_free': (C.c_int, [_ctx, C.c_void_p]),
    'hp3d_memcpy': (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    'hp3d_upload_async': (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
    'hp3d_wait_upload': (C.c_int, [_ctx]),
    'hp3d_get_counter': (C.c_int, [_ctx, C.c_char_p, C.POINTER(C.c_longlong)]),
    'hp3d_set_profiling': (C.c_int, [_ctx, C.c_int]),
    'hp3d_prof_count': (C.c_int, [_ctx]),
    'hp3d_prof_get': (C.c_int, [_ctx, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_float),
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'hp3d_get_timing': (C.c_int, [_ctx, C.POINTER(C.c_float), C.c_int]),
    'hp3d_comm_unique_id': (C.c_int, [C.c_void_p]),
    'hp3d_comm_init': (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    'hp3d_bcast_weights': (C.c_int, [_ctx, C.c_int]),
    'hp3d_allgather': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p]),
    'hp3d_allgather_dev': (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p]),
    'hp3d_comm_destroy': (C.c_int, [_ctx]),
    'hp3d_crc32c': (C.c_uint32, [C.c_void_p, C.c_size_t]),
}
COMM_ID_BYTES = 128
TIMING_STAGES = ('HandSegNet', 'mask_crop', 'PoseNet2D', 'lifting', 'total')
EXPORTS = tuple(sorted(_SIGNATURES))

_lib = None
_lib_path = None
_loaded = {}


class Hp3dError(RuntimeError):
    pass


def lib_path():
    return os.environ.get('HP3D_LIB', DEFAULT_LIB)


def load(path=None):
    """dlopen the engine and attach prototypes.  Raises if the library is absent: there is no
    Python/NumPy fallback for the product path."""
    global _lib, _lib_path
    path = os.path.abspath(path or lib_path())
    if path in _loaded:
        return _loaded[path]
    if not os.path.exists(path):
        raise Hp3dError("HIP engine library not found at %s -- build it with `python -m hand3d_amd.build` "
                        "(hipcc, gfx950). The product has no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if include/hp3d.h and the library disagree
        fn.restype = res
        fn.argtypes = args
    _loaded[path] = lib
    _lib, _lib_path = lib, path
    return lib


def device_count(path=None):
    """HIP devices visible to this process (hp3d_device_count); 0 when the runtime reports an error (no GPU, no driver)."""
    n = C.c_int(0)
    rc = load(path).hp3d_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0


def device_pci_bus_id(device, path=None):
    """PCI address 'dddd:bb:dd.f' of HIP device `device` (hp3d_device_pci_bus_id); raises Hp3dError when the runtime has none."""
    buf = C.create_string_buffer(32)
    rc = load(path).hp3d_device_pci_bus_id(int(device), buf, 32)
    if rc != 0:
        raise Hp3dError('hp3d_device_pci_bus_id(%d) failed: %d' % (device, rc))
    return buf.value.decode()


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class DevBuf(object):
    """A device allocation owned through the C ABI (hp3d_dev_alloc / hp3d_dev_free); int(buf) is the device address."""

    def __init__(self, engine, ptr, nbytes):
        self.engine, self.ptr, self.nbytes = engine, ptr, nbytes
        if engine is not None and hasattr(engine, '_devbufs'):
            engine._devbufs.add(self)         # Engine.close() releases whatever is still alive (ptr becomes 0)

    def __int__(self):
        return self.ptr

    def __index__(self):
        return self.ptr

    def at(self, offset_bytes):
        return self.ptr + int(offset_bytes)

    def free(self):
        if self.ptr and self.engine is not None and self.engine.h:
            self.engine.lib.hp3d_dev_free(self.engine.h, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine(object):
    """One hp3d_ctx: a device, a stream, packed weights and a pre-allocated arena."""

    def __init__(self, device=0, path=None):
        self.lib = load(path)
        h = _ctx()
        rc = self.lib.hp3d_create(int(device), C.byref(h))
        if rc != 0:
            raise Hp3dError("hp3d_create(%d) failed: %s" % (device, self.lib.hp3d_last_error(None).decode()))
        self.h = h
        self.device = device
        self._pinned = []
        self._devbufs = weakref.WeakSet()

    def close(self):
        """Destroys the context.  Device buffers handed out by dev_alloc / to_device that are still alive are freed here
        (their `ptr` becomes 0), and so are the page-locked buffers behind pinned_empty(): arrays returned by pinned_empty()
        must not be touched after close() -- hp3d_host_free waits for pending uploads first."""
        if getattr(self, 'h', None):
            for b in list(getattr(self, '_devbufs', [])):
                b.free()
            for p in getattr(self, '_pinned', []):
                self.lib.hp3d_host_free(self.h, C.c_void_p(p))
            self._pinned = []
            self.lib.hp3d_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.hp3d_last_error(self.h).decode()
            if rc == -1:
                raise AssertionError(msg)            # the reference's bare asserts
            if rc == -4:
                raise NotImplementedError(msg)
            raise Hp3dError("hp3d error %d: %s" % (rc, msg))

    # -- options / weights -------------------------------------------------------------------
    def set_option(self, key, value):
        self._chk(self.lib.hp3d_set_option(self.h, key.encode(), value.encode()))

    def set_weight(self, name, array):
        a = _f32(array)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        self._chk(self.lib.hp3d_set_weight(self.h, name.encode(), a.ctypes.data_as(_f), shape, a.ndim))

    def load_weight_dict(self, weight_dict):
        for k in sorted(weight_dict):
            self.set_weight(k, weight_dict[k])

    def finalize_weights(self, dtype=0):
        """dtype 0 = float32; 1 (or 'f16') = half-precision HandSegNet / PoseNet2D trunks (BASELINE config 5)."""
        dtype = {'f32': 0, 'f16': 1}.get(dtype, dtype)
        self._chk(self.lib.hp3d_finalize_weights(self.h, int(dtype)))

    def nets_mask(self):
        return self.lib.hp3d_nets_mask(self.h)

    def blob_bytes(self):
        n = C.c_size_t()
        self._chk(self.lib.hp3d_weights_blob_bytes(self.h, C.byref(n)))
        return n.value

    def blob_export(self, dev_ptr):
        self._chk(self.lib.hp3d_weights_blob_export(self.h, C.c_void_p(dev_ptr)))

    def blob_import(self, dev_ptr, nets_mask):
        self._chk(self.lib.hp3d_weights_blob_import(self.h, C.c_void_p(dev_ptr), int(nets_mask)))

    def sync(self):
        self._chk(self.lib.hp3d_sync(self.h))

    def stream(self):
        return self.lib.hp3d_stream(self.h)

    # -- whole-path --------------------------------------------------------------------------
    def infer_full(self, image, hand_side, want_mask=False, outputs=('scoremap', 'crop', 'scale', 'center', 'kpmap', 'coord3d')):
        image, hand_side = _f32(image), _f32(hand_side)
        assert image.ndim == 4 and image.shape[3] == 3, "image must be [B,H,W,3]"
        B, H, W, _ = image.shape
        assert hand_side.shape == (B, 2), "hand_side must be [B,2]"
        o = {
            'scoremap': np.empty((B, H, W, 2), np.float32) if 'scoremap' in outputs else None,
            'crop': np.empty((B, 256, 256, 3), np.float32) if 'crop' in outputs else None,
            'scale': np.empty((B, 1), np.float32) if 'scale' in outputs else None,
            'center': np.empty((B, 2), np.float32) if 'center' in outputs else None,
            'kpmap': np.empty((B, 256, 256, 21), np.float32) if 'kpmap' in outputs else None,
            'coord3d': np.empty((B, 21, 3), np.float32) if 'coord3d' in outputs else None,
            'mask': np.empty((B, H, W), np.float32) if want_mask else None,
            # detect_keypoints / trafo_coords on the device (utils/general.py:331-357): int32 (row, col) in the crop and
            # float64 (row, col) in the image; need no 'kpmap'
            'kp_crop': np.empty((B, 21, 2), np.int32) if 'kp_crop' in outputs else None,
            'kp_hw': np.empty((B, 21, 2), np.float64) if 'kp_hw' in outputs else None,
        }
        self._chk(self.lib.hp3d_infer_full_kp(self.h, B, H, W, _ptr(image), _ptr(hand_side), _ptr(o['scoremap']),
                                              _ptr(o['crop']), _ptr(o['scale']), _ptr(o['center']), _ptr(o['kpmap']),
                                              _ptr(o['coord3d']), _ptr(o['mask']), _ptr(o['kp_crop']), _ptr(o['kp_hw'])))
        return o

    def infer_full_u8(self, image_u8, hand_side, H=240, W=320, want_mask=False):
        """uint8 frames [B,Hin,Win,3] -> normalise + resize on device -> full pipeline."""
        img = np.ascontiguousarray(image_u8, dtype=np.uint8)
        hand_side = _f32(hand_side)
        assert img.ndim == 4 and img.shape[3] == 3, "image must be [B,Hin,Win,3] uint8"
        B, Hin, Win, _ = img.shape
        assert hand_side.shape == (B, 2), "hand_side must be [B,2]"
        o = {'scoremap': np.empty((B, H, W, 2), np.float32), 'crop': np.empty((B, 256, 256, 3), np.float32),
             'scale': np.empty((B, 1), np.float32), 'center': np.empty((B, 2), np.float32),
             'kpmap': np.empty((B, 256, 256, 21), np.float32), 'coord3d': np.empty((B, 21, 3), np.float32),
             'mask': np.empty((B, H, W), np.float32) if want_mask else None}
        self._chk(self.lib.hp3d_infer_full_u8(self.h, B, Hin, Win, _ptr(img), H, W, _ptr(hand_side), _ptr(o['scoremap']),
                                              _ptr(o['crop']), _ptr(o['scale']), _ptr(o['center']), _ptr(o['kpmap']),
                                              _ptr(o['coord3d']), _ptr(o['mask'])))
        return o

    def preprocess_u8(self, image_u8, H, W):
        img = np.ascontiguousarray(image_u8, dtype=np.uint8)
        B, Hin, Win, _ = img.shape
        out = np.empty((B, H, W, 3), np.float32)
        self._chk(self.lib.hp3d_preprocess_u8(self.h, _ptr(img), B, Hin, Win, H, W, _ptr(out)))
        return out

    def infer_full_dev(self, B, H, W, image_ptr, hand_side_ptr, scoremap=0, crop=0, scale=0, center=0, kpmap=0,
                       coord3d=0, mask=0, kp_crop=0, kp_hw=0):
        """Device-pointer variant (ints); stream-ordered, call sync() before reading.  kp_crop (int32 [B,21,2]) /
        kp_hw (float64 [B,21,2]): detect_keypoints / trafo_coords evaluated on the device."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        if kp_crop or kp_hw:
            self._chk(self.lib.hp3d_infer_full_kp_dev(self.h, B, H, W, v(image_ptr), v(hand_side_ptr), v(scoremap), v(crop),
                                                      v(scale), v(center), v(kpmap), v(coord3d), v(mask), v(kp_crop), v(kp_hw)))
        else:
            self._chk(self.lib.hp3d_infer_full_dev(self.h, B, H, W, v(image_ptr), v(hand_side_ptr), v(scoremap), v(crop),
                                                   v(scale), v(center), v(kpmap), v(coord3d), v(mask)))

    def infer_2d(self, image):
        image = _f32(image)
        assert image.ndim == 4 and image.shape[3] == 3, "image must be [B,H,W,3]"
        B, H, W, _ = image.shape
        kp = np.empty((B, 256, 256, 21), np.float32)
        crop = np.empty((B, 256, 256, 3), np.float32)
        scale = np.empty((B, 1), np.float32)
        center = np.empty((B, 2), np.float32)
        self._chk(self.lib.hp3d_infer_2d(self.h, B, H, W, _ptr(image), _ptr(kp), _ptr(crop), _ptr(scale), _ptr(center)))
        return kp, crop, scale, center

    def infer_2d_keypoints(self, image, want_scoremap=False):
        """inference2d + detect_keypoints + trafo_coords on the device (eval2d.py:58,93-94): returns
        (kp_crop int32 [B,21,2], kp_hw float64 [B,21,2], scale_crop, center[, keypoints_scoremap])."""
        image = _f32(image)
        assert image.ndim == 4 and image.shape[3] == 3, "image must be [B,H,W,3]"
        B, H, W, _ = image.shape
        kp = np.empty((B, 256, 256, 21), np.float32) if want_scoremap else None
        scale = np.empty((B, 1), np.float32)
        center = np.empty((B, 2), np.float32)
        kpc = np.empty((B, 21, 2), np.int32)
        kph = np.empty((B, 21, 2), np.float64)
        self._chk(self.lib.hp3d_infer_2d_kp(self.h, B, H, W, _ptr(image), _ptr(kp), None, _ptr(scale), _ptr(center),
                                            _ptr(kpc), _ptr(kph)))
        return (kpc, kph, scale, center, kp) if want_scoremap else (kpc, kph, scale, center)

    def detect_keypoints(self, scoremap, out_hw=(256, 256)):
        """detect_keypoints(resize_images(scoremap, out_hw)) per image: [B,h,w,C] -> int32 [B,C,2]."""
        x = _f32(scoremap)
        B, h, w, Cc = x.shape
        out = np.empty((B, Cc, 2), np.int32)
        self._chk(self.lib.hp3d_detect_keypoints(self.h, _ptr(x), B, h, w, Cc, int(out_hw[0]), int(out_hw[1]), _ptr(out)))
        return out

    def handsegnet(self, image, want_small=False):
        image = _f32(image)
        assert image.ndim == 4 and image.shape[3] == 3, "image must be [B,H,W,3]"
        B, H, W, _ = image.shape
        large = np.empty((B, H, W, 2), np.float32)
        small = np.empty((B, H // 8, W // 8, 2), np.float32) if want_small else None
        self._chk(self.lib.hp3d_handsegnet(self.h, B, H, W, _ptr(image), _ptr(large), _ptr(small)))
        return (large, small) if want_small else large

    def posenet2d(self, image_crop):
        image_crop = _f32(image_crop)
        assert image_crop.ndim == 4 and image_crop.shape[3] == 3, "image_crop must be [B,H,W,3]"
        B, H, W, _ = image_crop.shape
        outs = [np.empty((B, H // 8, W // 8, 21), np.float32) for _ in range(3)]
        self._chk(self.lib.hp3d_posenet2d(self.h, B, H, W, _ptr(image_crop), *[_ptr(x) for x in outs]))
        return outs

    def poseprior(self, variant, scoremap256, hand_side):
        assert variant in VARIANTS, "Unknown variant."
        sm, hs = _f32(scoremap256), _f32(hand_side)
        assert sm.ndim == 4 and sm.shape[1:] == (256, 256, 21), "scoremap must be [B,256,256,21]"
        B = sm.shape[0]
        rel = np.empty((B, 21, 3), np.float32)
        c3d = np.empty((B, 21, 3), np.float32)
        R = np.empty((B, 3, 3), np.float32)
        self._chk(self.lib.hp3d_poseprior(self.h, B, VARIANTS[variant], _ptr(sm), _ptr(hs), _ptr(rel), _ptr(c3d), _ptr(R)))
        return rel, c3d, (R if variant == 'proposed' else None)

    def pose3d(self, scoremap32, hand_side):
        sm, hs = _f32(scoremap32), _f32(hand_side)
        B = sm.shape[0]
        assert sm.shape[1:] == (32, 32, 21)
        rel = np.empty((B, 21, 3), np.float32)
        can = np.empty((B, 21, 3), np.float32)
        R = np.empty((B, 3, 3), np.float32)
        self._chk(self.lib.hp3d_pose3d(self.h, B, _ptr(sm), _ptr(hs), _ptr(rel), _ptr(can), _ptr(R)))
        return rel, can, R

    # -- per-op ------------------------------------------------------------------------------
    def conv2d(self, x, w, b, stride=1, act=True, pool=False):
        x, w, b = _f32(x), _f32(w), _f32(b)
        B, H, W, Cin = x.shape
        k, _, _, Cout = w.shape
        Ho, Wo = -(-H // stride), -(-W // stride)
        if pool:
            Ho, Wo = Ho // 2, Wo // 2
        out = np.empty((B, Ho, Wo, Cout), np.float32)
        self._chk(self.lib.hp3d_conv2d(self.h, _ptr(x), B, H, W, Cin, _ptr(w), _ptr(b), k, stride, Cout, int(act),
                                       int(pool), _ptr(out)))
        return out

    def maxpool2(self, x):
        x = _f32(x)
        B, H, W, Cc = x.shape
        out = np.empty((B, H // 2, W // 2, Cc), np.float32)
        self._chk(self.lib.hp3d_maxpool2(self.h, _ptr(x), B, H, W, Cc, _ptr(out)))
        return out

    def avgpool8(self, x):
        x = _f32(x)
        B, H, W, Cc = x.shape
        out = np.empty((B, H // 8, W // 8, Cc), np.float32)
        self._chk(self.lib.hp3d_avgpool8(self.h, _ptr(x), B, H, W, Cc, _ptr(out)))
        return out

    def resize_bilinear(self, x, oh, ow):
        x = _f32(x)
        B, H, W, Cc = x.shape
        out = np.empty((B, oh, ow, Cc), np.float32)
        self._chk(self.lib.hp3d_resize_bilinear(self.h, _ptr(x), B, H, W, Cc, oh, ow, _ptr(out)))
        return out

    def crop_and_resize(self, image, center, scale, crop_size=256):
        image, center, scale = _f32(image), _f32(center), _f32(scale).reshape(-1)
        B, H, W, Cc = image.shape
        out = np.empty((B, crop_size, crop_size, Cc), np.float32)
        self._chk(self.lib.hp3d_crop_and_resize(self.h, _ptr(image), B, H, W, Cc, _ptr(center), _ptr(scale), crop_size,
                                                _ptr(out)))
        return out

    def mask_from_scoremap(self, scoremap):
        sm = _f32(scoremap)
        B, H, W, c2 = sm.shape
        assert c2 == 2
        mask = np.empty((B, H, W), np.float32)
        center = np.empty((B, 2), np.float32)
        size = np.empty((B, 1), np.float32)
        scale = np.empty((B, 1), np.float32)
        seed = np.empty((B, 2), np.int32)
        self._chk(self.lib.hp3d_mask_from_scoremap(self.h, _ptr(sm), B, H, W, _ptr(mask), _ptr(center), _ptr(size),
                                                   _ptr(scale), _ptr(seed)))
        return mask, center, size, scale, seed

    def fc(self, x, w, b, act=False):
        x, w, b = _f32(x), _f32(w), _f32(b)
        B, Cin = x.shape
        Cout = w.shape[1]
        out = np.empty((B, Cout), np.float32)
        self._chk(self.lib.hp3d_fc(self.h, _ptr(x), B, Cin, _ptr(w), _ptr(b), Cout, int(act), _ptr(out)))
        return out

    def argmax2d(self, x):
        x = _f32(x)
        B, H, W, Cc = x.shape
        out = np.empty((B, Cc, 2), np.int32)
        self._chk(self.lib.hp3d_argmax2d(self.h, _ptr(x), B, H, W, Cc, _ptr(out)))
        return out

    # -- measurement -------------------------------------------------------------------------
    def set_profiling(self, on):
        self._chk(self.lib.hp3d_set_profiling(self.h, int(on)))

    def counter(self, name):
        v = C.c_longlong()
        self._chk(self.lib.hp3d_get_counter(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def get_timing(self):
        """{stage: GPU ms} over the profiled launches (hp3d_get_timing; stages: TIMING_STAGES)."""
        buf = (C.c_float * len(TIMING_STAGES))()
        self._chk(self.lib.hp3d_get_timing(self.h, buf, len(TIMING_STAGES)))
        return dict(zip(TIMING_STAGES, [float(v) for v in buf]))

    # -- multi-GPU: native RCCL on the engine stream -------------------------------
    def comm_unique_id(self):
        """128-byte RCCL id (create on ONE rank, hand to the others through the launcher's side channel)."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        self._chk(self.lib.hp3d_comm_unique_id(buf))
        return bytes(buf.raw)

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        self._chk(self.lib.hp3d_comm_init(self.h, int(rank), int(world), C.c_char_p(bytes(unique_id))))

    def bcast_weights(self, root=0):
        self._chk(self.lib.hp3d_bcast_weights(self.h, int(root)))

    def allgather(self, local, world):
        """all-gather of equal-sized float32 arrays: [n, ...] per rank -> [world * n, ...]."""
        a = _f32(local)
        out = np.empty((int(world) * a.shape[0],) + a.shape[1:], np.float32)
        self._chk(self.lib.hp3d_allgather(self.h, _ptr(a), a.size, _ptr(out)))
        return out

    def allgather_dev(self, dev_buf, count, world):
        """all-gather straight from a device buffer of `count` floats per rank -> float32 [world * count] on the host."""
        out = np.empty(int(world) * int(count), np.float32)
        self._chk(self.lib.hp3d_allgather_dev(self.h, C.c_void_p(int(dev_buf)), int(count), _ptr(out)))
        return out

    def comm_destroy(self):
        self._chk(self.lib.hp3d_comm_destroy(self.h))

    # -- device / pinned host memory through the C ABI (no torch) ---------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.hp3d_dev_alloc(self.h, int(nbytes), C.byref(p)))
        return DevBuf(self, p.value, int(nbytes))

    def to_device(self, array):
        """Copy a NumPy array into a fresh device buffer (blocking)."""
        a = np.ascontiguousarray(array)
        buf = self.dev_alloc(a.nbytes)
        self._chk(self.lib.hp3d_memcpy(self.h, C.c_void_p(buf.ptr), _ptr(a), a.nbytes, 0))
        return buf

    def to_host(self, buf, shape, dtype=np.float32, offset_bytes=0):
        out = np.empty(shape, dtype)
        self._chk(self.lib.hp3d_memcpy(self.h, _ptr(out), C.c_void_p(int(buf) + int(offset_bytes)), out.nbytes, 1))
        return out

    def pinned_empty(self, shape, dtype=np.float32):
        """NumPy array over page-locked host memory (for hp3d_upload_async).  The memory belongs to the engine: it is released
        by close() (after pending uploads have finished), so the array must not be used past that point."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self.lib.hp3d_host_alloc(self.h, n, C.byref(p)))
        arr = np.frombuffer((C.c_char * n).from_address(p.value), dtype=dtype).reshape(shape)
        self._pinned.append(p.value)
        return arr

    def upload_async(self, buf, pinned_array):
        self._chk(self.lib.hp3d_upload_async(self.h, C.c_void_p(int(buf)), _ptr(pinned_array), pinned_array.nbytes))

    def wait_upload(self):
        self._chk(self.lib.hp3d_wait_upload(self.h))

    def profile(self):
        """[(layer, kernel, ms, flops, bytes)] of the last whole-path call."""
        n = self.lib.hp3d_prof_count(self.h)
        rows = []
        name, kern = C.create_string_buffer(96), C.create_string_buffer(96)
        ms, fl, by = C.c_float(), C.c_double(), C.c_double()
        for i in range(n):
            self._chk(self.lib.hp3d_prof_get(self.h, i, name, 96, kern, 96, C.byref(ms), C.byref(fl), C.byref(by)))
            rows.append((name.value.decode(), kern.value.decode(), ms.value, fl.value, by.value))
        return rows
