"""Python side of the image kernels of the pose data pipeline (das_amd/csrc/augment.hip, C ABI in include/das_hip.h):
f32 HWC (BGR) CUDA tensors in, CUDA tensors out. No CPU fallback: a CPU tensor raises."""
import ctypes as C

import torch

from . import _lib
from .ops import _need_gpu, _ptr, _stream


class FramePlan:
    """A decoded frame (HWC uint8 on the HOST) and the image ops recorded for it, to be run on the GPU later — by another
    process than the one that drew the augmentation parameters (das_amd.loader.ProcessLoader: worker processes decode
    and do the annotation arithmetic without ever touching the GPU; the trainer replays the recorded ops).
    Every function of this module accepts a plan in place of a tensor: it then only records its arguments and returns a
    NEW plan with the shape the result will have (plans are immutable: a test-time pipeline copies `results` per
    augmentation). `run` performs exactly the calls the immediate path would have made, in the same order, on the
    uploaded frame — the two paths are bit-identical by construction (tests/test_pipeline_gpu.py)."""
    __slots__ = ('frame', 'rgb', 'to_float', 'ops', 'shape')

    def __init__(self, frame, rgb=False, to_float=True, ops=(), shape=None):
        self.frame, self.rgb, self.to_float, self.ops = frame, bool(rgb), bool(to_float), tuple(ops)
        self.shape = tuple(frame.shape) if shape is None else tuple(shape)

    @property
    def dtype(self):
        return torch.float32 if self.to_float or any(op == 'normalize_pad_chw' for op, _ in self.ops) else torch.uint8

    def then(self, op, args, shape):
        return FramePlan(self.frame, self.rgb, self.to_float, self.ops + ((op, args),), shape)

    def with_frame(self, frame):
        """The same plan on another copy of the frame (a view of shared memory, a device tensor)."""
        return FramePlan(frame, self.rgb, self.to_float, self.ops, self.shape)

    def run(self, device, out=None):
        """Upload (unless the frame already is a device tensor), then the recorded ops; `out`: destination of a final
        normalize_pad_chw (a slot of the batch tensor)."""
        t = self.frame
        if not torch.is_tensor(t):
            t = torch.from_numpy(t)
        if not t.is_cuda:
            t = t.to(device, non_blocking=True)
        if self.rgb:
            t = t.flip(-1)           # RGB (as PIL decodes) -> BGR (as the reference's cv2.imread)
        if self.to_float:
            t = t.float()
        t = t.contiguous()
        fns = globals()
        for i, (op, args) in enumerate(self.ops):
            if op == 'normalize_pad_chw' and i == len(self.ops) - 1:
                t = normalize_pad_chw(t, *args, out=out)
            else:
                t = fns[op](t, *args)
        return t


def _hwc(img):
    _need_gpu(img)
    assert img.dim() == 3 and img.dtype == torch.float32 and img.is_contiguous(), (img.shape, img.dtype)
    return img.shape


def resize_bilinear(img, size):
    """cv2.resize(img, (Wd, Hd), INTER_LINEAR) (mmcv.imresize / imrescale): float32 images (train pipeline, ResizePose)
    or uint8 images (test pipeline: OpenCV's fixed-point path)."""
    Wd, Hd = int(size[0]), int(size[1])
    if isinstance(img, FramePlan):
        return img.then('resize_bilinear', ((Wd, Hd),), (Hd, Wd, img.shape[2]))
    if img.dtype == torch.uint8:
        _need_gpu(img)
        assert img.dim() == 3 and img.is_contiguous()
        H, W, Cc = img.shape
        out = torch.empty(Hd, Wd, Cc, dtype=torch.uint8, device=img.device)
        _lib.check(_lib.load().das_img_resize_bilinear_u8(_ptr(img), _ptr(out), H, W, Hd, Wd, Cc, _stream()),
                   'das_img_resize_bilinear_u8')
        return out
    H, W, Cc = _hwc(img)
    out = torch.empty(Hd, Wd, Cc, dtype=torch.float32, device=img.device)
    _lib.check(_lib.load().das_img_resize_bilinear(_ptr(img), _ptr(out), H, W, Hd, Wd, Cc, _stream()), 'das_img_resize_bilinear')
    return out


def flip_horizontal(img):
    if isinstance(img, FramePlan):
        return img.then('flip_horizontal', (), img.shape)
    H, W, Cc = _hwc(img)
    out = torch.empty_like(img)
    _lib.check(_lib.load().das_img_flip_horizontal(_ptr(img), _ptr(out), H, W, Cc, _stream()), 'das_img_flip_horizontal')
    return out


def photometric_(img, brightness=None, contrast=None, contrast_first=True, saturation=None, hue=None, perm=None):
    """mmdet PhotoMetricDistortion with the drawn parameters (None = step not applied), in place."""
    if isinstance(img, FramePlan):
        return img.then('photometric_', (brightness, contrast, contrast_first, saturation, hue,
                                         None if perm is None else tuple(int(v) for v in perm)), img.shape)
    H, W, Cc = _hwc(img)
    assert Cc == 3
    p = _lib.DasPhotometric(use_brightness=int(brightness is not None), use_contrast=int(contrast is not None),
                            contrast_first=int(bool(contrast_first)), use_saturation=int(saturation is not None),
                            use_hue=int(hue is not None), brightness=float(brightness or 0.0),
                            contrast=float(contrast if contrast is not None else 1.0),
                            saturation=float(saturation if saturation is not None else 1.0), hue=float(hue or 0.0))
    for c, v in enumerate(perm if perm is not None else (0, 1, 2)):
        p.perm[c] = int(v)
    _lib.check(_lib.load().das_img_photometric(_ptr(img), H, W, C.byref(p), _stream()), 'das_img_photometric')
    return img


def warp_affine(img, M, size, border):
    """cv2.warpAffine(img, M (forward 2x3, f64), (Wd, Hd), INTER_LINEAR, BORDER_CONSTANT, borderValue=border)."""
    if isinstance(img, FramePlan):
        M = tuple(tuple(float(v) for v in row) for row in M)
        return img.then('warp_affine', (M, (int(size[0]), int(size[1])), tuple(float(v) for v in border)),
                        (int(size[1]), int(size[0]), 3))
    H, W, Cc = _hwc(img)
    assert Cc == 3
    Wd, Hd = int(size[0]), int(size[1])
    out = torch.empty(Hd, Wd, 3, dtype=torch.float32, device=img.device)
    m = (C.c_double * 6)(*[float(v) for v in list(M[0]) + list(M[1])])
    b = (C.c_float * 3)(*[float(v) for v in border])
    _lib.check(_lib.load().das_img_warp_affine(_ptr(img), _ptr(out), H, W, Hd, Wd, m, b, _stream()), 'das_img_warp_affine')
    return out


def normalize_pad_chw(img, mean, std, to_rgb, pad_hw=None, out=None):
    """mmcv.imnormalize + zero pad to pad_hw + HWC -> CHW in one pass; `out` may be a (3, Hp, Wp) slice of a batch."""
    if isinstance(img, FramePlan):
        assert out is None
        Hp, Wp = pad_hw if pad_hw is not None else img.shape[:2]
        return img.then('normalize_pad_chw', (tuple(float(v) for v in mean), tuple(float(v) for v in std), bool(to_rgb),
                                              (int(Hp), int(Wp))), (3, int(Hp), int(Wp)))
    if img.dtype == torch.uint8:      # (mmcv.imnormalize: img.astype(np.float32) first; exact)
        img = img.float()
    H, W, Cc = _hwc(img)
    assert Cc == 3
    Hp, Wp = pad_hw if pad_hw is not None else (H, W)
    if out is None:
        out = torch.empty(3, Hp, Wp, dtype=torch.float32, device=img.device)
    assert out.shape == (3, Hp, Wp) and out.is_contiguous() and out.dtype == torch.float32
    m = (C.c_double * 3)(*[float(v) for v in mean])
    s = (C.c_double * 3)(*[float(v) for v in std])
    _lib.check(_lib.load().das_img_normalize_pad_chw(_ptr(img), _ptr(out), H, W, Hp, Wp, m, s, int(bool(to_rgb)), _stream()),
               'das_img_normalize_pad_chw')
    return out
