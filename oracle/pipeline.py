"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the image half of the reference's pose data pipeline
(mmdet3d/datasets/pipelines/transforms_3d.py:19-61,235-356,864-1129; configs/das/exp_panoptic.py:59-98). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path (das_amd/pipelines.py) runs
the HIP kernels of das_amd/csrc/augment.hip.

PARITY UNPINNED for everything in this file: the arithmetic lives in third-party packages that are absent from
/root/reference and from this image — OpenCV 4.x (`cv2.resize`, `cv2.warpAffine`, `cv2.cvtColor`, `cv2.subtract /
multiply`, `cv2.getAffineTransform`), mmcv-full 1.3.10 (`imrescale`, `imflip`, `imnormalize`, `impad_to_multiple`) and
mmdet 2.14.0 (`Resize`, `RandomFlip`, `PhotoMetricDistortion`, `Normalize`, `Pad`). It restates their published
algorithms (OpenCV: modules/imgproc/src/{resize,imgwarp,color_hsv}.cpp, modules/core/src/arithm.cpp). The reference's
OWN arithmetic on this path — the annotation transforms — is pinned against the imported reference files
(tests/golden/make_golden_pipeline.py -> tests/golden/pipeline_*.npz)."""
import numpy as np

F32 = np.float32
EPS = np.float32(1.1920929e-07)


# ------------------------------------------------------------------ cv2.resize(INTER_LINEAR), float images
def resize_bilinear(img, size):
    """img (H, W, C) f32; size = (Wd, Hd) as cv2 takes it."""
    Hs, Ws = img.shape[:2]
    Wd, Hd = size
    scale_x, scale_y = 1.0 / (float(Wd) / Ws), 1.0 / (float(Hd) / Hs)

    def axis(n_dst, n_src, scale):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(F32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(F32)).astype(F32)
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0, n_src - 1
        return s, np.minimum(s + 1, n_src - 1), f
    sx, sx1, fx = axis(Wd, Ws, scale_x)
    sy, sy1, fy = axis(Hd, Hs, scale_y)
    a0, a1 = (F32(1) - fx)[None, :, None], fx[None, :, None]
    b0, b1 = (F32(1) - fy)[:, None, None], fy[:, None, None]
    r0 = img[sy][:, sx] * a0 + img[sy][:, sx1] * a1
    r1 = img[sy1][:, sx] * a0 + img[sy1][:, sx1] * a1
    return (r0 * b0 + r1 * b1).astype(F32)


def resize_bilinear_u8(img, size):
    """cv2.resize(INTER_LINEAR) on uint8: 11-bit fixed-point coefficients, int32 horizontal pass, the vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    Hs, Ws = img.shape[:2]
    Wd, Hd = size
    scale_x, scale_y = 1.0 / (float(Wd) / Ws), 1.0 / (float(Hd) / Hs)

    def axis(n_dst, n_src, scale):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(F32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(F32)).astype(F32)
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0, n_src - 1
        c0 = np.clip(np.rint((F32(1) - f) * F32(2048)), -32768, 32767).astype(np.int64)
        c1 = np.clip(np.rint(f * F32(2048)), -32768, 32767).astype(np.int64)
        return s, np.minimum(s + 1, n_src - 1), c0, c1
    sx, sx1, a0, a1 = axis(Wd, Ws, scale_x)
    sy, sy1, b0, b1 = axis(Hd, Hs, scale_y)
    im = img.astype(np.int64)
    r0 = im[sy][:, sx] * a0[None, :, None] + im[sy][:, sx1] * a1[None, :, None]
    r1 = im[sy1][:, sx] * a0[None, :, None] + im[sy1][:, sx1] * a1[None, :, None]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def rescale_size(w, h, scale):
    """mmcv.rescale_size for a (long, short) tuple: new (w, h) and the factor."""
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def flip_horizontal(img):
    return np.ascontiguousarray(img[:, ::-1])


# ------------------------------------------------------------------ mmdet PhotoMetricDistortion on OpenCV's float HSV
def bgr2hsv(img):
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(r, np.maximum(g, b))
    vmin = np.minimum(r, np.minimum(g, b))
    diff = (v - vmin).astype(F32)
    s = (diff / (np.abs(v) + EPS)).astype(F32)
    diff = (F32(60) / (diff + EPS)).astype(F32)
    h = np.where(v == r, (g - b) * diff, np.where(v == g, (b - r) * diff + F32(120), (r - g) * diff + F32(240))).astype(F32)
    h = np.where(h < 0, h + F32(360), h).astype(F32)
    return np.stack([h, s, v], -1)


def hsv2bgr(img):
    h, s, v = img[..., 0].copy(), img[..., 1], img[..., 2]
    hh = (h * F32(6.0 / 360.0)).astype(F32)
    while (hh < 0).any():
        hh = np.where(hh < 0, hh + F32(6), hh).astype(F32)
    while (hh >= 6).any():
        hh = np.where(hh >= 6, hh - F32(6), hh).astype(F32)
    sector = np.floor(hh).astype(np.int64)
    hh = (hh - sector.astype(F32)).astype(F32)
    bad = (sector < 0) | (sector >= 6)
    sector[bad], hh[bad] = 0, 0
    one = F32(1)
    tab = np.stack([v, v * (one - s), v * (one - s * hh), v * (one - s * (one - hh))], -1).astype(F32)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sd[sector]
    out = np.take_along_axis(tab, idx, -1)
    grey = (s == 0)[..., None]
    return np.where(grey, v[..., None], out).astype(F32)


def photometric(img, p):
    """p: dict(brightness, contrast, contrast_first, saturation, hue, perm) with None = step not applied."""
    img = img.astype(F32).copy()
    if p.get('brightness') is not None:
        img += F32(p['brightness'])
    if p.get('contrast_first') and p.get('contrast') is not None:
        img *= F32(p['contrast'])
    hsv = bgr2hsv(img)
    if p.get('saturation') is not None:
        hsv[..., 1] *= F32(p['saturation'])
    if p.get('hue') is not None:
        hsv[..., 0] += F32(p['hue'])
        hsv[..., 0][hsv[..., 0] > 360] -= 360
        hsv[..., 0][hsv[..., 0] < 0] += 360
    img = hsv2bgr(hsv)
    if not p.get('contrast_first') and p.get('contrast') is not None:
        img *= F32(p['contrast'])
    if p.get('perm') is not None:
        img = img[..., list(p['perm'])]
    return np.ascontiguousarray(img, dtype=F32)


# ------------------------------------------------------------------ cv2.getAffineTransform / warpAffine
def get_affine_transform_cv(src, dst):
    """cv2.getAffineTransform: the 2x3 map taking three src points to three dst points (f64 solve)."""
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[i, :3] = [src[i][0], src[i][1], 1]
        A[i + 3, 3:] = [src[i][0], src[i][1], 1]
        b[i], b[i + 3] = dst[i][0], dst[i][1]
    return np.linalg.solve(A, b).reshape(2, 3)


def invert_affine_cv(M):
    m = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(-1)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def warp_affine(img, M, size, border):
    """cv2.warpAffine(img f32 HWC, M forward 2x3, (Wd, Hd), INTER_LINEAR, BORDER_CONSTANT, borderValue=border)."""
    Hs, Ws = img.shape[:2]
    Wd, Hd = size
    m = invert_affine_cv(M)
    AB, IB = 10, 5
    ABS, TAB = 1 << AB, 1 << IB
    rd = ABS // TAB // 2
    xs, ys = np.arange(Wd, dtype=np.float64), np.arange(Hd, dtype=np.float64)
    adelta = np.rint(m[0] * xs * ABS).astype(np.int64)
    bdelta = np.rint(m[3] * xs * ABS).astype(np.int64)
    X0 = np.rint((m[1] * ys + m[2]) * ABS).astype(np.int64) + rd
    Y0 = np.rint((m[4] * ys + m[5]) * ABS).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None]) >> (AB - IB)
    Y = (Y0[:, None] + bdelta[None]) >> (AB - IB)
    sx = np.clip(X >> IB, -32768, 32767)
    sy = np.clip(Y >> IB, -32768, 32767)
    fx = ((X & (TAB - 1)).astype(F32) * F32(1.0 / TAB)).astype(F32)
    fy = ((Y & (TAB - 1)).astype(F32) * F32(1.0 / TAB)).astype(F32)
    one = F32(1)
    w = [(one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx]
    border = np.asarray(border, dtype=F32)
    out = np.zeros((Hd, Wd, 3), dtype=F32)
    acc = None
    for k in range(4):
        yy, xx = sy + (k >> 1), sx + (k & 1)
        ok = (yy >= 0) & (yy < Hs) & (xx >= 0) & (xx < Ws)
        v = np.where(ok[..., None], img[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)], border[None, None]).astype(F32)
        t = (v * w[k][..., None]).astype(F32)
        acc = t if acc is None else (acc + t).astype(F32)
    all_out = (sx >= Ws) | (sx + 1 < 0) | (sy >= Hs) | (sy + 1 < 0)
    out[:] = np.where(all_out[..., None], border[None, None], acc)
    return out


# ------------------------------------------------------------------ mmcv.imnormalize + impad_to_multiple + HWC -> CHW
def normalize(img, mean, std, to_rgb=True):
    mean = np.float64(np.asarray(mean).reshape(1, -1))
    stdinv = 1 / np.float64(np.asarray(std).reshape(1, -1))
    img = img[..., ::-1] if to_rgb else img

    def scalar_f64(v):       # cv::arithm_op: a Scalar that is not integer-valued makes the working type f64
        return bool((v != np.rint(v)).any())
    d = (img.astype(np.float64) - mean).astype(F32) if scalar_f64(mean) else (img - mean.astype(F32)).astype(F32)
    return (d.astype(np.float64) * stdinv).astype(F32) if scalar_f64(stdinv) else (d * stdinv.astype(F32)).astype(F32)


def pad_to_multiple(img, divisor):
    H, W = img.shape[:2]
    Hp, Wp = int(np.ceil(H / divisor)) * divisor, int(np.ceil(W / divisor)) * divisor
    out = np.zeros((Hp, Wp) + img.shape[2:], dtype=img.dtype)
    out[:H, :W] = img
    return out
