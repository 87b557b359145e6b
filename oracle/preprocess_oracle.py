"""CPU restatement of the reference's input transform for the extraction loaders -- TEST INFRASTRUCTURE ONLY
(imported by tests/, tools/make_golden.py; never by the product path).

selftraining.py:43-47 (get_data) / :66-70 (get_source_data):
    T.Compose([Resize((height, width)), T.ToTensor(), T.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
applied by reid/utils/data/preprocessor.py:22-30 to `Image.open(fpath).convert('RGB')`.

torchvision is not vendored by the reference and absent from this image; what its three transforms do to a PIL RGB image is
published behaviour and restated here:
  * Resize((h, w))      -> img.resize((w, h), Image.BILINEAR): Pillow's two-pass separable resampling on 8-bit channels
                           (libImaging/Resample.c: triangle filter whose support grows with the down-scale factor, float64
                           coefficients normalised per output pixel, quantised to 22-bit fixed point, horizontal pass first,
                           8-bit intermediate, round-half-up by the initial 1 << 21);
  * ToTensor()          -> uint8 HWC -> float32 CHW / 255;
  * Normalize(mean,std) -> (x - mean[c]) / std[c] in float32.
Pinned: tools/make_golden.py asserts `resize_bilinear_u8` == PIL.Image.resize (Pillow 12.2.0 in this image) bit for bit on
random images of many shapes (down-, up-scaling, identity) and stores PIL's own outputs in tests/golden/preprocess.npz.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def bilinear_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the triangle filter over the whole axis.
    -> (xmin[out], xcnt[out], kk[out, ksize] int32)"""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32); xcnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        x = (np.arange(n, dtype=np.float64) + lo - center + 0.5) * ss
        w = np.where(np.abs(x) < 1.0, 1.0 - np.abs(x), 0.0)
        ww = 0.0
        for v in w:            # sequential float64 sum, as the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        xmin[xx] = lo; xcnt[xx] = n; kk[xx, :n] = q
    return xmin, xcnt, kk


def _pass(img, xmin, xcnt, kk, axis):
    """one resampling pass over `axis` of a uint8 [H, W, C] image"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)          # [in, other, C]
    out = np.empty((len(xmin),) + src.shape[1:], np.uint8)
    for xx in range(len(xmin)):
        n = int(xcnt[xx]); lo = int(xmin[xx])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, height, width):
    """uint8 [h, w, 3] -> uint8 [height, width, 3] == PIL.Image.fromarray(img).resize((width, height), Image.BILINEAR)"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    out = img
    if w != width:                       # horizontal pass first (ImagingResample), 8-bit intermediate
        out = _pass(out, *bilinear_coeffs(w, width), axis=1)
    if h != height:
        out = _pass(out, *bilinear_coeffs(h, height), axis=0)
    return out


def to_tensor_normalize(img_u8, mean=MEAN, std=STD):
    """ToTensor + Normalize: uint8 [H, W, 3] -> float32 [3, H, W]"""
    x = np.transpose(img_u8, (2, 0, 1)).astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32)[:, None, None]; s = np.asarray(std, np.float32)[:, None, None]
    return (x - m) / s


def transform(img_u8, height, width):
    return to_tensor_normalize(resize_bilinear_u8(img_u8, height, width))
