"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's training-batch assembly, the contract of the input
pipeline: mit_semseg/dataset.py:110-199 (TrainDataset.__getitem__) given decoded uint8 arrays and the random
choices it draws (flip flags, short-side size).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product path (mit_semseg/data.py + csrc/input_pipeline.hip) never does.

The arithmetic of the image path lives in a third-party dependency of the reference, Pillow (`im.resize`, dataset.py:9-19).
It is restated here from Pillow's published algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc; src/libImaging/Geometry.c: ImagingScaleAffine for NEAREST) and PINNED against
the Pillow in this image (12.2.0) by tests/test_input_oracle_cpu.py over random sizes, and against the unmodified reference
class (tests/golden/make_input_golden.py -> tests/golden/input_golden.npz).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8 bits for the result, 2 spare for overshoot
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)        # dataset.py:34-36
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def round2nearest_multiple(x, p):
    """dataset.py:66-67"""
    return ((x - 1) // p + 1) * p


# ---------------------------------------------------------------------------------------------------------
# Pillow BILINEAR (triangle filter, support scaled by the reduction factor = antialiasing), 8 bits per channel
# ---------------------------------------------------------------------------------------------------------
def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the full box [0, in_size).
    Returns (bounds int32 [out][2] = (xmin, count), kk int32 [out][ksize])."""
    scale = in_size / out_size                     # C doubles
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                    # bilinear: support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)         # C cast: truncation (values are >= -0.5 here)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            v = 1.0 - a if a < 1.0 else 0.0
            w[x] = v
            ww += v
        if ww != 0.0:
            for x in range(xmax):
                w[x] /= ww
        for x in range(ksize):                     # normalize_coeffs_8bpc
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis0(img, out_size):
    """resample along axis 0 of a [in][...] uint8 array"""
    bounds, kk = resample_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = _clip8(acc)
    return out


def pil_resize_bilinear(img, size):
    """PIL.Image.resize(size=(w, h), Image.BILINEAR) of a uint8 [H][W] or [H][W][C] array: horizontal pass, then vertical
    (Resample.c ImagingResampleInner; a pass whose size does not change is skipped)."""
    ow, oh = int(size[0]), int(size[1])
    h, w = img.shape[:2]
    out = img
    if ow != w:
        out = np.swapaxes(_resample_axis0(np.swapaxes(out, 0, 1), ow), 0, 1)
    if oh != h:
        out = _resample_axis0(out, oh)
    return np.ascontiguousarray(out)


# ---------------------------------------------------------------------------------------------------------
# Pillow NEAREST (Geometry.c ImagingScaleAffine: coordinates advance by repeated double additions)
# ---------------------------------------------------------------------------------------------------------
def nearest_index_table(in_size, out_size):
    """source index per output index, -1 where the source coordinate falls outside (left at 0 by Pillow)"""
    a = in_size / out_size
    xo = a * 0.5                                   # a[2] + a[0] * 0.5 with box origin 0
    tab = np.full(out_size, -1, dtype=np.int32)
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        if 0 <= xin < in_size:
            tab[x] = xin
        xo += a
    return tab


def pil_resize_nearest(img, size):
    ow, oh = int(size[0]), int(size[1])
    h, w = img.shape[:2]
    if (ow, oh) == (w, h):
        return img.copy()
    xt, yt = nearest_index_table(w, ow), nearest_index_table(h, oh)
    out = np.zeros((oh, ow) + img.shape[2:], dtype=img.dtype)
    ys, xs = np.nonzero(yt >= 0)[0], np.nonzero(xt >= 0)[0]
    out[np.ix_(ys, xs)] = img[np.ix_(yt[ys], xt[xs])]
    return out


# ---------------------------------------------------------------------------------------------------------
# dataset.py:110-199
# ---------------------------------------------------------------------------------------------------------
def batch_geometry(sizes_hw, this_short_size, img_max_size, padding_constant):
    """dataset.py:127-142: per-sample resized (w, h) and the padded batch (H, W).  sizes_hw: [(height, width)]"""
    n = len(sizes_hw)
    bw = np.zeros(n, np.int32)
    bh = np.zeros(n, np.int32)
    for i, (ih, iw) in enumerate(sizes_hw):
        this_scale = min(this_short_size / min(ih, iw), img_max_size / max(ih, iw))
        bw[i] = iw * this_scale                    # float -> int32 store: truncation
        bh[i] = ih * this_scale
    batch_w = int(round2nearest_multiple(np.max(bw), padding_constant))
    batch_h = int(round2nearest_multiple(np.max(bh), padding_constant))
    return bw, bh, batch_h, batch_w


def image_to_tensor(img):
    """dataset.py:53-58 + torchvision Normalize: float32(u8) / 255., then (t - mean) / std in fp32; returns [3][h][w]"""
    t = (img.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)
    return (t - MEAN[:, None, None]) / STD[:, None, None]


def assemble_train_batch(images, segms, flips, this_short_size, img_max_size, padding_constant, segm_downsampling_rate):
    """images: list of uint8 [H][W][3] (decoded RGB); segms: list of uint8 [H][W]; flips: list of bool.
    Returns {'img_data': float32 [B][3][BH][BW], 'seg_label': int64 [B][BH/s][BW/s]} as dataset.py:143-196."""
    s = segm_downsampling_rate
    assert padding_constant >= s
    bw, bh, batch_h, batch_w = batch_geometry([im.shape[:2] for im in images], this_short_size, img_max_size, padding_constant)
    batch_images = np.zeros((len(images), 3, batch_h, batch_w), dtype=np.float32)
    batch_segms = np.zeros((len(images), batch_h // s, batch_w // s), dtype=np.int64)
    for i, (img, segm) in enumerate(zip(images, segms)):
        assert img.shape[:2] == segm.shape
        if flips[i]:
            img, segm = img[:, ::-1], segm[:, ::-1]
        img = pil_resize_bilinear(np.ascontiguousarray(img), (bw[i], bh[i]))
        segm = pil_resize_nearest(np.ascontiguousarray(segm), (bw[i], bh[i]))
        rw, rh = round2nearest_multiple(segm.shape[1], s), round2nearest_multiple(segm.shape[0], s)
        canvas = np.zeros((rh, rw), dtype=np.uint8)
        canvas[:segm.shape[0], :segm.shape[1]] = segm
        segm = pil_resize_nearest(canvas, (rw // s, rh // s))
        t = image_to_tensor(img)
        batch_images[i, :, :t.shape[1], :t.shape[2]] = t
        batch_segms[i, :segm.shape[0], :segm.shape[1]] = segm.astype(np.int64) - 1
    return {'img_data': batch_images, 'seg_label': batch_segms}


def eval_image_inputs(img, img_sizes, img_max_size, padding_constant, segm=None):
    """dataset.py:210-255 (ValDataset) / 263-296 (TestDataset): per short-side size the whole image resized to the target
    rounded up to a multiple of padding_constant, normalised -> [1][3][th][tw] fp32; label map: int64 - 1, [1][H][W]"""
    h, w = img.shape[:2]
    outs = []
    for this_short_size in img_sizes:
        scale = min(this_short_size / float(min(h, w)), img_max_size / float(max(h, w)))
        th, tw = int(h * scale), int(w * scale)
        tw = round2nearest_multiple(tw, padding_constant)
        th = round2nearest_multiple(th, padding_constant)
        outs.append(image_to_tensor(pil_resize_bilinear(img, (tw, th)))[None])
    out = {'img_data': outs}
    if segm is not None:
        out['seg_label'] = (segm.astype(np.int64) - 1)[None]
    return out
