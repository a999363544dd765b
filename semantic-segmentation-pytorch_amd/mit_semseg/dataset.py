"""Training-batch assembly on the device -- the input-pipeline contract of the reference (mit_semseg/dataset.py:110-199,
`TrainDataset.__getitem__`): per-sample resize to the drawn short-side size (Pillow BILINEAR for the image, NEAREST for the
label map), random horizontal flip, label down-sampling by `segm_downsampling_rate` through a zero canvas, ToTensor +
Normalize, label - 1, zero-padded placement into the batch.  The reference does all of it on 16 CPU workers
(train.py:163-177); here only the file decode stays on the host, everything from the decoded uint8 array on runs as HIP
kernels (csrc/input_pipeline.hip), bit-exact against Pillow.

    asm = TrainBatchAssembler(imgSizes, imgMaxSize, padding_constant, segm_downsampling_rate, device)
    feed = asm.assemble(images_u8, segms_u8, flips, this_short_size)      # {'img_data', 'seg_label'} on the device

`TrainDataset` mirrors the reference class (same record grouping, same numpy random draws) on top of the assembler.
Its items are device tensors: iterate it directly or through a `DataLoader(..., batch_size=1, num_workers=0,
collate_fn=user_scattered_collate)` -- the 16 CPU workers of train.py:163-177 existed to do on the host what the kernels do here;
only the file decode is left, and that can be prefetched by any host-side loader that yields uint8 arrays.
There is no CPU fallback: without the HIP library / a HIP device the assembler raises."""
import functools
import json
import os

import numpy as np
import torch

from . import _native
from .ops import _p, _st, _require_cuda

MEAN = (0.485, 0.456, 0.406)         # dataset.py:34-36
STD = (0.229, 0.224, 0.225)


def round2nearest_multiple(x, p):
    """dataset.py:66-67: the smallest multiple of p that is >= x"""
    return ((x - 1) // p + 1) * p


@functools.lru_cache(maxsize=512)
def _resample_tables(in_size, out_size):
    """Pillow tap tables (host): int32 [out*2] bounds followed by [out*ksize] taps, and ksize"""
    L = _native.lib()
    ks = L.semseg_input_resample_ksize(in_size, out_size)
    buf = np.zeros(out_size * 2 + out_size * ks, dtype=np.int32)
    _native.check(L.semseg_input_resample_coeffs(in_size, out_size, buf.ctypes.data, buf[out_size * 2:].ctypes.data),
                  'input_resample_coeffs')
    return buf, ks


@functools.lru_cache(maxsize=512)
def _nearest_table(in_size, out_size):
    L = _native.lib()
    tab = np.zeros(out_size, dtype=np.int32)
    _native.check(L.semseg_input_nearest_table(in_size, out_size, tab.ctypes.data), 'input_nearest_table')
    return tab


def _label_axis_table(src_size, mid_size, rate, flip):
    """index table of one axis of the label path: NEAREST src_size -> mid_size, paste into a zero canvas of
    round2nearest_multiple(mid_size, rate), NEAREST canvas -> canvas / rate (dataset.py:168-179); -1 = canvas / outside"""
    t1 = _nearest_table(src_size, mid_size)
    canvas = round2nearest_multiple(mid_size, rate)
    t2 = _nearest_table(canvas, canvas // rate)
    inside = (t2 >= 0) & (t2 < mid_size)
    out = np.full(t2.shape, -1, dtype=np.int32)
    src = t1[np.clip(t2, 0, mid_size - 1)]
    if flip:
        src = np.where(src >= 0, src_size - 1 - src, -1)
    out[inside] = src[inside]
    return out


def batch_geometry(sizes_hw, this_short_size, imgMaxSize, padding_constant):
    """dataset.py:127-142: per-sample resized widths / heights (int32 truncation) and the padded batch (H, W)"""
    n = len(sizes_hw)
    bw, bh = np.zeros(n, np.int32), np.zeros(n, np.int32)
    for i, (h, w) in enumerate(sizes_hw):
        this_scale = min(this_short_size / min(h, w), imgMaxSize / max(h, w))
        bw[i] = w * this_scale
        bh[i] = h * this_scale
    batch_w = int(round2nearest_multiple(np.max(bw), padding_constant))
    batch_h = int(round2nearest_multiple(np.max(bh), padding_constant))
    return bw, bh, batch_h, batch_w


def batch_shape_stream(sizes_wh, counts, imgSizes=(300, 375, 450, 525, 600), imgMaxSize=1000, padding_constant=8,
                       batch_per_gpu=2, seed=0):
    """Endless generator of the padded per-GPU batch shapes (H, W) that `TrainDataset` produces for a record list with
    the given size histogram (`sizes_wh[i]` = (width, height) occurring `counts[i]` times) -- the record grouping
    (portrait / landscape lists, dataset.py:85-108), the reshuffle at every pass, the short-side draw and the padding rule
    of dataset.py:110-142, without touching any image.  Synthetic benchmarks / tests of the variable-size path (BASELINE
    configs[3]) draw their shapes from it; every rank passes its own `seed` (train.py gives each GPU its own loader)."""
    rng = np.random.RandomState(seed)
    recs = np.repeat(np.arange(len(counts)), counts)
    rng.shuffle(recs)
    groups = ([], [])
    cur = 0
    while True:
        w, h = sizes_wh[recs[cur]]
        g = groups[0 if h > w else 1]
        g.append((int(h), int(w)))
        cur += 1
        if cur >= len(recs):
            cur = 0
            rng.shuffle(recs)
        if len(g) == batch_per_gpu:
            short = rng.choice(imgSizes) if isinstance(imgSizes, (list, tuple)) else imgSizes
            _, _, bh, bw = batch_geometry(g, short, imgMaxSize, padding_constant)
            del g[:]
            yield bh, bw


class TrainBatchAssembler:
    def __init__(self, imgSizes, imgMaxSize, padding_constant, segm_downsampling_rate, device='cuda'):
        assert padding_constant >= segm_downsampling_rate, \
            'padding constant must be equal or large than segm downsamping rate'            # dataset.py:141-142
        self.imgSizes, self.imgMaxSize = imgSizes, imgMaxSize
        self.padding_constant, self.segm_downsampling_rate = padding_constant, segm_downsampling_rate
        self.device = torch.device(device)
        self._mean_std = (_native.c_f * 6)(*MEAN, *STD)

    def geometry(self, sizes_hw, this_short_size):
        return batch_geometry(sizes_hw, this_short_size, self.imgMaxSize, self.padding_constant)

    def assemble(self, images, segms, flips, this_short_size):
        """images: uint8 [H][W][3] tensors (decoded RGB), segms: uint8 [H][W] tensors, host or device; flips: bools.
        Returns {'img_data': fp32 logical [B,3,BH,BW] in NHWC memory, 'seg_label': int64 [B,BH/s,BW/s]} on the device."""
        L = _native.lib()
        dev = self.device
        s = self.segm_downsampling_rate
        bw, bh, BH, BW = self.geometry([tuple(im.shape[:2]) for im in images], this_short_size)
        B = len(images)
        img_data = torch.zeros((B, BH, BW, 3), dtype=torch.float32, device=dev)
        seg_label = torch.zeros((B, BH // s, BW // s), dtype=torch.int64, device=dev)
        _require_cuda(img_data)
        for i in range(B):
            img = images[i].to(dev, non_blocking=True).contiguous()
            seg = segms[i].to(dev, non_blocking=True).contiguous()
            if img.dtype != torch.uint8 or seg.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or \
                    tuple(seg.shape) != tuple(img.shape[:2]):
                raise ValueError('expected uint8 [H,W,3] image and uint8 [H,W] label map of the same size')
            H, W = int(img.shape[0]), int(img.shape[1])
            ow, oh = int(bw[i]), int(bh[i])
            flip = bool(flips[i])
            th, ksh = _resample_tables(W, ow)
            tv, ksv = _resample_tables(H, oh)
            ytab = _label_axis_table(H, oh, s, False)
            xtab = _label_axis_table(W, ow, s, flip)
            lh, lw = ytab.shape[0], xtab.shape[0]
            host = torch.from_numpy(np.concatenate([th, tv, ytab, xtab]))
            tabs = host.to(dev, non_blocking=True)
            o_tv = th.shape[0]
            o_y = o_tv + tv.shape[0]
            o_x = o_y + lh
            tmp = torch.empty((H, ow, 3), dtype=torch.uint8, device=dev)
            _native.check(L.semseg_input_resample_h_u8(_p(img), H, W, int(flip), _p(tabs), _p(tabs[ow * 2:]), ksh, _p(tmp), ow,
                                                       _st()), 'input_resample_h_u8')
            _native.check(L.semseg_input_resample_v_normalize(_p(tmp), H, ow, _p(tabs[o_tv:]), _p(tabs[o_tv + oh * 2:]), ksv, oh,
                                                              self._mean_std, _p(img_data[i]), BW, _st()),
                          'input_resample_v_normalize')
            _native.check(L.semseg_input_label_gather(_p(seg), W, _p(tabs[o_y:]), _p(tabs[o_x:]), lh, lw, _p(seg_label[i]),
                                                      BW // s, _st()), 'input_label_gather')
        return {'img_data': img_data.permute(0, 3, 1, 2), 'seg_label': seg_label}


class EvalImageAssembler:
    """Inputs of the evaluation / test loops (dataset.py:210-255, 263-296): ONE image resized (stretched, not padded) to every
    short-side size of `imgSizes`, each target rounded up to a multiple of `padding_constant`, normalised; the label map is
    only shifted by -1.  Same kernels as the training assembler."""

    def __init__(self, imgSizes, imgMaxSize, padding_constant, device='cuda'):
        self.imgSizes = tuple(imgSizes) if isinstance(imgSizes, (list, tuple)) else (imgSizes,)
        self.imgMaxSize, self.padding_constant = imgMaxSize, padding_constant
        self.device = torch.device(device)
        self._mean_std = (_native.c_f * 6)(*MEAN, *STD)

    def target_sizes(self, ori_height, ori_width):
        """dataset.py:225-232: [(target_height, target_width)] per short-side size"""
        out = []
        for this_short_size in self.imgSizes:
            scale = min(this_short_size / float(min(ori_height, ori_width)), self.imgMaxSize / float(max(ori_height, ori_width)))
            th, tw = int(ori_height * scale), int(ori_width * scale)
            out.append((round2nearest_multiple(th, self.padding_constant), round2nearest_multiple(tw, self.padding_constant)))
        return out

    def assemble(self, image, segm=None):
        """image: uint8 [H][W][3]; segm: uint8 [H][W] or None.  Returns {'img_data': [fp32 logical [1,3,th,tw] (NHWC
        memory) per scale], 'seg_label': int64 [1,H,W] (if segm)} on the device."""
        L = _native.lib()
        dev = self.device
        img = image.to(dev, non_blocking=True).contiguous()
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError('expected a uint8 [H,W,3] image')
        _require_cuda(img)
        H, W = int(img.shape[0]), int(img.shape[1])
        outs = []
        for th_, tw_ in self.target_sizes(H, W):
            tabh, ksh = _resample_tables(W, tw_)
            tabv, ksv = _resample_tables(H, th_)
            tabs = torch.from_numpy(np.concatenate([tabh, tabv])).to(dev, non_blocking=True)
            o_v = tabh.shape[0]
            tmp = torch.empty((H, tw_, 3), dtype=torch.uint8, device=dev)
            out = torch.empty((1, th_, tw_, 3), dtype=torch.float32, device=dev)
            _native.check(L.semseg_input_resample_h_u8(_p(img), H, W, 0, _p(tabs), _p(tabs[tw_ * 2:]), ksh, _p(tmp), tw_, _st()),
                          'input_resample_h_u8')
            _native.check(L.semseg_input_resample_v_normalize(_p(tmp), H, tw_, _p(tabs[o_v:]), _p(tabs[o_v + th_ * 2:]), ksv, th_,
                                                              self._mean_std, _p(out), tw_, _st()), 'input_resample_v_normalize')
            outs.append(out.permute(0, 3, 1, 2))
        feed = {'img_data': outs}
        if segm is not None:
            seg = segm.to(dev, non_blocking=True).contiguous()
            if seg.dtype != torch.uint8 or tuple(seg.shape) != (H, W):
                raise ValueError('expected a uint8 [H,W] label map of the image size')
            ident = torch.arange(max(H, W), dtype=torch.int32, device=dev)
            lab = torch.empty((1, H, W), dtype=torch.int64, device=dev)
            _native.check(L.semseg_input_label_gather(_p(seg), W, _p(ident), _p(ident), H, W, _p(lab), W, _st()),
                          'input_label_gather')
            feed['seg_label'] = lab
        return feed


class _EvalBase:
    def __init__(self, odgt, opt, device, max_sample=-1, start_idx=-1, end_idx=-1):
        if isinstance(odgt, list):
            self.list_sample = odgt
        else:
            self.list_sample = [json.loads(x.rstrip()) for x in open(odgt, 'r')]
        if max_sample > 0:
            self.list_sample = self.list_sample[0:max_sample]
        if start_idx >= 0 and end_idx >= 0:
            self.list_sample = self.list_sample[start_idx:end_idx]
        self.num_sample = len(self.list_sample)
        assert self.num_sample > 0
        self.assembler = EvalImageAssembler(opt.imgSizes, opt.imgMaxSize, opt.padding_constant, device)

    def __len__(self):
        return self.num_sample


class ValDataset(_EvalBase):
    """Mirror of the reference `ValDataset` (dataset.py:206-255): img_ori (host uint8), img_data per scale, seg_label, info."""

    def __init__(self, root_dataset, odgt, opt, device='cuda', **kwargs):
        super().__init__(odgt, opt, device, **kwargs)
        self.root_dataset = root_dataset

    def __getitem__(self, index):
        from PIL import Image
        rec = self.list_sample[index]
        img = Image.open(os.path.join(self.root_dataset, rec['fpath_img'])).convert('RGB')
        segm = Image.open(os.path.join(self.root_dataset, rec['fpath_segm']))
        assert segm.mode == 'L'
        assert img.size == segm.size
        arr = np.array(img)
        out = self.assembler.assemble(torch.from_numpy(arr), torch.from_numpy(np.array(segm)))
        out['img_ori'] = arr
        out['info'] = rec['fpath_img']
        return out


class TestDataset(_EvalBase):
    """Mirror of the reference `TestDataset` (dataset.py:258-296)."""

    def __init__(self, odgt, opt, device='cuda', **kwargs):
        super().__init__(odgt, opt, device, **kwargs)

    def __getitem__(self, index):
        from PIL import Image
        rec = self.list_sample[index]
        arr = np.array(Image.open(rec['fpath_img']).convert('RGB'))
        out = self.assembler.assemble(torch.from_numpy(arr))
        out['img_ori'] = arr
        out['info'] = rec['fpath_img']
        return out


class TrainDataset:
    """Mirror of the reference `TrainDataset` (dataset.py:70-203): one item = one per-GPU batch.  Same record grouping
    (portrait / landscape lists, dataset.py:85-108), the same numpy random draws in the same order (shuffle seeded by the
    first index, short-side size, one flip per sample), PIL decode on the host -- and the assembly on the device."""

    def __init__(self, root_dataset, odgt, opt, batch_per_gpu=1, device='cuda', max_sample=-1, start_idx=-1, end_idx=-1):
        if isinstance(odgt, list):
            self.list_sample = odgt
        else:
            self.list_sample = [json.loads(x.rstrip()) for x in open(odgt, 'r')]
        if max_sample > 0:
            self.list_sample = self.list_sample[0:max_sample]
        if start_idx >= 0 and end_idx >= 0:
            self.list_sample = self.list_sample[start_idx:end_idx]
        self.num_sample = len(self.list_sample)
        assert self.num_sample > 0
        self.root_dataset = root_dataset
        self.imgSizes, self.imgMaxSize = opt.imgSizes, opt.imgMaxSize
        self.padding_constant, self.segm_downsampling_rate = opt.padding_constant, opt.segm_downsampling_rate
        self.batch_per_gpu = batch_per_gpu
        self.random_flip = bool(getattr(opt, 'random_flip', True))          # dataset.py:157 draws the flip unconditionally; the
        self.batch_record_list = [[], []]                                   # option exists in the config, so it is honoured
        self.cur_idx = 0
        self.if_shuffled = False
        self.assembler = TrainBatchAssembler(self.imgSizes, self.imgMaxSize, self.padding_constant,
                                             self.segm_downsampling_rate, device)

    def _get_sub_batch(self):
        while True:
            this_sample = self.list_sample[self.cur_idx]
            self.batch_record_list[0 if this_sample['height'] > this_sample['width'] else 1].append(this_sample)
            self.cur_idx += 1
            if self.cur_idx >= self.num_sample:
                self.cur_idx = 0
                np.random.shuffle(self.list_sample)
            for k in (0, 1):
                if len(self.batch_record_list[k]) == self.batch_per_gpu:
                    batch_records = self.batch_record_list[k]
                    self.batch_record_list[k] = []
                    return batch_records

    def plan(self, index):
        """the SEQUENTIAL part of __getitem__ (dataset.py:153-160): record selection and every numpy random draw, in the
        reference's order (shuffle seeded by the first index, short-side size, one flip per sample) -- no file is touched, so a
        planner thread can run ahead while a pool decodes the files of earlier batches in parallel.  Returns (records, flips,
        short size)."""
        if not self.if_shuffled:
            np.random.seed(index)
            np.random.shuffle(self.list_sample)
            self.if_shuffled = True
        batch_records = self._get_sub_batch()
        if isinstance(self.imgSizes, (list, tuple)):
            this_short_size = np.random.choice(self.imgSizes)
        else:
            this_short_size = self.imgSizes
        flips = [bool(np.random.choice([0, 1])) if self.random_flip else False for _ in batch_records]
        return batch_records, flips, this_short_size

    def load_record(self, rec):
        """decode the image / annotation pair of one record (dataset.py:162-169) into uint8 tensors: independent of every other
        record, and Pillow releases the GIL while it decodes -- the unit of work of the decode pool"""
        from PIL import Image
        img = Image.open(os.path.join(self.root_dataset, rec['fpath_img'])).convert('RGB')
        segm = Image.open(os.path.join(self.root_dataset, rec['fpath_segm']))
        assert segm.mode == 'L'
        assert img.size == segm.size
        return torch.from_numpy(np.array(img)), torch.from_numpy(np.array(segm))

    def decode(self, index):
        """host half of __getitem__ (plan + file decode) in one call"""
        batch_records, flips, this_short_size = self.plan(index)
        pairs = [self.load_record(rec) for rec in batch_records]
        return [p[0] for p in pairs], [p[1] for p in pairs], flips, this_short_size

    def assemble(self, decoded):
        """device half: resize / flip / normalise / label down-sampling / padding in HIP kernels"""
        return self.assembler.assemble(*decoded)

    def __getitem__(self, index):
        return self.assemble(self.decode(index))

    def __len__(self):
        return int(1e10)          # dataset.py:201-203: every loader keeps its own list
