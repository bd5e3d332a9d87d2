"""Device tail of the train / test image pipeline of the CPR configs
(configs2/TinyPersonV2/coarsepointv2/coarse_point_refine_base_TinyPersonV2_640.py:17-50):

    Resize(scale_factor=1.0, keep_ratio=True)   pixels: identity at the shipped scale (asserted); boxes: clipped to the image
                                                (bbox_clip_border=True, transforms.py:241-249) -- same kernel as the flip
    RandomFlip(flip_ratio)                      decision on the host, pixels + boxes flipped on the device
    Normalize(mean, std, to_rgb) -> Pad(size_divisor) -> DefaultFormatBundle -> collate
                                                ONE kernel: uint8 HWC -> (N,Hp,Wp,4) fp32 channels-last (cpr_preprocess_u8)
    Collect(keys=...)                           same keys / img_metas entries as the reference hands to forward_train

The result's ``img`` is an NCHW-shaped channels-last view with 4 channels (4th = 0): ResNet.forward consumes it without a
layout pass.  Decoded images of different sizes are batched by padding to the largest (mmcv collate pads to the batch
maximum as well)."""
import numpy as np
import torch

from .. import _lib, ops


def pil_bgr_loader(path):
    """uint8 HxWx3 BGR like mmcv.imread(cv2 backend) hands to the pipeline (PIL decodes RGB)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])


class GpuImagePipeline:
    def __init__(self, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True, size_divisor=32,
                 flip_ratio=0.0, scale_factor=1.0, device='cuda',
                 keys=('img', 'gt_bboxes', 'gt_labels', 'gt_bboxes_ignore', 'gt_true_bboxes'), bbox_clip_border=True):
        assert float(scale_factor) == 1.0, 'the shipped CPR/P2P configs resize with scale_factor=[1.0]'
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        # mmcv.imnormalize_: stdinv = 1 / np.float64(std), applied to a float32 image (cv2 converts the scalar to float)
        self.stdinv = (1.0 / np.float64(self.std)).astype(np.float32)
        self.to_rgb, self.size_divisor, self.flip_ratio = bool(to_rgb), int(size_divisor), float(flip_ratio)
        self.device, self.keys = device, tuple(keys)
        self.bbox_clip_border = bool(bbox_clip_border)       # Resize's default (transforms.py:66)

    def __call__(self, samples, rng=None):
        """samples: list of dicts with ``img`` (uint8 HxWx3 BGR, numpy or torch) and the gt_* numpy fields."""
        n = len(samples)
        rng = rng or np.random
        flips = np.array([1 if (self.flip_ratio > 0 and rng.rand() < self.flip_ratio) else 0 for _ in range(n)],
                         dtype=np.int32)
        shapes = [tuple(s['img'].shape) for s in samples]
        H, W = max(s[0] for s in shapes), max(s[1] for s in shapes)
        d = self.size_divisor
        Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
        dev = self.device
        out = torch.empty((n, Hp, Wp, 4), device=dev, dtype=torch.float32)
        uniform = all(s == shapes[0] for s in shapes)
        if uniform:
            stack = torch.from_numpy(np.stack([np.asarray(s['img']) for s in samples])).to(dev)
            self._launch(stack, torch.from_numpy(flips).to(dev), out, n, H, W, Hp, Wp)
        else:
            for i, s in enumerate(samples):       # ragged batch: one launch per image into its slot
                h, w = shapes[i][:2]
                im = torch.from_numpy(np.ascontiguousarray(s['img']))[None].to(dev)
                self._launch(im, torch.from_numpy(flips[i:i + 1]).to(dev), out[i:i + 1], 1, h, w, Hp, Wp)
        metas = []
        for i, s in enumerate(samples):
            h, w = shapes[i][:2]
            ph, pw = (h + d - 1) // d * d, (w + d - 1) // d * d
            metas.append(dict(filename=s.get('filename'), ori_filename=s.get('ori_filename'),
                              ori_shape=s.get('ori_shape', shapes[i]), img_shape=(h, w, 3), pad_shape=(ph, pw, 3),
                              scale_factor=np.array([1.0, 1.0, 1.0, 1.0], dtype=np.float32), flip=bool(flips[i]),
                              flip_direction='horizontal' if flips[i] else None,
                              img_norm_cfg=dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)))
        batch = dict(img=ops.as_nchw(out), img_metas=metas)
        img_hw = torch.tensor([[s[0], s[1]] for s in shapes], dtype=torch.int32, device=dev).reshape(-1)
        flips_d = torch.from_numpy(flips).to(dev)
        for key in ('gt_bboxes', 'gt_bboxes_ignore', 'gt_true_bboxes'):   # = the pipeline's bbox_fields (loading.py:246-278)
            if key in self.keys and all(key in s for s in samples):
                batch[key] = self._boxes([s[key] for s in samples], flips_d, img_hw)
        for key in ('gt_labels', 'gt_anns_id'):
            if key in self.keys and all(key in s for s in samples):
                counts = [len(s[key]) for s in samples]          # one host->device copy, per-image views (cat_rows re-joins them)
                flat = np.concatenate([np.asarray(s[key], dtype=np.int64).reshape(-1) for s in samples]) if sum(counts) else \
                    np.zeros((0,), np.int64)
                batch[key] = list(torch.split(torch.from_numpy(flat).to(dev), counts))
        return batch

    def _launch(self, img_u8, flips, out, n, H, W, Hp, Wp):
        assert img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and img_u8.shape[-1] == 3
        import ctypes
        m = (ctypes.c_float * 3)(*self.mean.tolist())
        s = (ctypes.c_float * 3)(*self.stdinv.tolist())
        _lib.call('cpr_preprocess_u8', ops._ptr(img_u8), ops._ptr(flips), ctypes.cast(m, ctypes.c_void_p),
                  ctypes.cast(s, ctypes.c_void_p), int(self.to_rgb), ops._ptr(out), n, H, W, Hp, Wp, ops._stream())

    def _boxes(self, per_img, flips_d, img_hw):
        counts = [len(b) for b in per_img]
        flat = np.concatenate([np.asarray(b, dtype=np.float32).reshape(-1, 4) for b in per_img]) if sum(counts) else \
            np.zeros((0, 4), np.float32)
        t = torch.from_numpy(np.ascontiguousarray(flat)).to(self.device)
        if len(flat):
            img_of = torch.from_numpy(np.repeat(np.arange(len(counts), dtype=np.int32), counts)).to(self.device)
            _lib.call('cpr_clip_flip_boxes', ops._ptr(t), ops._ptr(img_of), ops._ptr(flips_d), ops._ptr(img_hw), len(flat),
                      int(self.bbox_clip_border), ops._stream())
        return list(torch.split(t, counts))
