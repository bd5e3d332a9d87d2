"""BasicLocator detector (T/mmdet/models/point/detectors/locator.py:6-32 on top of
SingleStageDetector/BaseDetector: T/mmdet/models/detectors/single_stage.py:35-104, base.py:114-247).
The drop-in boundary: ``forward_train(img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes)
-> dict of losses`` and ``simple_test(img, img_metas, rescale, **gt_kwargs)``."""
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from ..registry import DETECTORS, build_backbone, build_head, build_neck


@DETECTORS.register_module()
class BasicLocator(nn.Module):
    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None):
        super().__init__()
        backbone = dict(backbone)
        backbone.setdefault('pretrained', pretrained)  # no checkpoints offline; weights arrive via load_state_dict
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg)
        bbox_head.update(test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    @property
    def with_neck(self):
        return self.neck is not None

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_true_bboxes=None):
        batch_input_shape = tuple(img[0].size()[-2:])
        for m in img_metas:
            m['batch_input_shape'] = batch_input_shape
        x = self.extract_feat(img)
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_true_bboxes)

    def simple_test(self, img, img_metas, rescale=False, **kwargs):
        """Returns the head's refine output: list[(dets (G,6+), labels)] (bbox2result is host-side glue)."""
        feat = self.extract_feat(img)
        return self.bbox_head.simple_test(feat, img_metas, rescale=rescale, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        for k in list(kwargs):  # the fork unwraps the per-aug lists of gt_* kwargs (base.py:147-150)
            if k.startswith('gt_') and isinstance(kwargs[k], (list, tuple)) and len(kwargs[k]) == 1 \
                    and isinstance(kwargs[k][0], (list, tuple)):
                kwargs[k] = kwargs[k][0]
        if isinstance(imgs, (list, tuple)):
            assert len(imgs) == 1, 'aug test is outside the hot path'
            imgs, img_metas = imgs[0], img_metas[0]
        return self.simple_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    @staticmethod
    def _parse_losses(losses):
        """base.py:179-212: sum every key containing 'loss'; log vars averaged over ranks."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError('%s is not a tensor or list of tensors' % name)
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        for name, value in log_vars.items():
            if dist.is_available() and dist.is_initialized():
                value = value.data.clone()
                dist.all_reduce(value.div_(dist.get_world_size()))
            log_vars[name] = value.item()
        return loss, log_vars

    def train_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))
