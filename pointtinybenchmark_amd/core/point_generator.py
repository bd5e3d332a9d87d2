"""PointGenerator (T/mmdet/core/anchor/point_generator.py:7-37): (x*stride, y*stride, stride), NO half-stride
offset; index generation only (exact integer arithmetic), kept in torch."""
import torch


class PointGenerator:
    def grid_points(self, featmap_size, stride=16, device='cuda'):
        h, w = featmap_size
        sx = torch.arange(0., w, device=device) * stride
        sy = torch.arange(0., h, device=device) * stride
        xx = sx.repeat(h)
        yy = sy.view(-1, 1).repeat(1, w).view(-1)
        return torch.stack([xx, yy, xx.new_full((xx.shape[0],), stride)], dim=-1)

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        h, w = featmap_size
        vh, vw = valid_size
        assert vh <= h and vw <= w
        vx = torch.zeros(w, dtype=torch.bool, device=device)
        vy = torch.zeros(h, dtype=torch.bool, device=device)
        vx[:vw] = 1
        vy[:vh] = 1
        return (vx[None, :] & vy[:, None]).reshape(-1)
