"""Loss modules with the reference's registry names (FocalLoss / SmoothL1Loss / CrossEntropyLoss
from mmdet 2.14.0, RLELoss3D from mmdet3d/models/losses/residual_log_likelihood_loss.py) and the
DASHead loss assembly (das_head.py:281-486).

The modules carry only their hyper-parameters; the arithmetic runs in HIP kernels."""
import math

import torch.nn as nn

from .registry import LOSSES


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid, 'only sigmoid focal loss is supported'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert use_sigmoid and not use_mask and class_weight is None
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight


@LOSSES.register_module()
class RLELoss3D(nn.Module):
    def __init__(self, residual=True, avg_factor=False, loss_weight=1.0, **kwargs):
        super().__init__()
        self.residual, self.avg_factor, self.loss_weight = residual, avg_factor, loss_weight
        self.amp = 1 / math.sqrt(2 * math.pi)


def das_head_loss(head, cls_scores, pose_preds, centernesses, aux_pose_preds, gt_labels_3d, gt_poses_3d, centers2d,
                  depths):
    raise NotImplementedError(
        'DASHead.loss: the HIP training path (target assignment, focal / SmoothL1 / RLE / BCE losses and the '
        'backward kernels) is not built yet in this round; inference + decode are. See DESIGN.md "what comes next".')
