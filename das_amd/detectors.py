"""DAS detector (reference: mmdet3d/models/detectors/das.py:5-39 on top of mmdet's
SingleStageDetector / BaseDetector protocol): backbone -> neck -> bbox_head, `forward_train`
returning the loss dict, `simple_test` returning per-image pose dicts, `train_step` /
`forward(return_loss=...)` as the runner calls them (tools/train.py, mmdet3d/apis/test.py:39)."""
from collections import OrderedDict
from collections.abc import Mapping

import torch
import torch.distributed as dist
import torch.nn as nn

from .registry import DETECTORS, build_backbone, build_head, build_neck


# A/B switch (bench.py --no-early-targets): compute the loss's ground-truth half beside the backbone (side stream)
EARLY_TARGETS = True
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class LazyLogVars(Mapping):
    """The step's log variables (names + one stacked tensor, already averaged over the ranks) as a read-only mapping
    name -> float that does not stall the step: the device-to-host copy is queued at once into page-locked memory and
    the host only waits for it when a value is READ (mmdet's `_parse_losses` reads every value back right after the
    forward pass — a host synchronisation per step; the logger looks at them every `log_config.interval` steps)."""

    # Page-locked landing buffers, reused round-robin: allocating one (hipHostMalloc) waits for the device to drain — with the host
    # three steps ahead of the GPU that is a 100 ms stall, and torch's caching host allocator allocates a NEW block whenever the
    # previous step's has not been released by the GPU yet, i.e. in every one of the first steps (seen as single 135-146 ms steps
    # in a run with three warm-up steps). A slot is taken again when its previous owner is gone or has been read.
    _RING, _NEXT = [], [0]

    @classmethod
    def _landing(cls, vals):
        n = 8
        while len(cls._RING) < n:
            cls._RING.append([None, None])
        for _ in range(n):
            slot = cls._RING[cls._NEXT[0] % n]
            cls._NEXT[0] += 1
            owner = slot[1]() if slot[1] is not None else None
            if owner is not None and owner._dict is None:
                if owner._done is not None and not owner._done.query():
                    continue                      # (still in flight: leave it alone)
                owner.resolve()                   # (its copy has landed: read it out, the buffer is free)
            if slot[0] is None or slot[0].shape != vals.shape or slot[0].dtype != vals.dtype:
                slot[0] = torch.empty(vals.shape, dtype=vals.dtype, pin_memory=True)
            return slot
        return [torch.empty(vals.shape, dtype=vals.dtype, pin_memory=True), None]

    def __init__(self, names, vals):
        import weakref
        self.names = list(names)
        self._dict = None
        if vals.is_cuda:
            slot = self._landing(vals)
            slot[1] = weakref.ref(self)
            self._host = slot[0]
            self._host.copy_(vals, non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record()
        else:
            self._host, self._done = vals, None

    def resolve(self):
        if self._dict is None:
            if self._done is not None:
                self._done.synchronize()
            self._dict = OrderedDict(zip(self.names, self._host.tolist()))
        return self._dict

    def __getitem__(self, k):
        return self.resolve()[k]

    def __iter__(self):
        return iter(self.names)

    def __len__(self):
        return len(self.names)

    def __repr__(self):
        return repr(self.resolve())


@DETECTORS.register_module()
class DAS(nn.Module):
    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        backbone = dict(backbone)
        if pretrained is not None:
            backbone['pretrained'] = pretrained
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        # the neck starts at level >= 1 (every DAS config): the backbone's finest map has no consumer — it is not computed
        # (eval) / reduced to its BatchNorm statistics (train); das_amd/backbones.py UpsampleUnit.forward_unused
        if self.neck is not None and getattr(self.neck, 'start_level', 0) >= 1 and hasattr(self.backbone, 'skip_unused_finest'):
            self.backbone.skip_unused_finest = True
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.CLASSES = ('person',)

    @property
    def with_neck(self):
        return self.neck is not None

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_neck:
            self.neck.init_weights()
        self.bbox_head.init_weights()

    def set_compute_dtype(self, dtype):
        """bf16 (default, the benchmarked precision) or f32 (exact-f32 MFMA path, parity runs)."""
        self.backbone.compute_dtype = dtype
        return self

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_poses_3d, gt_labels_3d, centers2d, depths,
                      gt_bboxes_ignore=None):
        # The ground-truth half of the loss (target assignment, positive rows, counts) needs one device-to-host copy. It
        # runs on a side stream AFTER the backbone has been queued: the copy waits for those few small kernels only, the
        # main stream keeps executing the backbone meanwhile, and everything after it (head, losses) is queued without
        # another synchronisation.
        hw = tuple(img.shape[-2:])
        prepare = getattr(self.bbox_head, 'prepare_targets', None) if EARLY_TARGETS else None
        targets = None
        if prepare and img.is_cuda:
            main = torch.cuda.current_stream()
            from .datasets import uploaded_events
            # the ground truth's own upload events (datasets.mark_uploaded) if it carries them: the side stream then waits for
            # those copies only. Without them: an event recorded now — behind everything the training stream has queued, so
            # the host's read-back below waits for the previous step to finish (host and GPU in lockstep).
            ready = uploaded_events(gt_poses_3d, centers2d, depths)
            if ready is None:
                ready = [torch.cuda.Event()]
                ready[0].record(main)
        if hasattr(self.bbox_head, 'prefetch_scales'):
            self.bbox_head.prefetch_scales()    # (a device-to-host copy the head needs: started before the backbone is queued)
        # backbone + neck: two hipGraph replays when the trunk was captured for this batch shape (das_amd/graphs.py)
        trunk = getattr(self, '_graphed_trunk', None)
        if trunk is not None and self.training and torch.is_grad_enabled() and trunk.matches(img):
            x = trunk(img)
        else:
            x = self.extract_feat(img)
        if prepare and img.is_cuda:
            side = _side_stream(img.device)
            for ev in ready:
                side.wait_event(ev)
            for t in list(gt_poses_3d) + list(centers2d or ()) + list(depths or ()):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                targets = prepare(hw, img.shape[0], img.device, gt_poses_3d, centers2d, depths)
            main.wait_stream(side)
            for t in (targets or {}).values():
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
        extra = dict(targets=targets, input_hw=hw) if prepare else {}
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_poses_3d, gt_labels_3d, centers2d,
                                            depths, gt_bboxes_ignore, **extra)

    def simple_test(self, img, img_metas, rescale=False, **kwargs):
        # backbone + neck + head: one hipGraph replay when the forward was captured for this input shape and these
        # parameters (das_amd/graphs.py enable_inference_graph), else launch by launch
        g = getattr(self, '_graphed_infer', None)
        if g is not None and torch.is_tensor(img) and g.matches(img):
            outs = g(img)
        else:
            x = self.extract_feat(img)
            outs = self.bbox_head(x)
        return self.bbox_head.get_poses(*outs, img_metas, rescale=rescale)

    def aug_test(self, imgs, img_metas, rescale=False):
        raise NotImplementedError

    def forward_test(self, imgs, img_metas, **kwargs):
        """mmdet BaseDetector.forward_test: lists with one entry per test-time augmentation."""
        if not isinstance(imgs, (list, tuple)):
            imgs, img_metas = [imgs], [img_metas]
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        with torch.no_grad():
            return self.forward_test(img, img_metas, **kwargs)

    @staticmethod
    def _parse_losses(losses, lazy=False):
        """Sum of every entry whose key contains 'loss'; log vars averaged over ranks.
        lazy: log_vars come back as a LazyLogVars — the device-to-host copy (a host synchronisation that drains the
        launch queue in the middle of the step, right before backward has to be queued) happens when it is resolved."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            # (the mean of a 0-dim tensor is the tensor — what the head's losses are: no reduce launch, no mean backward)
            if isinstance(value, torch.Tensor):
                log_vars[name] = value if value.dim() == 0 else value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v if v.dim() == 0 else v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        terms = [v for k, v in log_vars.items() if 'loss' in k]
        loss = sum(terms[1:], terms[0]) if terms else 0      # (0 + v is v: the same sum without its first add)
        log_vars['loss'] = loss
        # mmdet averages every log var over the ranks and reads it back one by one (a collective and a host
        # sync per entry); same values here from ONE stacked all-reduce and ONE device-to-host copy
        vals = torch.stack([v.detach().float().reshape(()) for v in log_vars.values()])
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(vals.div_(dist.get_world_size()))
        if lazy:
            return loss, LazyLogVars(list(log_vars), vals)
        for name, v in zip(list(log_vars), vals.tolist()):
            log_vars[name] = v
        return loss, log_vars

    def train_step(self, data, optimizer=None, lazy_log=False):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses, lazy=lazy_log)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def val_step(self, data, optimizer=None):
        return self.train_step(data, optimizer)
