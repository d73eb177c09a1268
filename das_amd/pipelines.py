"""Data pipeline pieces: `Compose` and the registry the dataset configs resolve their
`pipeline=[dict(type=...)]` entries through (mmdet.datasets.pipelines.Compose / PIPELINES in the reference)."""
from .datasets import PIPELINES
from .registry import build_from_cfg


class Compose:
    """Chain of transforms; a transform returning None drops the sample (mmdet's Compose semantics)."""

    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data
