"""Generate tests/golden/decode_soft.npz: the REFERENCE's `DASHead.get_poses` with `nms_type='soft'`
(das_head.py:784-790 -> pose_nms.py:128-194 soft_oks_nms) at the full 512x832 level sizes.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden_decode_soft.py
Inputs come from tests/golden/cases.py seeds; only the reference's outputs are stored.
"""
import numpy as np
import torch

import make_golden as mg   # (sets sys.path, imports cases / refstub)
from make_golden import cases, refstub


def main():
    R = refstub.load()
    torch.manual_seed(0)
    full = mg.build_full_head(R)
    full.train(False)
    arrs = {}
    for tag, cfg in cases.SOFT_DECODE_CFGS.items():
        cls, pose, ctr = cases.full_decode_inputs(**cases.SOFT_DECODE_INPUTS[tag])
        metas = [dict(scale_factor=np.array([1.3, 1.3, 1.3, 1.3], dtype=np.float32), filename='a'),
                 dict(scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), filename='b')]
        with torch.no_grad():
            res = full.get_poses([o.clone() for o in cls], [o.clone() for o in pose], [o.clone() for o in ctr], metas,
                                 cfg=cfg)
        for b, r in enumerate(res):
            arrs[f'{tag}_poses{b}'], arrs[f'{tag}_centers{b}'] = r['poses'], r['centers']
            arrs[f'{tag}_scores{b}'] = np.array(r['scores'], dtype=np.float32)
    mg.save('decode_soft', **arrs)


if __name__ == '__main__':
    main()
