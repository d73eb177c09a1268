#!/usr/bin/env python
"""MuCo-3DHP annotation file -> the training subsets `MuCo3DHPDataset` reads (the reference's
mytools/muco2coco.py): <root>/annotations/train_{unaug,aug,all}_interv{1,2}.json, every annotation in category 1.

    python tools/convert_muco.py --root data/muco
"""
import argparse
import json
import os


def subsets(db):
    for a in db['annotations']:
        a['category_id'] = 1
    groups = {'_unaug': [im for im in db['images'] if im['file_name'].startswith('unaugmented')],
              '_aug': [im for im in db['images'] if im['file_name'].startswith('augmented')], '_all': db['images']}
    for suffix, imgs in groups.items():
        for step in (1, 2):
            keep = imgs[::step]
            ids = {im['id'] for im in keep}
            yield f'train{suffix}_interv{step}', dict(
                images=keep, annotations=[a for a in db['annotations'] if a['image_id'] in ids],
                categories=[dict(supercategory='person', id=1, name='person')])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', default='data/muco')
    args = ap.parse_args()
    with open(os.path.join(args.root, 'annotations/MuCo-3DHP.json')) as f:
        db = json.load(f)
    for name, sub in subsets(db):
        path = os.path.join(args.root, 'annotations', name + '.json')
        print(path, len(sub['images']))
        with open(path, 'w') as f:
            json.dump(sub, f)


if __name__ == '__main__':
    main()
