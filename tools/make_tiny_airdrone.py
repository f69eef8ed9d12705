#!/usr/bin/env python
"""Synthesise a TINY on-disk dataset in the AirDrone layout (reference README.md:79-102, configs/stereo_tracking/ocsort/
yolox_s_mmyolo_mot_airdrone_disp.py:5,143-146) so that the configs[4] code path - CocoVID json -> MOTDispDataset ->
PNG decode -> raw-byte upload -> dense path -> tracker -> MOTDroneMetrics - can run without the real data:

    <root>/AirSim_drone/val/<seq>/{left,right,disparity,depth}/%06d.png     uint8 RGB / uint16 gray PNGs
    <root>/AirSim_drone/val/<seq>/gt/gt.txt                                 frame,id,x,y,w,h,conf,X,Y,Z
    <root>/AirSim_drone/annotations/val_cocoformat_80.json                  the converter's output format
                                                                            (tools/dataset_converters/AirSim_drone/convertAnnToCocoFormat.py:49-191)
Encodings: disparity PNG = 16 * px, 65535 = invalid (loading_disparity.py:82,129-134); depth PNG = 100 * metres (:233).
The PNG writer cycles through all five scanline filters, so reading the dataset exercises the whole decoder.
usage: python tools/make_tiny_airdrone.py <root> [--videos 3 --frames 12 --height 96 --width 160]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(root, videos=3, frames=12, height=96, width=160, max_disp=32, objects=3, seed=0, distance_thr=80.0,
         area_thr=30.0):
    from stereotracking_amd.datasets import write_png
    from stereotracking_amd.sequence import synthetic_sequence
    base = os.path.join(root, 'AirSim_drone')
    os.makedirs(os.path.join(base, 'annotations'), exist_ok=True)
    out = dict(categories=[dict(id=1, name='drone')], videos=[], images=[], annotations=[])
    vid_id, img_id, ann_id, ins_id = 1, 1, 1, 0
    for v in range(videos):
        name = f'seq{v:02d}'
        vdir = os.path.join(base, 'val', name)
        for d in ('left', 'right', 'disparity', 'depth', 'gt'):
            os.makedirs(os.path.join(vdir, d), exist_ok=True)
        out['videos'].append(dict(id=vid_id, name=name, fps=60, width=width, height=height))
        ins_map, gt_lines = {}, []
        rng = np.random.RandomState(1000 + seed + v)
        for t, f in enumerate(synthetic_sequence(frames, objects, height, width, max_disp, seed=seed + v)):
            fn = f'{t:06d}.png'
            filt = [(y + t) % 5 for y in range(height)]
            write_png(os.path.join(vdir, 'left', fn), f['left'][::-1].transpose(1, 2, 0), filters=filt)    # BGR -> RGB
            write_png(os.path.join(vdir, 'right', fn), f['right'][::-1].transpose(1, 2, 0), filters=filt)
            codes = (f['disp'] * 16.0).astype(np.uint16)
            y0, x0 = rng.randint(0, height - 8), rng.randint(0, width - 8)
            codes[y0:y0 + 6, x0:x0 + 6] = 65535                                      # an invalid patch
            write_png(os.path.join(vdir, 'disparity', fn), codes, filters=filt)
            depth_m = 0.25 * 640.0 / np.maximum(f['disp'], 1e-3)
            write_png(os.path.join(vdir, 'depth', fn), np.clip(np.rint(depth_m * 100.0), 0, 65535).astype(np.uint16),
                      filters=filt)
            out['images'].append(dict(id=img_id, video_id=vid_id, file_name=os.path.join(name, 'left', fn),
                                      height=height, width=width, frame_id=t, mot_frame_ids=t + 1))
            for k, x1, y1, x2, y2, z in f['gt']:
                bbox = [float(x1), float(y1), float(x2 - x1), float(y2 - y1)]
                loc = [0.0, 0.0, float(z)]
                gt_lines.append('%d,%d,%.1f,%.1f,%.1f,%.1f,1,%.3f,%.3f,%.3f' % (t + 1, int(k), *bbox, *loc))
                if bbox[2] * bbox[3] < area_thr or loc[2] > distance_thr:            # converter :65-67
                    continue
                if int(k) not in ins_map:
                    ins_map[int(k)] = ins_id
                    ins_id += 1
                out['annotations'].append(dict(id=ann_id, image_id=img_id, category_id=1, bbox=bbox,
                                               area=bbox[2] * bbox[3], depth=loc[2], location=loc, iscrowd=False,
                                               visibility=1.0, mot_instance_id=int(k), mot_conf=1.0,
                                               instance_id=ins_map[int(k)]))
                ann_id += 1
            img_id += 1
        with open(os.path.join(vdir, 'gt', 'gt.txt'), 'w') as fh:
            fh.write('\n'.join(gt_lines) + '\n')
        vid_id += 1
    out['num_instances'] = ins_id
    ann = os.path.join(base, 'annotations', f'val_cocoformat_{int(distance_thr)}.json')
    with open(ann, 'w') as fh:
        json.dump(out, fh)
    return base, ann


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('root')
    ap.add_argument('--videos', type=int, default=3)
    ap.add_argument('--frames', type=int, default=12)
    ap.add_argument('--height', type=int, default=96)
    ap.add_argument('--width', type=int, default=160)
    a = ap.parse_args()
    print(make(a.root, a.videos, a.frames, a.height, a.width))
