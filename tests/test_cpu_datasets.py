"""CPU: the AirDrone reader (stereotracking_amd/datasets.py; BASELINE configs[4] readiness) - PNG decode through the
native st_png_unfilter, CocoVID -> MOTDispDataset parsing with the reference's instance filter rules
(mmtrack/datasets/mot_disp_dataset.py:38-97), VideoSampler's whole-video split (samplers/video_sampler.py:25-70), the
test-time transforms built from a pipeline config shaped like the reference's (configs/stereo_tracking/ocsort/
yolox_s_mmyolo_mot_airdrone_disp.py:104-116)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from stereotracking_amd import datasets as ds  # noqa: E402


@pytest.mark.parametrize('shape,dtype', [((13, 17), np.uint16), ((9, 11, 3), np.uint8), ((7, 5), np.uint8),
                                         ((6, 10, 4), np.uint8), ((5, 4, 3), np.uint16), ((4, 6, 2), np.uint8)])
def test_png_round_trip_every_filter(tmp_path, shape, dtype):
    rng = np.random.RandomState(hash((shape, str(dtype))) % 2 ** 31)
    hi = 65536 if dtype == np.uint16 else 256
    arr = rng.randint(0, hi, shape).astype(dtype)
    arr.reshape(-1)[:3] = (0, hi - 1, hi // 2)
    for filt in (0, 1, 2, 3, 4, [y % 5 for y in range(shape[0])]):
        p = str(tmp_path / 'a.png')
        ds.write_png(p, arr, filters=filt)
        got = ds.read_png(p)
        assert got.dtype == dtype and got.shape == arr.shape and np.array_equal(got, arr), filt
    with pytest.raises(ValueError):
        ds.read_png(b'not a png at all')


def test_png_unfilter_rejects_bad_filter_type():
    import ctypes as C
    from stereotracking_amd import _lib
    lib = _lib.load()
    raw = bytes([7, 1, 2, 3])                       # filter type 7 does not exist
    out = np.zeros(3, np.uint8)
    assert lib.st_png_unfilter(C.c_char_p(raw), 1, 3, 1, C.c_void_p(out.ctypes.data)) != 0
    assert b'filter type 7' in lib.st_last_error()


@pytest.fixture(scope='module')
def tiny(tmp_path_factory):
    from make_tiny_airdrone import make
    root = str(tmp_path_factory.mktemp('airdrone'))
    base, ann = make(root, videos=3, frames=6, height=48, width=96, max_disp=16, objects=3)
    return base, ann


def test_dataset_paths_instances_and_sampler(tiny):
    base, ann = tiny
    d = ds.MOTDispDataset(ann_file='annotations/val_cocoformat_80.json', data_root=base + os.sep,
                          data_prefix=dict(img_path='val/'), depth_dir_name='depth', metainfo=dict(CLASSES=('drone',)))
    assert len(d) == 18 and [n for n, _ in d.video_indices()] == ['seq00', 'seq01', 'seq02']
    info = d.get_data_info(7)
    assert info['frame_id'] == 1 and info['video_length'] == 6 and info['cat2label'] == {1: 0}
    assert info['img_path'].endswith(os.path.join('val', 'seq01', 'left', '000001.png'))
    assert info['disp_path'].endswith(os.path.join('seq01', 'disparity', '000001.png'))
    assert info['depth_path'].endswith(os.path.join('seq01', 'depth', '000001.png'))
    assert info['right_path'].endswith(os.path.join('seq01', 'right', '000001.png'))
    assert info['img_path'].split(os.sep)[-3] == 'seq01'            # how the metric names the video (:171)
    for ins in info['instances']:
        assert set(ins) == {'ignore_flag', 'instance_id', 'category_id', 'bbox_label', 'bbox', 'location', 'mot_conf',
                            'visibility'} and ins['bbox'][2] > ins['bbox'][0]
    # the reference's instance filter (mot_disp_dataset.py:66-77): ignored / outside / degenerate / foreign category
    coco = json.load(open(ann))
    img = coco['images'][0]
    base_ann = dict(id=10_000, image_id=img['id'], category_id=1, instance_id=99, area=100.0, location=[0, 0, 5.0],
                    mot_conf=1.0, visibility=1.0, iscrowd=False)
    cases = [dict(base_ann, bbox=[5, 5, 10, 10]),                                   # kept
             dict(base_ann, bbox=[5, 5, 10, 10], ignore=True),
             dict(base_ann, bbox=[-30, 5, 10, 10]),                                 # no overlap with the image
             dict(base_ann, bbox=[5, 5, 0.5, 10]),                                  # w < 1
             dict(base_ann, bbox=[5, 5, 10, 10], area=0),
             dict(base_ann, bbox=[5, 5, 10, 10], category_id=2),
             dict(base_ann, bbox=[5, 5, 10, 10], iscrowd=True)]                     # kept, ignore_flag 1
    got = d.parse_data_info(dict(raw_img_info=dict(img, img_id=img['id'], video_length=6), raw_ann_info=cases))
    assert [(i['bbox'], i['ignore_flag']) for i in got['instances']] == [([5, 5, 15, 15], 0), ([5, 5, 15, 15], 1)]
    # VideoSampler: np.array_split of the video list over the ranks, frames of a video stay in order
    s0, s1 = ds.VideoSampler(d, rank=0, world_size=2), ds.VideoSampler(d, rank=1, world_size=2)
    assert [n for n, _ in s0.videos] == ['seq00', 'seq01'] and [n for n, _ in s1.videos] == ['seq02']
    assert list(s0) == list(range(12)) and list(s1) == list(range(12, 18))


def test_pipeline_from_reference_shaped_config(tiny):
    base, ann = tiny
    from stereotracking_amd.sequence import synthetic_sequence
    pipeline = [dict(type='LoadImageFromFile'),
                dict(type='LoadDisparityFromFile', to_3channel=True, post_processing=dict(disp_thr_h=50, disp_thr_l=0)),
                dict(type='Resize_Disparity', scale=(96, 48), keep_ratio=True),
                dict(type='Pad_Disparity', size_divisor=32, pad_val=dict(img=(114.0, 114.0, 114.0), disp=0, disp_mask=0)),
                dict(type='PackTrackInputs_Disparity', pack_single_img=True,
                     meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor'))]
    d = ds.DATASETS.build(dict(type='MOTDispDataset', data_root=base + os.sep,
                               ann_file='annotations/val_cocoformat_80.json', data_prefix=dict(img_path='val/'),
                               depth_dir_name='depth', metainfo=dict(CLASSES=('drone',)), ref_img_sampler=None,
                               load_as_video=True, test_mode=True, pipeline=pipeline))
    item = d[2]
    frame = list(synthetic_sequence(6, 3, 48, 96, 16, seed=0))[2]
    inp, sample = item['inputs'], item['data_samples']
    assert inp['img'].shape == (1, 3, 48, 96) and inp['img'].dtype == torch.uint8
    assert np.array_equal(inp['img'][0].numpy(), frame['left'])                       # BGR, as mmcv yields it
    assert inp['disp_postp'].shape == (1, 3, 48, 96) and inp['disp_postp'].dtype == torch.float32
    codes = inp['disp_codes'][0].numpy().view(np.uint16)
    invalid = codes == 65535
    assert invalid.sum() == 36 and np.array_equal(inp['disp_mask'][0, 0].numpy(), (~invalid).astype(np.uint8))
    want = np.where(invalid, 0.0, frame['disp']).astype(np.float32)
    for c in range(3):
        assert np.array_equal(inp['disp_postp'][0, c].numpy(), want)                 # 65535 -> 0, / 16 (:129-134)
    m = sample.metainfo
    assert m['frame_id'] == 2 and m['video_length'] == 6 and m['ori_shape'] == (48, 96) and m['scale_factor'] == (1.0, 1.0)
    assert len(m['instances']) == len(d.data_list[2]['instances'])
    # a non-identity scale: sizes / scale_factor as mmcv.rescale_size + mmdet Resize record them; the resampling itself is a
    # device pass (st_resize_planes; tests/test_resize_gpu.py) - without a GPU an array to resample is an error, never a
    # silent CPU path
    rz = ds.Resize_Disparity(scale=(640, 360))
    assert rz.new_size(48, 96) == (640, 320) and ds.Resize_Disparity(scale=(640, 360), keep_ratio=False).new_size(48, 96) == (640, 360)
    r = rz(dict(img_shape=(48, 96)))
    assert r['img_shape'] == (320, 640) and r['scale_factor'] == (640 / 96, 320 / 48)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='GPU'):
            rz(dict(img_shape=(48, 96), img=np.zeros((48, 96, 3), np.uint8)))
    # depth: AirSim encoding value / 100 (loading_disparity.py:233)
    r = ds.LoadDepthFromFile()(dict(d.get_data_info(2)))
    assert r['depth'].shape == (48, 96, 1) and abs(float(r['depth'].max()) - 80.0) < 1e-3


def test_resize_oracle_restatement_properties():
    """oracle/resize.py (the numpy restatement the GPU test checks st_resize_planes against) on cases whose answers follow
    from the algorithm: constant images stay constant, the exact 2 x 2 decimation is the rounded box mean, an upscale by an
    integer factor reproduces the source at the sample centres' nearest pixels, a horizontal ramp stays monotone and inside
    the source range, nearest picks floor(dx * src / dst)."""
    from oracle import resize as orz
    assert orz.rescale_size(720, 1280, (640, 360)) == (640, 360) and orz.rescale_size(48, 96, (640, 360)) == (640, 320)
    const = np.full((37, 53, 3), 201, np.uint8)
    assert (orz.resize_bilinear_u8(const, 19, 71) == 201).all()
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (48, 96, 3)).astype(np.uint8)
    half = orz.resize_bilinear_u8(img, 24, 48)
    x = img.astype(np.int64)
    assert np.array_equal(half, ((x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    ramp = np.tile(np.arange(96, dtype=np.uint8)[None, :] * 2, (48, 1))
    up = orz.resize_bilinear_u8(ramp, 100, 250)
    assert up.shape == (100, 250) and (np.diff(up.astype(int), axis=1) >= 0).all() and up.min() == 0 and up.max() == 190
    # rows of a horizontal ramp agree to one code (the two truncating shifts of the vertical pass split b0 + b1 = 2048)
    assert np.abs(up.astype(int) - up[0].astype(int)).max() <= 1
    same = orz.resize_bilinear_u8(img, 48, 96)                    # identity size: fx = 0 everywhere
    assert np.array_equal(same, img)
    codes = rng.randint(0, 65536, (48, 96)).astype(np.uint16)
    nn = orz.resize_nearest(codes, 31, 200)
    for dy, dx in ((0, 0), (30, 199), (7, 101), (15, 3)):
        assert nn[dy, dx] == codes[min(int(np.floor(dy * 48 / 31)), 47), min(int(np.floor(dx * 96 / 200)), 95)]
    assert np.array_equal(orz.resize_nearest(codes, 24, 48), codes[0::2, 0::2])
