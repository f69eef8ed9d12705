"""GPU: Resize_Disparity with a non-identity scale (reference mmtrack/datasets/transforms/transforms_disparity.py:23-137) -
the HIP resampler st_resize_planes (csrc/pack_pool.hip) BIT-EXACT against the numpy restatement of OpenCV's 8-bit
INTER_LINEAR / INTER_NEAREST (oracle/resize.py), through the C ABI and through the registered transform."""
import numpy as np
import pytest
import torch

from oracle import resize as orz
from stereotracking_amd import datasets as ds
from stereotracking_amd._lib import check, load, ptr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('h,w,h2,w2', [(720, 1280, 360, 640),      # the exact 2 x 2 decimation (box mean)
                                       (720, 1280, 540, 960),      # 0.75: generic down-scaling
                                       (48, 96, 320, 640),         # up-scaling, borders replicate
                                       (37, 53, 19, 71),           # odd sizes, mixed directions
                                       (33, 65, 33, 65)])          # identity size through the kernel
def test_resize_planes_bit_exact_against_the_restatement(h, w, h2, w2, cuda):
    lib = load()
    rng = np.random.RandomState(h + w2)
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    src = torch.from_numpy(img).to(cuda)
    # interleaved (h, w, 3), as the decoder yields frames
    dst = torch.empty(h2, w2, 3, dtype=torch.uint8, device=cuda)
    check(lib.st_resize_planes(ptr(src), 3, h, w, 1, ptr(dst), h2, w2, 1, 1, None), 'st_resize_planes')
    want = orz.resize_bilinear_u8(img, h2, w2)
    assert np.array_equal(dst.cpu().numpy(), want)
    # planar (3, h, w), as the sequence drivers keep them: the same values plane by plane
    srcp = src.permute(2, 0, 1).contiguous()
    dstp = torch.empty(3, h2, w2, dtype=torch.uint8, device=cuda)
    check(lib.st_resize_planes(ptr(srcp), 3, h, w, 0, ptr(dstp), h2, w2, 1, 1, None), 'st_resize_planes')
    assert np.array_equal(dstp.cpu().numpy(), want.transpose(2, 0, 1))
    # nearest: uint16 codes (2 bytes), fp32 maps (4 bytes), uint8 masks (1 byte)
    codes = rng.randint(0, 65536, (h, w)).astype(np.uint16)
    for arr in (codes, (codes / 16.0).astype(np.float32), (codes < 60000).astype(np.uint8)):
        t = torch.from_numpy(arr.view(np.int16) if arr.dtype == np.uint16 else arr).to(cuda)
        o = torch.empty(h2, w2, dtype=t.dtype, device=cuda)
        check(lib.st_resize_planes(ptr(t), 1, h, w, 0, ptr(o), h2, w2, t.element_size(), 0, None), 'st_resize_planes')
        got = o.cpu().numpy()
        assert np.array_equal(got.view(arr.dtype) if arr.dtype == np.uint16 else got, orz.resize_nearest(arr, h2, w2))


def test_resize_planes_rejects_bad_arguments(cuda):
    lib = load()
    t = torch.zeros(8, 8, dtype=torch.float32, device=cuda)
    assert lib.st_resize_planes(ptr(t), 1, 8, 8, 0, ptr(t), 4, 4, 4, 1, None) != 0       # bilinear is the 8-bit path
    assert lib.st_resize_planes(ptr(t), 1, 8, 8, 0, ptr(t), 4, 4, 3, 0, None) != 0       # element size
    assert lib.st_resize_planes(None, 1, 8, 8, 0, ptr(t), 4, 4, 4, 0, None) != 0


def test_resize_disparity_transform_non_identity_scale(cuda):
    """The registered transform on a decoded sample: 720 x 1280 -> scale (960, 540): img / right bilinear, disp_postp /
    codes / mask / depth nearest, img_shape and scale_factor recorded (mmdet Resize), equal to the restatement."""
    rng = np.random.RandomState(5)
    h, w = 720, 1280
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    right = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    codes = rng.randint(0, 3000, (h, w)).astype(np.uint16)
    codes[rng.rand(h, w) < 0.01] = 65535
    disp = np.repeat((np.where(codes == 65535, 0, codes) / 16.0).astype(np.float32)[:, :, None], 3, axis=2)
    mask = (codes < 65535).astype(np.uint8)
    depth = rng.uniform(1, 80, (h, w, 1)).astype(np.float32)
    res = dict(img=img, right=right, disp_codes=codes, disp_postp=disp, disp_mask=mask, depth=depth, depth_postp=depth,
               img_shape=(h, w), ori_shape=(h, w))
    out = ds.TRANSFORMS.build(dict(type='Resize_Disparity', scale=(960, 540), keep_ratio=True))(dict(res))
    assert out['img_shape'] == (540, 960) and out['scale_factor'] == (0.75, 0.75) and out['ori_shape'] == (h, w)
    assert np.array_equal(out['img'], orz.resize_bilinear_u8(img, 540, 960))
    assert np.array_equal(out['right'], orz.resize_bilinear_u8(right, 540, 960))
    assert out['disp_codes'].dtype == np.uint16 and np.array_equal(out['disp_codes'], orz.resize_nearest(codes, 540, 960))
    assert np.array_equal(out['disp_postp'], orz.resize_nearest(disp, 540, 960))
    assert np.array_equal(out['disp_mask'], orz.resize_nearest(mask, 540, 960))
    assert np.array_equal(out['depth_postp'], orz.resize_nearest(depth, 540, 960)) and out['depth'] is out['depth_postp']
    # nearest sampling commutes with the code -> px conversion (why the device path may resize the raw codes)
    assert np.array_equal(out['disp_postp'][..., 0], (np.where(out['disp_codes'] == 65535, 0, out['disp_codes']) / 16.0).astype(np.float32))
    # the identity scale touches nothing
    same = ds.Resize_Disparity(scale=(1280, 720))(dict(res))
    assert same['img'] is img and same['scale_factor'] == (1.0, 1.0)
