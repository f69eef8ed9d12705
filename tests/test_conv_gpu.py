"""GPU parity of the conv primitive (st_conv2d_nhwc) against plain PyTorch CPU convolution.

This is the one floating-point GEMM-shaped kernel of the path, so (per the tier rules) it keeps a
torch fp32/fp64 reference: tolerance 1e-4 relative to the fp64 result's scale — two fp32
summation orders of K <= 4608 products differ by ~sqrt(K)*2^-24.
"""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from stereotracking_amd import _lib
from stereotracking_amd._lib import StConvDesc, check, ptr

pytestmark = pytest.mark.gpu


def pack(w, bias=None, bn=None, eps=1e-3):
    lib = _lib.load()
    cout, cin, kh, kw = w.shape
    nf = lib.st_conv_packed_floats(cout, cin, kh, kw)
    wp = torch.empty(nf, dtype=torch.float32)
    bp = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32)
    g = [ptr(t.contiguous()) if t is not None else None for t in (bn or (None,) * 4)]
    check(lib.st_conv_pack_weights(ptr(w.contiguous()), ptr(bias) if bias is not None else None, g[0], g[1], g[2],
                                   g[3], eps, cout, cin, kh, kw, ptr(wp), ptr(bp)))
    return wp, bp


def run_conv(x_nchw, w, bias, stride, pad, act, dev, variant=-1, res=None, post_scale=1.0, split=None, up=False,
             in_ld=None, in_off=0):
    lib = _lib.load()
    N, Cin, Hi, Wi = x_nchw.shape
    Cout, _, KH, KW = w.shape
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    in_ld = in_ld or Cin
    xin = torch.randn(N, Hi, Wi, in_ld) * 3.0  # garbage in the unused channels
    xin[..., in_off:in_off + Cin] = x_nchw.permute(0, 2, 3, 1)
    xin = xin.contiguous().to(dev)
    wp, bp = pack(w, bias)
    wp, bp = wp.to(dev), bp.to(dev)
    d = StConvDesc()
    d.in_dev = xin.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, Hi, Wi, Cin, in_ld, in_off
    d.wgt_dev = wp.data_ptr(); d.bias_dev = bp.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, KH, KW, stride, pad
    wino = None
    if variant in (43, 44, 57):   # Winograd instances: the same folded weights in transformed, fragment-ordered form
        wino = torch.empty(lib.st_wino_packed_floats(Cout, Cin), dtype=torch.float32)
        wp_host = wp.cpu()   # keep alive: ptr() does not hold a reference
        check(lib.st_wino_pack_weights(ptr(wp_host), Cout, Cin, ptr(wino)), 'st_wino_pack_weights')
        wino = wino.to(dev)
        d.wgt_wino_dev = wino.data_ptr()
    s = Cout if split is None else split
    out1 = torch.full((N, Ho, Wo, s + 4), -777.0, device=dev)  # ld = s+4, off = 4
    d.out1_dev = out1.data_ptr(); d.out1_ld, d.out1_off, d.split = s + 4, 4, s
    out2 = None
    if split is not None:
        out2 = torch.full((N, Ho, Wo, Cout - s), -777.0, device=dev)
        d.out2_dev = out2.data_ptr(); d.out2_ld, d.out2_off = Cout - s, 0
    upb = None
    if up:
        upb = torch.full((N, 2 * Ho, 2 * Wo, Cout), -777.0, device=dev)
        d.up_dev = upb.data_ptr(); d.up_ld, d.up_off = Cout, 0
    resb = None
    if res is not None:
        resb = res.permute(0, 2, 3, 1).contiguous().to(dev)
        d.res_dev = resb.data_ptr(); d.res_ld, d.res_off = Cout, 0
    d.post_scale = post_scale
    d.act = act
    stream = _lib.current_stream()
    if variant >= 0:
        check(lib.st_conv2d_nhwc_variant(C.byref(d), stream, variant), 'conv')
    else:
        check(lib.st_conv2d_nhwc(C.byref(d), stream), 'conv')
    torch.cuda.synchronize()
    o1 = out1.cpu()
    assert torch.all(o1[..., :4] == -777.0), 'kernel wrote outside its channel slice'
    full = o1[..., 4:]
    if out2 is not None:
        full = torch.cat([full, out2.cpu()], dim=-1)
    return full.permute(0, 3, 1, 2), (upb.cpu().permute(0, 3, 1, 2) if upb is not None else None)


def ref_conv(x, w, bias, stride, pad, act, res=None, post_scale=1.0):
    y = F.conv2d(x.double(), w.double(), bias.double() if bias is not None else None, stride, pad)
    if act:
        y = F.silu(y)
    if res is not None:
        y = (y + res.double()) * post_scale
    return y


def assert_close(got, ref, tol=1e-4):
    scale = ref.abs().max().item() + 1e-6
    err = (got.double() - ref).abs().max().item()
    assert err <= tol * scale, f'max abs err {err:.3e} vs scale {scale:.3e}'


CASES = [
    # N, Cin, H, W, Cout, k, stride  (shapes of the path at reduced resolution + ragged sizes)
    (2, 12, 20, 36, 32, 3, 1),     # stem conv on focus-packed input (Cin=12: K=108 crosses taps inside a chunk)
    (1, 32, 23, 41, 64, 3, 2),     # stride-2 stage conv, odd sizes
    (2, 64, 17, 19, 64, 1, 1),     # CSP 1x1
    (1, 128, 9, 13, 256, 3, 1),    # head tower
    (1, 512, 6, 10, 512, 1, 1),    # SPP conv2-like (K=512..)
    (1, 24, 10, 12, 48, 3, 2),     # tiny-config channels (Cin not a multiple of 32)
    (1, 128, 7, 9, 5, 1, 1),       # prediction conv, Cout=5 (padded to 32)
    (3, 48, 5, 7, 48, 3, 1),       # cost-volume aggregation shape
]


@pytest.mark.parametrize('case', CASES)
def test_conv_matches_torch(case, cuda):
    torch.manual_seed(sum(case))
    N, Cin, H, W, Cout, k, stride = case
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout)
    got, _ = run_conv(x, w, b, stride, k // 2, 1, cuda)
    assert_close(got, ref_conv(x, w, b, stride, k // 2, 1))


@pytest.mark.parametrize('variant,cout', [(0, 128), (1, 64), (2, 32), (3, 64), (4, 32), (0, 256), (3, 128), (5, 64),
                                          (6, 32), (7, 128), (8, 64), (5, 192), (6, 96), (9, 64), (10, 32), (11, 64),
                                          (9, 128), (10, 96), (12, 128), (13, 64), (14, 128), (15, 32), (16, 64),
                                          (17, 256), (18, 192), (19, 128), (20, 256), (21, 128)])
def test_conv_all_tile_variants(variant, cout, cuda):
    torch.manual_seed(variant)
    x = torch.randn(2, 64, 13, 21)
    w = torch.randn(cout, 64, 3, 3) / 24.0
    b = torch.randn(cout)
    got, _ = run_conv(x, w, b, 1, 1, 1, cuda, variant=variant)
    assert_close(got, ref_conv(x, w, b, 1, 1, 1))


@pytest.mark.parametrize('variant,cout', [(0, 128), (1, 64), (2, 32), (3, 64), (4, 32), (5, 64), (6, 32), (7, 128),
                                          (8, 64), (9, 64), (10, 32), (11, 64), (12, 128), (13, 64), (14, 128), (15, 32),
                                          (16, 64), (17, 128), (18, 64), (19, 128), (20, 128), (21, 128)])
def test_conv_pointwise_instances_of_all_tile_variants(variant, cout, cuda):
    """1x1 / stride 1 / no padding layers run the POINTWISE instance of the picked tile variant (division-free set-up,
    one add per K-chunk): ragged pixel count, K tail (Cin = 72), input channel slice, against torch."""
    torch.manual_seed(100 + variant)
    x = torch.randn(2, 72, 11, 13) + 1.5   # non-zero mean: a wrong K-tail or row mask shows up
    w = torch.randn(cout, 72, 1, 1) / 72 ** 0.5
    b = torch.randn(cout)
    got, _ = run_conv(x, w, b, 1, 0, 1, cuda, variant=variant, in_ld=80, in_off=4)
    assert_close(got, ref_conv(x, w, b, 1, 0, 1))


@pytest.mark.parametrize('variant,cout,split', [(3, 128, 64),    # tile (64) on one side of the split: uniform stores
                                                (3, 128, 32),    # split inside a tile: per-element side selection
                                                (0, 256, 128), (7, 256, 128), (13, 128, 64)])
@pytest.mark.parametrize('res', [False, True])
def test_conv_split_stores_whole_tile_and_mixed(variant, cout, split, res, cuda):
    torch.manual_seed(variant + cout + split)
    x = torch.randn(2, 64, 16, 16)            # 512 pixels: full tiles for every variant above
    w = torch.randn(cout, 64, 1, 1) / 8.0
    b = torch.randn(cout)
    r = torch.randn(2, cout, 16, 16) if res else None
    got, _ = run_conv(x, w, b, 1, 0, 1, cuda, variant=variant, split=split, res=r, post_scale=0.5 if res else 1.0)
    assert_close(got, ref_conv(x, w, b, 1, 0, 1, r, 0.5 if res else 1.0))


@pytest.mark.parametrize('shape', [(2, 8, 12), (3, 5, 3), (1, 23, 40)])   # rows of 12 / 3 / 40 pixels: the 16 rows of a
def test_conv_upsampled_store_coordinates(shape, cuda):                      # lane wrap lines and images
    N, H, W = shape
    torch.manual_seed(H * W)
    x = torch.randn(N, 64, H, W)
    w = torch.randn(64, 64, 1, 1) / 8.0
    b = torch.randn(64)
    for variant in (3, 0, 13):
        if variant == 0:
            w2, b2 = torch.randn(128, 64, 1, 1) / 8.0, torch.randn(128)
            got, up = run_conv(x, w2, b2, 1, 0, 1, cuda, variant=variant, up=True)
            ref = ref_conv(x, w2, b2, 1, 0, 1)
        else:
            got, up = run_conv(x, w, b, 1, 0, 1, cuda, variant=variant, up=True)
            ref = ref_conv(x, w, b, 1, 0, 1)
        assert_close(got, ref)
        assert torch.equal(up, F.interpolate(got, scale_factor=2, mode='nearest'))


@pytest.mark.parametrize('variant', [12, 13, 15, 18])
@pytest.mark.parametrize('case', [(2, 12, 20, 36, 128, 3, 1), (1, 32, 23, 41, 64, 3, 2), (1, 24, 10, 12, 64, 3, 2),
                                  (2, 64, 7, 9, 192, 1, 1)])
def test_conv_lds_dma_variants_zero_fill_and_ragged(variant, case, cuda):
    """LDS-DMA staging: padding taps, ragged M tiles and the K tail must read as zeros."""
    N, Cin, H, W, Cout, k, stride = case
    bn = {12: 128, 13: 64, 15: 32, 18: 64}[variant]
    if ((Cout + 31) // 32 * 32) % bn:
        pytest.skip('tile does not divide Cout')
    torch.manual_seed(variant + sum(case))
    x = torch.randn(N, Cin, H, W) + 3.0  # non-zero mean: a missed zero fill shows up at the borders
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout)
    got, _ = run_conv(x, w, b, stride, k // 2, 1, cuda, variant=variant)
    assert_close(got, ref_conv(x, w, b, stride, k // 2, 1))


def test_conv_epilogue_residual_scale_split_upsample(cuda):
    torch.manual_seed(7)
    x = torch.randn(2, 64, 8, 12)
    w = torch.randn(96, 64, 1, 1) / 8.0
    b = torch.randn(96)
    res = torch.randn(2, 96, 8, 12)
    got, _ = run_conv(x, w, b, 1, 0, 1, cuda, res=res, post_scale=0.5)
    assert_close(got, ref_conv(x, w, b, 1, 0, 1, res, 0.5))
    got, up = run_conv(x, w, b, 1, 0, 1, cuda, split=64, up=True)
    ref = ref_conv(x, w, b, 1, 0, 1)
    assert_close(got, ref)
    assert torch.equal(up, F.interpolate(got, scale_factor=2, mode='nearest'))


def test_conv_no_act_and_input_slice(cuda):
    torch.manual_seed(9)
    x = torch.randn(1, 32, 6, 9)
    w = torch.randn(6, 32, 1, 1) / 5.0
    b = torch.randn(6)
    got, _ = run_conv(x, w, b, 1, 0, 0, cuda, in_ld=96, in_off=32)
    assert_close(got, ref_conv(x, w, b, 1, 0, 0))


def test_bn_folding_fp64(cuda):
    torch.manual_seed(11)
    cout, cin = 40, 16
    w = torch.randn(cout, cin, 3, 3)
    gamma, beta = torch.rand(cout) + 0.5, torch.randn(cout)
    mean, var = torch.randn(cout), torch.rand(cout) + 0.5
    wp, bp = pack(w, None, (gamma, beta, mean, var), 1e-3)
    scale = gamma.double() / torch.sqrt(var.double() + 1e-3)
    wref = (w.double() * scale[:, None, None, None]).float()
    Kpad = (cin * 9 + 31) // 32 * 32
    wpk = wp.view(-1, Kpad)[:cout, :cin * 9].view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    assert torch.equal(wpk, wref)
    assert torch.equal(bp[:cout], (beta.double() - mean.double() * scale).float())
    assert torch.all(bp[cout:] == 0) and torch.all(wp.view(-1, Kpad)[cout:] == 0)


@pytest.mark.parametrize('N,H,W,Cc', [(2, 23, 40, 64), (1, 6, 10, 24), (1, 70, 90, 16)])  # last: > LDS, sweep kernel
def test_spp_pool_matches_torch_maxpool(N, H, W, Cc, cuda):
    lib = _lib.load()
    torch.manual_seed(H)
    x = torch.randn(N, Cc, H, W)
    cat = torch.full((N, H, W, 4 * Cc + 8), -5.0)
    cat[..., 8:8 + Cc] = x.permute(0, 2, 3, 1)
    cat = cat.to(cuda)
    check(lib.st_spp_pool(ptr(cat), 4 * Cc + 8, 8, N, H, W, Cc, ptr(cat), 4 * Cc + 8, 8, _lib.current_stream()))
    torch.cuda.synchronize()
    got = cat.cpu()
    assert torch.all(got[..., :8] == -5.0)
    ref = torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1).permute(0, 2, 3, 1)
    assert torch.equal(got[..., 8:], ref)
    # separate input buffer (x is copied into the first C channels of out)
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    out = torch.empty(N, H, W, 4 * Cc, device=cuda)
    check(lib.st_spp_pool(ptr(xin), Cc, 0, N, H, W, Cc, ptr(out), 4 * Cc, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize('planes', [3, 1])
@pytest.mark.parametrize('N,H,W,cout,act', [(2, 64, 96, 32, 1), (1, 50, 264, 24, 1), (1, 16, 136, 64, 0),
                                            (3, 34, 36, 8, 1), (40, 32, 128, 32, 1)])
def test_fused_focus_stem_matches_torch(N, H, W, cout, act, planes, cuda):
    """st_stem_focus_conv == Focus slicing (TL, BL, TR, BR) -> 3x3/s1/p1 conv -> BN (running stats) -> SiLU of
    the reference stem (csp_darknet_disparity_v1.py:104-111), including ragged tiles (H/2, W/2 not multiples of
    the 8 x 64 tile), Cout < 32 (padding lanes must not be stored), Cout = 64 (two MFMA column blocks), an
    output slice of a wider buffer, and more tiles than persistent workgroups (N = 40: 640 tiles > 512 slots).
    planes = 1: identical image planes, plane-summed weights, K = 36."""
    lib = _lib.load()
    torch.manual_seed(20 + cout)
    x = torch.rand(N, 3, H, W) * 255.0
    if planes == 1:   # the disparity input: a 3-channel repeat of one map; planes 1, 2 are never read
        x = x[:, :1].repeat(1, 3, 1, 1).contiguous()
    w = torch.randn(cout, 12, 3, 3) / 200.0
    gamma, beta = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    mean, var = torch.randn(cout) * 0.5, torch.rand(cout) + 0.5
    eps = 1e-3
    wp = torch.empty(lib.st_stem_packed_floats(cout), dtype=torch.float32)
    bp = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32)
    check(lib.st_stem_pack_weights(ptr(w), None, ptr(gamma), ptr(beta), ptr(mean), ptr(var), eps, cout, planes,
                                   ptr(wp), ptr(bp)))
    xd, wd, bd = x.to(cuda), wp.to(cuda), bp.to(cuda)
    ld, off = cout + 8, 4
    out = torch.full((N, H // 2, W // 2, ld), -777.0, device=cuda)
    check(lib.st_stem_focus_conv(ptr(xd), N, H, W, planes, ptr(wd), ptr(bd), cout, ptr(out), ld, off, act, None))
    torch.cuda.synchronize()
    xx = x.double()
    foc = torch.cat((xx[..., ::2, ::2], xx[..., 1::2, ::2], xx[..., ::2, 1::2], xx[..., 1::2, 1::2]), dim=1)
    ref = F.conv2d(foc, w.double(), None, 1, 1)
    ref = (ref - mean.double()[None, :, None, None]) / torch.sqrt(var.double() + eps)[None, :, None, None] \
        * gamma.double()[None, :, None, None] + beta.double()[None, :, None, None]
    if act:
        ref = F.silu(ref)
    got = out.cpu()
    assert_close(got[..., off:off + cout].permute(0, 3, 1, 2), ref)
    assert torch.all(got[..., :off] == -777.0) and torch.all(got[..., off + cout:] == -777.0)


@pytest.mark.parametrize('N,h,w,H,W,cout,pad', [
    (2, 50, 72, 64, 96, 32, 114.0),      # padded bottom + right, ragged tiles
    (3, 64, 96, 64, 96, 24, 0.0),        # no padding, Cout < 32
    (8, 720, 1280, 736, 1280, 32, 114.0),   # the bench / AirDrone geometry
    (33, 30, 132, 32, 160, 64, 7.0),     # first 32 frames only are legal: see below
])
def test_fused_stem_raw_uint8_frames_equal_pack_then_stem(N, h, w, H, W, cout, pad, cuda):
    """st_stem_focus_conv_u8 (uint8 frames in allocations of their own; cast + pad inside the window staging) ==
    st_pack_raw_frames (data_preprocessor_disparity_v1.py:38-51: cast, pad to H x W with pad_value) followed by
    st_stem_focus_conv, bit for bit; more than 32 frames per launch is refused."""
    import ctypes as C
    lib = _lib.load()
    torch.manual_seed(3)
    frames = [torch.randint(0, 256, (3, h, w), dtype=torch.uint8).to(cuda) for _ in range(N)]
    wgt = torch.randn(cout, 12, 3, 3) / 200.0
    gamma, beta, mean, var = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2, torch.randn(cout) * 0.5, torch.rand(cout) + 0.5
    wp = torch.empty(lib.st_stem_packed_floats(cout), dtype=torch.float32)
    bp = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32)
    check(lib.st_stem_pack_weights(ptr(wgt), None, ptr(gamma), ptr(beta), ptr(mean), ptr(var), 1e-3, cout, 3, ptr(wp), ptr(bp)))
    wd, bd = wp.to(cuda), bp.to(cuda)
    table = (C.c_void_p * N)(*[f.data_ptr() for f in frames])
    if N > 32:
        got = torch.zeros(N, H // 2, W // 2, cout, device=cuda)
        assert lib.st_stem_focus_conv_u8(table, N, h, w, H, W, pad, ptr(wd), ptr(bd), cout, ptr(got), cout, 0, 1, None) != 0
        N = 32
    x = torch.empty(N, 3, H, W, device=cuda)
    check(lib.st_pack_raw_frames(table, N, h, w, H, W, pad, ptr(x), None))
    ref = torch.full((N, H // 2, W // 2, cout), -7.0, device=cuda)
    got = torch.full((N, H // 2, W // 2, cout), -7.0, device=cuda)
    check(lib.st_stem_focus_conv(ptr(x), N, H, W, 3, ptr(wd), ptr(bd), cout, ptr(ref), cout, 0, 1, None))
    check(lib.st_stem_focus_conv_u8(table, N, h, w, H, W, pad, ptr(wd), ptr(bd), cout, ptr(got), cout, 0, 1, None))
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    assert float(ref.abs().max()) > 0.1 and not bool((ref == -7.0).any())
    # refused: non-integral pad value, width not a multiple of 4 (callers fall back to the pack pass)
    assert lib.st_stem_focus_conv_u8(table, N, h, w, H, W, 113.5, ptr(wd), ptr(bd), cout, ptr(got), cout, 0, 1, None) != 0
    assert lib.st_stem_focus_conv_u8(table, N, h, w - 2, H, W, pad, ptr(wd), ptr(bd), cout, ptr(got), cout, 0, 1, None) != 0


def test_fused_stem_rejects_bad_arguments(cuda):
    lib = _lib.load()
    x = torch.zeros(1, 3, 32, 32, device=cuda)
    o = torch.zeros(1, 16, 16, 80, device=cuda)
    w = torch.zeros(lib.st_stem_packed_floats(32), device=cuda)
    assert lib.st_stem_focus_conv(ptr(x), 1, 31, 32, 3, ptr(w), ptr(w), 32, ptr(o), 32, 0, 1, None) != 0   # odd H
    assert lib.st_stem_focus_conv(ptr(x), 1, 30, 30, 3, ptr(w), ptr(w), 32, ptr(o), 32, 0, 1, None) != 0   # W % 4
    assert lib.st_stem_focus_conv(ptr(x), 1, 30, 32, 3, ptr(w), ptr(w), 80, ptr(o), 80, 0, 1, None) != 0   # Cout > 64
    assert lib.st_stem_focus_conv(ptr(x), 1, 30, 32, 3, ptr(w), ptr(w), 32, ptr(o), 32, 8, 1, None) != 0   # slice > ld
    assert lib.st_stem_focus_conv(ptr(x), 1, 32, 32, 2, ptr(w), ptr(w), 32, ptr(o), 32, 0, 1, None) != 0   # planes
    assert b'stem_focus_conv' in lib.st_last_error()


@pytest.mark.parametrize('cin,cout,split,res,act,shape,in_ld,in_off', [
    (64, 64, 32, False, 1, (2, 23, 41), None, 0),     # CSP main+short conv: split outputs, ragged last tile
    (64, 64, None, True, 1, (1, 40, 67), None, 0),    # CSP final conv of the disparity branch: (a + res) * 0.5
    (32, 32, None, False, 1, (3, 17, 33), 64, 32),    # bottleneck conv1 reading a channel slice of a wider buffer
    (64, 24, None, False, 0, (1, 9, 130), None, 0),   # Cout < 32 (padded MFMA rows must not be stored), no act
    (32, 64, 40, True, 1, (1, 31, 29), None, 0),      # split not on a 32 boundary (both outputs inside one block)
    (64, 64, None, False, 1, (70, 32, 32), None, 0),  # 560 tiles > 512 persistent workgroups
])
def test_pointwise_streaming_kernel_matches_torch(cin, cout, split, res, act, shape, in_ld, in_off, cuda):
    """Tile variant 41 (pointwise_conv.hip) == the generic conv semantics on 1x1 layers: bias, SiLU,
    (v + res) * post_scale, split outputs, channel-offset input / output slices, ragged tiles."""
    torch.manual_seed(cin + cout + shape[1])
    N, H, W = shape
    x = torch.randn(N, cin, H, W)
    w = torch.randn(cout, cin, 1, 1) / (cin ** 0.5)
    b = torch.randn(cout)
    r = torch.randn(N, cout, H, W) if res else None
    got, _ = run_conv(x, w, b, 1, 0, act, cuda, variant=41, res=r, post_scale=0.5 if res else 1.0, split=split,
                      in_ld=in_ld, in_off=in_off)
    assert_close(got, ref_conv(x, w, b, 1, 0, act, r, 0.5 if res else 1.0))
    # and bit-for-bit deterministic against itself across launches (persistent tiles, no atomics)
    again, _ = run_conv(x, w, b, 1, 0, act, cuda, variant=41, res=r, post_scale=0.5 if res else 1.0, split=split,
                        in_ld=in_ld, in_off=in_off)
    assert torch.equal(got, again)


def test_pointwise_streaming_kernel_rejects_other_shapes(cuda):
    lib = _lib.load()
    x = torch.randn(1, 48, 8, 8)
    w = torch.randn(32, 48, 1, 1)
    with pytest.raises(Exception, match='not supported by the streaming kernel'):
        run_conv(x, w, torch.zeros(32), 1, 0, 1, cuda, variant=41)      # Cin = 48
    with pytest.raises(Exception, match='not supported by the streaming kernel'):
        run_conv(torch.randn(1, 32, 8, 8), torch.randn(32, 32, 3, 3), torch.zeros(32), 1, 1, 1, cuda, variant=41)
    assert b'streaming kernel' in lib.st_last_error()


@pytest.mark.parametrize('cin,N,H,W', [(64, 2, 23, 41), (32, 1, 64, 96)])
def test_chained_pointwise_pair_matches_torch(cin, N, H, W, cuda):
    """st_conv1x1_chain: main+short 1x1 conv (split 32|32 into two buffers) with the bottleneck conv1 (32 -> 32)
    chained on its first 32 outputs from registers == the two convolutions run one after the other."""
    lib = _lib.load()
    torch.manual_seed(cin + H)
    x = torch.randn(N, cin, H, W)
    wa, ba = torch.randn(64, cin, 1, 1) / cin ** 0.5, torch.randn(64)
    wb, bb = torch.randn(32, 32, 1, 1) / 32 ** 0.5, torch.randn(32)
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    (wpa, bpa), (wpb, bpb) = pack(wa, ba), pack(wb, bb)
    wpa, bpa, wpb, bpb = wpa.to(cuda), bpa.to(cuda), wpb.to(cuda), bpb.to(cuda)
    main = torch.full((N, H, W, 32), -777.0, device=cuda)
    cat = torch.full((N, H, W, 64), -777.0, device=cuda)      # short goes to channels [32, 64)
    tmp = torch.full((N, H, W, 36), -777.0, device=cuda)      # chained output at channel offset 4
    a = StConvDesc()
    a.in_dev = xin.data_ptr(); a.N, a.Hi, a.Wi, a.Cin, a.in_ld, a.in_off = N, H, W, cin, cin, 0
    a.wgt_dev = wpa.data_ptr(); a.bias_dev = bpa.data_ptr()
    a.Cout, a.KH, a.KW, a.stride, a.pad = 64, 1, 1, 1, 0
    a.out1_dev = main.data_ptr(); a.out1_ld, a.out1_off, a.split = 32, 0, 32
    a.out2_dev = cat.data_ptr(); a.out2_ld, a.out2_off = 64, 32
    a.act = 1
    b = StConvDesc()
    b.in_dev = main.data_ptr(); b.N, b.Hi, b.Wi, b.Cin, b.in_ld, b.in_off = N, H, W, 32, 32, 0
    b.wgt_dev = wpb.data_ptr(); b.bias_dev = bpb.data_ptr()
    b.Cout, b.KH, b.KW, b.stride, b.pad = 32, 1, 1, 1, 0
    b.out1_dev = tmp.data_ptr(); b.out1_ld, b.out1_off, b.split = 36, 4, 32
    b.act = 1
    check(lib.st_conv1x1_chain(C.byref(a), C.byref(b), _lib.current_stream()))
    torch.cuda.synchronize()
    ya = ref_conv(x, wa, ba, 1, 0, 1)
    yb = F.silu(F.conv2d(ya[:, :32], wb.double(), bb.double()))
    assert_close(main.cpu().permute(0, 3, 1, 2), ya[:, :32])
    assert_close(cat.cpu()[..., 32:].permute(0, 3, 1, 2), ya[:, 32:])
    assert torch.all(cat.cpu()[..., :32] == -777.0)
    assert_close(tmp.cpu()[..., 4:].permute(0, 3, 1, 2), yb)
    assert torch.all(tmp.cpu()[..., :4] == -777.0)
    # the chained result is bit-identical to running conv1 separately on the stored main output (same kernel,
    # same operand values, same summation order)
    tmp2 = torch.full((N, H, W, 36), -777.0, device=cuda)
    b.out1_dev = tmp2.data_ptr()
    check(lib.st_conv2d_nhwc_variant(C.byref(b), _lib.current_stream(), 41))
    torch.cuda.synchronize()
    assert torch.equal(tmp, tmp2)
    # pairs that cannot be chained are refused
    b.Cin = 64
    assert lib.st_conv1x1_chain(C.byref(a), C.byref(b), _lib.current_stream()) != 0


@pytest.mark.parametrize('cin,cout,res,act,shape,in_ld,in_off', [
    (48, 48, False, 1, (2, 23, 41), None, 0),     # cost-volume aggregation conv: Cout = 48 exact, ragged tiles
    (48, 48, False, 0, (1, 8, 64), None, 0),      # last aggregation layer: no activation, exact tiles
    (32, 32, True, 1, (2, 19, 70), None, 0),      # CSP bottleneck conv2 + identity
    (64, 64, True, 1, (1, 9, 33), 96, 32),        # input channel slice of a wider buffer, + residual
    (32, 64, False, 1, (1, 5, 7), None, 0),       # image smaller than one tile
    (64, 32, False, 1, (3, 12, 40), None, 0),
])
def test_direct_conv3x3_kernel_matches_torch(cin, cout, res, act, shape, in_ld, in_off, cuda):
    """Tile variant 42 (direct_conv.hip, 16x16x4 MFMA, window in LDS, per-tap weights) == conv3x3/s1/p1 + bias +
    SiLU (+ (v + res) * post_scale): zero padding from the DMA range check, ragged tiles, channel slices."""
    torch.manual_seed(cin * 3 + cout + shape[2])
    N, H, W = shape
    x = torch.randn(N, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    b = torch.randn(cout)
    r = torch.randn(N, cout, H, W) if res else None
    got, _ = run_conv(x, w, b, 1, 1, act, cuda, variant=42, res=r, post_scale=0.5 if res else 1.0, in_ld=in_ld,
                      in_off=in_off)
    assert_close(got, ref_conv(x, w, b, 1, 1, act, r, 0.5 if res else 1.0))


def test_direct_conv3x3_kernel_rejects_other_shapes(cuda):
    for cin, cout, k, s in ((40, 48, 3, 1), (48, 48, 1, 1), (48, 48, 3, 2), (32, 128, 3, 1)):
        with pytest.raises(Exception, match='not supported by the direct 3x3 kernel'):
            run_conv(torch.randn(1, cin, 8, 8), torch.randn(cout, cin, k, k), torch.zeros(cout), s, k // 2, 1, cuda,
                     variant=42)


def test_experimental_variants_are_not_in_the_product_library(stlib, cuda):
    """The wave-specialised experiments (ids 22..29) and the timing-only ablation builds (ids >= 100, wrong results by
    construction) exist only in the tools build (make ABLATION=1): the product library refuses them."""
    from stereotracking_amd._lib import StConvDesc
    x = torch.randn(1, 8, 8, 32, device=cuda)
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = 1, 8, 8, 32, 32, 0
    d.wgt_dev = x.data_ptr(); d.bias_dev = x.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = 32, 1, 1, 1, 0
    d.out1_dev = x.data_ptr(); d.out1_ld, d.out1_off, d.split = 32, 0, 32
    import ctypes as C
    for v in (22, 25, 29, 100, 103, 600):
        assert stlib.st_conv2d_nhwc_variant(C.byref(d), None, v) != 0, v
    assert not hasattr(stlib, 'st_detector_set_skip')


@pytest.mark.parametrize('cin,cout,res,act,shape,in_ld,in_off', [
    (128, 128, False, 1, (2, 23, 41), None, 0),    # head tower at odd sizes: ragged tile blocks in x and y
    (64, 64, True, 1, (1, 46, 80), None, 0),       # CSP bottleneck conv2 + identity
    (128, 256, False, 1, (1, 20, 36), 160, 32),    # fused cls|reg first tower conv, input = a channel slice
    (256, 256, False, 0, (1, 9, 17), None, 0),     # 8 K-chunks, no activation, one partial tile block
    (32, 64, True, 1, (2, 7, 5), None, 0),         # single K-chunk, image smaller than one tile block
    (48, 48, False, 1, (2, 23, 41), None, 0),      # cost-volume aggregation conv: Cin tail (48 = 32 + 16), Cout padded to 64
    (32, 32, True, 1, (1, 46, 80), 64, 32),        # stage-1 bottleneck conv2 + identity: 32-cout workgroups
    (32, 32, True, 1, (2, 23, 41), None, 0),       # the same instance on ragged tile blocks (residual tile staged in LDS)
    (16, 32, True, 0, (1, 9, 9), None, 0),         # half a K-chunk + residual, no activation
    (64, 96, False, 0, (1, 12, 20), None, 0),      # three 32-cout blocks
    (16, 32, False, 1, (1, 9, 9), None, 0),        # half a K-chunk
])
def test_winograd_conv_matches_direct_convolution(cin, cout, res, act, shape, in_ld, in_off, cuda):
    """Tile variant 43 (wino_conv.hip): Winograd F(2x2,3x3) evaluates the SAME 3x3 / stride-1 convolution with 2.25x
    fewer multiplies; against torch conv2d in fp64 (tolerance 1e-4 of the output scale, as for every MFMA variant) and
    against the implicit-GEMM kernel on the same packed weights."""
    N, H, W = shape
    g = torch.Generator().manual_seed(cin * 1000 + cout + H)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(N, cout, H, W, generator=g) if res else None
    ref = ref_conv(x, w, b, 1, 1, act, r, 0.5 if res else 1.0)
    got, _ = run_conv(x, w, b, 1, 1, act, cuda, variant=43, res=r, post_scale=0.5 if res else 1.0, in_ld=in_ld,
                      in_off=in_off)
    assert_close(got, ref)
    base, _ = run_conv(x, w, b, 1, 1, act, cuda, variant=4, res=r, post_scale=0.5 if res else 1.0, in_ld=in_ld,
                       in_off=in_off)
    assert_close(got, base.double(), tol=2e-5)
    if (cout % 64 == 0 or 32 < cout < 64) and 'ablation' in os.path.basename(os.environ.get('ST_LIBRARY', '')):
        # (tools build only) variant 57: the PERSISTENT form (workgroups loop over the tile blocks, the next
        # block's first window and weight fragment prefetched, the two cout blocks transformed one after the other): the same
        # MFMA sequence per accumulator and the same transform additions per output - bit-identical to variant 43
        pers, _ = run_conv(x, w, b, 1, 1, act, cuda, variant=57, res=r, post_scale=0.5 if res else 1.0, in_ld=in_ld,
                           in_off=in_off)
        assert torch.equal(pers, got)
    if cout % 64 == 0:   # variant 44: the same layout computed by 32-cout workgroups - bit-identical to variant 43
        narrow, _ = run_conv(x, w, b, 1, 1, act, cuda, variant=44, res=r, post_scale=0.5 if res else 1.0, in_ld=in_ld,
                             in_off=in_off)
        assert torch.equal(narrow, got)


def test_winograd_instance_needs_its_weights_and_shapes(stlib, cuda):
    x = torch.randn(1, 8, 8, 64, device=cuda)
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = 1, 8, 8, 64, 64, 0
    d.wgt_dev = x.data_ptr(); d.bias_dev = x.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = 64, 3, 3, 1, 1
    d.out1_dev = x.data_ptr(); d.out1_ld, d.out1_off, d.split = 64, 0, 64
    assert stlib.st_conv2d_nhwc_variant(C.byref(d), None, 43) != 0      # no transformed weights
    d.wgt_wino_dev = x.data_ptr()
    d.stride = 2
    assert stlib.st_conv2d_nhwc_variant(C.byref(d), None, 43) != 0      # stride 2: not a Winograd layer


@pytest.mark.parametrize('N,H,W,in_ld,in_off', [
    (2, 23, 41, 32, 0),      # odd input sizes: ragged tiles in both directions, bottom/right padding row
    (1, 64, 128, 32, 0),     # exact tiles
    (3, 10, 134, 40, 8),     # input channel slice of a wider buffer; 67 output columns (3 column tiles)
])
def test_fused_stage1_front_matches_separate_convolutions(N, H, W, in_ld, in_off, cuda):
    """st_conv3x3s2_csp_front: 3x3/s2 ConvModule (32 -> 64) + CSP main|short 1x1 (64 -> 32|32) + bottleneck conv1
    (32 -> 32) as ONE launch == the three convolutions run one after the other (fp64 torch reference, and the
    library's own three launches within fp32 rounding)."""
    lib = _lib.load()
    torch.manual_seed(H * W + N)
    x = torch.randn(N, 32, H, W)
    w3, b3 = torch.randn(64, 32, 3, 3) / 288 ** 0.5, torch.randn(64) * 0.5
    wm, bm = torch.randn(64, 64, 1, 1) / 8.0, torch.randn(64) * 0.5
    wc, bc = torch.randn(32, 32, 1, 1) / 32 ** 0.5, torch.randn(32) * 0.5
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xin = torch.randn(N, H, W, in_ld) * 3.0
    xin[..., in_off:in_off + 32] = x.permute(0, 2, 3, 1)
    xin = xin.contiguous().to(cuda)
    packed = [pack(w3, b3), pack(wm, bm), pack(wc, bc)]
    dev = [(w.to(cuda), b.to(cuda)) for w, b in packed]
    frags = []
    for (wp, _), (co, ci) in zip(packed[1:], [(64, 64), (32, 32)]):
        f = torch.empty(lib.st_front_frag_floats(co, ci), dtype=torch.float32)
        check(lib.st_front_pack_frags(ptr(wp), co, ci, ptr(f)), 'st_front_pack_frags')
        frags.append(f.to(cuda))

    def descs(s3, main, cat, tmp):
        a = StConvDesc()
        a.in_dev = xin.data_ptr(); a.N, a.Hi, a.Wi, a.Cin, a.in_ld, a.in_off = N, H, W, 32, in_ld, in_off
        a.wgt_dev, a.bias_dev = dev[0][0].data_ptr(), dev[0][1].data_ptr()
        a.Cout, a.KH, a.KW, a.stride, a.pad, a.act, a.post_scale = 64, 3, 3, 2, 1, 1, 1.0
        a.out1_dev = s3.data_ptr(); a.out1_ld, a.out1_off, a.split = 64, 0, 64
        m = StConvDesc()
        m.in_dev = s3.data_ptr(); m.N, m.Hi, m.Wi, m.Cin, m.in_ld, m.in_off = N, Ho, Wo, 64, 64, 0
        m.wgt_dev, m.bias_dev = dev[1][0].data_ptr(), dev[1][1].data_ptr()
        m.Cout, m.KH, m.KW, m.stride, m.pad, m.act, m.post_scale = 64, 1, 1, 1, 0, 1, 1.0
        m.out1_dev = main.data_ptr(); m.out1_ld, m.out1_off, m.split = 32, 0, 32
        m.out2_dev = cat.data_ptr(); m.out2_ld, m.out2_off = 64, 32
        c = StConvDesc()
        c.in_dev = main.data_ptr(); c.N, c.Hi, c.Wi, c.Cin, c.in_ld, c.in_off = N, Ho, Wo, 32, 32, 0
        c.wgt_dev, c.bias_dev = dev[2][0].data_ptr(), dev[2][1].data_ptr()
        c.Cout, c.KH, c.KW, c.stride, c.pad, c.act, c.post_scale = 32, 1, 1, 1, 0, 1, 1.0
        c.out1_dev = tmp.data_ptr(); c.out1_ld, c.out1_off, c.split = 36, 4, 32
        return a, m, c

    def bufs():
        return (torch.full((N, Ho, Wo, 64), -777.0, device=cuda), torch.full((N, Ho, Wo, 32), -777.0, device=cuda),
                torch.full((N, Ho, Wo, 64), -777.0, device=cuda), torch.full((N, Ho, Wo, 36), -777.0, device=cuda))

    s3, main, cat, tmp = bufs()
    a, m, c = descs(s3, main, cat, tmp)
    stream = _lib.current_stream()
    check(lib.st_conv3x3s2_csp_front(C.byref(a), C.byref(m), C.byref(c), frags[0].data_ptr(), frags[1].data_ptr(), stream),
          'st_conv3x3s2_csp_front')
    torch.cuda.synchronize()
    assert torch.all(s3 == -777.0), 'the 64-channel stride-2 tensor must not be materialised'
    ya = ref_conv(x, w3, b3, 2, 1, 1)
    ym = F.silu(F.conv2d(ya, wm.double(), bm.double()))
    yc = F.silu(F.conv2d(ym[:, :32], wc.double(), bc.double()))
    assert_close(main.cpu().permute(0, 3, 1, 2), ym[:, :32])
    assert_close(cat.cpu()[..., 32:].permute(0, 3, 1, 2), ym[:, 32:])
    assert torch.all(cat.cpu()[..., :32] == -777.0)
    assert_close(tmp.cpu()[..., 4:].permute(0, 3, 1, 2), yc)
    assert torch.all(tmp.cpu()[..., :4] == -777.0)
    # against the library's own three launches: same arithmetic up to fp32 summation order
    s3b, mainb, catb, tmpb = bufs()
    a2, m2, c2 = descs(s3b, mainb, catb, tmpb)
    for d in (a2, m2, c2):
        check(lib.st_conv2d_nhwc(C.byref(d), stream), 'conv')
    torch.cuda.synchronize()
    for got, ref in ((main, mainb), (cat[..., 32:], catb[..., 32:]), (tmp[..., 4:], tmpb[..., 4:])):
        assert (got - ref).abs().max().item() <= 2e-5 * (ref.abs().max().item() + 1e-6)
    # shapes the kernel is not built for are refused, not mis-computed
    a.stride = 1
    assert lib.st_conv3x3s2_csp_front(C.byref(a), C.byref(m), C.byref(c), frags[0].data_ptr(), frags[1].data_ptr(), stream) != 0
    assert b'fused front' in lib.st_last_error()


@pytest.mark.parametrize('cin,cout,split,res,act,shape,in_ld,in_off', [
    (64, 64, None, False, 1, (2, 23, 41), None, 0),      # bottleneck conv1 of stage 2 / final conv of stage 1; ragged M
    (64, 64, None, True, 1, (1, 16, 64), 96, 32),        # + residual and post_scale (branch average), input slice
    (128, 128, 64, False, 1, (2, 19, 33), None, 0),      # CSP main | short, split store
    (128, 128, None, False, 0, (1, 9, 17), None, 0),     # no activation
    (128, 128, None, True, 1, (1, 12, 20), None, 0),
    (256, 128, 64, False, 1, (1, 23, 40), None, 0),      # PAFPN top-down main | short
    (128, 64, None, False, 1, (3, 7, 9), None, 0),
    (64, 64, 32, False, 1, (1, 5, 3), None, 0),          # fewer pixels than one tile per wave
])
def test_resident_pointwise_kernel_matches_torch(cin, cout, split, res, act, shape, in_ld, in_off, cuda):
    """Tile variant 46 (pointwise_resident.hip): weights resident in LDS, pixels loaded straight into MFMA operand
    registers; same results as the fp64 torch convolution and as the implicit-GEMM kernel within fp32 rounding."""
    N, H, W = shape
    torch.manual_seed(cin + cout + H)
    x = torch.randn(N, cin, H, W)
    w, b = torch.randn(cout, cin, 1, 1) / cin ** 0.5, torch.randn(cout)
    r = torch.randn(N, cout, H, W) if res else None
    got, _ = run_conv(x, w, b, 1, 0, act, cuda, variant=46, res=r, post_scale=0.5 if res else 1.0, split=split,
                      in_ld=in_ld, in_off=in_off)
    assert_close(got, ref_conv(x, w, b, 1, 0, act, res=r, post_scale=0.5 if res else 1.0))
    base, _ = run_conv(x, w, b, 1, 0, act, cuda, variant=-1, res=r, post_scale=0.5 if res else 1.0, split=split,
                       in_ld=in_ld, in_off=in_off)
    assert (got - base).abs().max().item() <= 2e-5 * (base.abs().max().item() + 1e-6)


def test_resident_pointwise_kernel_rejects_other_shapes(cuda):
    lib = _lib.load()
    for cin, cout, k in ((32, 32, 1), (512, 256, 1), (64, 64, 3), (128, 128, 3)):
        with pytest.raises(Exception):
            run_conv(torch.randn(1, cin, 8, 8), torch.randn(cout, cin, k, k), torch.zeros(cout), 1, k // 2, 1, cuda, variant=46)
        assert b'resident 1x1 conv' in lib.st_last_error()


@pytest.mark.parametrize('cin,N,H,W', [(128, 2, 23, 41), (256, 1, 12, 20)])
def test_chained_resident_pointwise_pair_matches_torch(cin, N, H, W, cuda):
    """st_conv1x1_chain one stage deeper: main | short (cin -> 64 | 64, split store) with the bottleneck conv1
    (64 -> 64) chained on the main half from registers, on the LDS-resident kernel."""
    lib = _lib.load()
    torch.manual_seed(cin + H)
    x = torch.randn(N, cin, H, W)
    wa, ba = torch.randn(128, cin, 1, 1) / cin ** 0.5, torch.randn(128)
    wb, bb = torch.randn(64, 64, 1, 1) / 8.0, torch.randn(64)
    xin = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    (wpa, bpa), (wpb, bpb) = pack(wa, ba), pack(wb, bb)
    wpa, bpa, wpb, bpb = wpa.to(cuda), bpa.to(cuda), wpb.to(cuda), bpb.to(cuda)
    main = torch.full((N, H, W, 64), -777.0, device=cuda)
    cat = torch.full((N, H, W, 128), -777.0, device=cuda)     # short goes to channels [64, 128)
    tmp = torch.full((N, H, W, 68), -777.0, device=cuda)      # chained output at channel offset 4
    a = StConvDesc()
    a.in_dev = xin.data_ptr(); a.N, a.Hi, a.Wi, a.Cin, a.in_ld, a.in_off = N, H, W, cin, cin, 0
    a.wgt_dev = wpa.data_ptr(); a.bias_dev = bpa.data_ptr()
    a.Cout, a.KH, a.KW, a.stride, a.pad = 128, 1, 1, 1, 0
    a.out1_dev = main.data_ptr(); a.out1_ld, a.out1_off, a.split = 64, 0, 64
    a.out2_dev = cat.data_ptr(); a.out2_ld, a.out2_off = 128, 64
    a.act, a.post_scale = 1, 1.0
    b = StConvDesc()
    b.in_dev = main.data_ptr(); b.N, b.Hi, b.Wi, b.Cin, b.in_ld, b.in_off = N, H, W, 64, 64, 0
    b.wgt_dev = wpb.data_ptr(); b.bias_dev = bpb.data_ptr()
    b.Cout, b.KH, b.KW, b.stride, b.pad = 64, 1, 1, 1, 0
    b.out1_dev = tmp.data_ptr(); b.out1_ld, b.out1_off, b.split = 68, 4, 64
    b.act, b.post_scale = 1, 1.0
    check(lib.st_conv1x1_chain(C.byref(a), C.byref(b), _lib.current_stream()))
    torch.cuda.synchronize()
    ya = ref_conv(x, wa, ba, 1, 0, 1)
    yb = F.silu(F.conv2d(ya[:, :64], wb.double(), bb.double()))
    assert_close(main.cpu().permute(0, 3, 1, 2), ya[:, :64])
    assert_close(cat.cpu()[..., 64:].permute(0, 3, 1, 2), ya[:, 64:])
    assert torch.all(cat.cpu()[..., :64] == -777.0)
    assert_close(tmp.cpu()[..., 4:].permute(0, 3, 1, 2), yb)
    assert torch.all(tmp.cpu()[..., :4] == -777.0)
    # bit-identical to running conv1 separately on the stored main output with the same kernel
    tmp2 = torch.full((N, H, W, 68), -777.0, device=cuda)
    b.out1_dev = tmp2.data_ptr()
    check(lib.st_conv2d_nhwc_variant(C.byref(b), _lib.current_stream(), 46))
    torch.cuda.synchronize()
    assert torch.equal(tmp, tmp2)


# ---- split-operand (bf16x3) instances of the implicit GEMM: variants 50 (128x128), 51 (64x64), 52 (128x64) -----------
# PARKED in the tools build since round 5: their tests live in tests/test_conv_split_tools.py behind the
# `split` marker (`ST_LIBRARY=...ablation.so pytest -m split`; deselected from every other run, conftest.py); on the
# product library the variants must be REFUSED:


def test_product_library_refuses_the_parked_split_instances(cuda):
    lib = _lib.load()
    if lib.st_split_instances_available():
        pytest.skip('tools build')
    x = torch.randn(1, 64, 8, 8)
    w = torch.randn(128, 64, 1, 1) / 8.0
    with pytest.raises(RuntimeError, match='parked'):
        run_conv(x, w, torch.zeros(128), 1, 0, 1, cuda, variant=51)
    from stereotracking_amd.engine import HipDetector
    det = HipDetector(1, 64, 64, 0.375, 0.33, 1)
    det.set_split(False)                                  # 0 stays a no-op
    with pytest.raises(RuntimeError, match='tools build'):
        det.set_split(True)
    from stereotracking_amd.pipeline import StereoDensePipeline
    with pytest.raises(RuntimeError, match='tools build'):
        StereoDensePipeline(1, (64, 64), 0.375, 0.33, 1, stereo=False, split_bf16=True)


# ---- fused tail of a stage-1 CSP branch: bottleneck conv2 (Winograd, + identity) -> final_conv (1x1 on the concat) ----
def _csp_tail(dev, N, H, W, avg, seed, in_ld=32, in_off=0):
    """-> (fused output, conv2 of the unfused Winograd launch, final of the unfused launches, fp64 reference), NCHW."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    tmp = torch.randn(N, 32, H, W, generator=g)                       # bottleneck conv1 output = conv2 input
    main = torch.randn(N, 32, H, W, generator=g)                      # identity of the bottleneck
    short = torch.randn(N, 32, H, W, generator=g)                     # short_conv output: concat channels [32, 64)
    other = torch.randn(N, 64, H, W, generator=g) if avg else None    # the other branch's stage output
    w2 = torch.randn(32, 32, 3, 3, generator=g) / (3.0 * 32 ** 0.5)
    b2 = torch.randn(32, generator=g) * 0.1
    wf = torch.randn(64, 64, 1, 1, generator=g) / 8.0
    bf = torch.randn(64, generator=g) * 0.1
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    xin = torch.randn(N, H, W, in_ld, generator=g) * 3.0
    xin[..., in_off:in_off + 32] = nhwc(tmp)
    xin = xin.to(dev)
    cat = torch.full((N, H, W, 72), -777.0)                            # concat buffer: ld 72, conv2 -> [4, 36), short -> [36, 68)
    cat[..., 36:68] = nhwc(short)
    cat = cat.to(dev)
    mainb, otherb = nhwc(main).to(dev), (nhwc(other).to(dev) if avg else None)
    wp2, bp2 = pack(w2, b2)
    wino = torch.empty(lib.st_wino_packed_floats(32, 32), dtype=torch.float32)
    check(lib.st_wino_pack_weights(ptr(wp2), 32, 32, ptr(wino)), 'st_wino_pack_weights')
    wpf, bpf = pack(wf, bf)
    frag = torch.empty(lib.st_csp_tail_frag_floats(), dtype=torch.float32)
    check(lib.st_csp_tail_pack_frags(ptr(wpf), ptr(frag)), 'st_csp_tail_pack_frags')
    wp2d, bp2d, winod, wpfd, bpfd, fragd = (t.to(dev) for t in (wp2, bp2, wino, wpf, bpf, frag))
    out = torch.full((N, H, W, 68), -777.0, device=dev)               # ld 68, off 4

    def descs(cat_t, out_t):
        c2 = StConvDesc()
        c2.in_dev = xin.data_ptr(); c2.N, c2.Hi, c2.Wi, c2.Cin, c2.in_ld, c2.in_off = N, H, W, 32, in_ld, in_off
        c2.wgt_dev = wp2d.data_ptr(); c2.bias_dev = bp2d.data_ptr(); c2.wgt_wino_dev = winod.data_ptr()
        c2.Cout, c2.KH, c2.KW, c2.stride, c2.pad = 32, 3, 3, 1, 1
        c2.out1_dev = cat_t.data_ptr(); c2.out1_ld, c2.out1_off, c2.split = 72, 4, 32
        c2.res_dev = mainb.data_ptr(); c2.res_ld, c2.res_off = 32, 0
        c2.post_scale, c2.act = 1.0, 1
        f = StConvDesc()
        f.in_dev = cat_t.data_ptr(); f.N, f.Hi, f.Wi, f.Cin, f.in_ld, f.in_off = N, H, W, 64, 72, 4
        f.wgt_dev = wpfd.data_ptr(); f.bias_dev = bpfd.data_ptr()
        f.Cout, f.KH, f.KW, f.stride, f.pad = 64, 1, 1, 1, 0
        f.out1_dev = out_t.data_ptr(); f.out1_ld, f.out1_off, f.split = 68, 4, 64
        if avg:
            f.res_dev = otherb.data_ptr(); f.res_ld, f.res_off = 64, 0
        f.post_scale, f.act = (0.5 if avg else 1.0), 1
        return c2, f

    stream = _lib.current_stream()
    c2, f = descs(cat, out)
    check(lib.st_conv3x3_csp_tail(C.byref(c2), C.byref(f), ptr(fragd), stream), 'st_conv3x3_csp_tail')
    torch.cuda.synchronize()
    o = out.cpu()
    assert torch.all(o[..., :4] == -777.0), 'kernel wrote outside its channel slice'
    assert torch.all(cat.cpu()[..., :36] == -777.0), 'the fused launch must not write conv2\'s output tensor'
    fused = o[..., 4:].permute(0, 3, 1, 2)
    # the two launches it replaces, on buffers of their own
    cat2, out2 = cat.clone(), torch.full_like(out, -777.0)
    c2u, fu = descs(cat2, out2)
    check(lib.st_conv2d_nhwc_variant(C.byref(c2u), stream, 43), 'conv2')      # the Winograd launch of the plan
    check(lib.st_conv2d_nhwc_variant(C.byref(fu), stream, 46), 'final')       # the LDS-resident 1x1 launch of the plan
    torch.cuda.synchronize()
    conv2_unfused = cat2.cpu()[..., 4:36].permute(0, 3, 1, 2)
    final_unfused = out2.cpu()[..., 4:].permute(0, 3, 1, 2)
    y = ref_conv(tmp, w2, b2, 1, 1, 1, main, 1.0)
    ref = ref_conv(torch.cat([y, short.double()], 1), wf, bf, 1, 0, 1, other, 0.5) if avg else \
        ref_conv(torch.cat([y, short.double()], 1), wf, bf, 1, 0, 1)
    # the same 1x1 in float64 on the UNFUSED launch's conv2 values: what the fused kernel must equal up to the fp32 rounding
    # of its 64-term dot products alone (its conv2 arithmetic is that of the Winograd launch, instruction for instruction)
    cat64 = torch.cat([conv2_unfused.double(), short.double()], 1)
    ref_u = ref_conv(cat64, wf, bf, 1, 0, 1, other, 0.5) if avg else ref_conv(cat64, wf, bf, 1, 0, 1)
    return fused, ref_u, final_unfused, ref


@pytest.mark.parametrize('N,H,W,avg,in_ld,in_off', [
    (2, 24, 48, False, 32, 0),      # exact tile blocks (16 x 8 pixels)
    (1, 23, 41, True, 32, 0),       # ragged in x and y, with the two-branch average
    (3, 46, 80, True, 36, 4),       # conv2 input = a channel slice (the fused front kernel stores conv1 at ld 36, off 4)
    (1, 5, 7, False, 32, 0),        # image smaller than one tile block
    (2, 184, 320, False, 32, 0),    # the stage-1 map of the benched configuration: every persistent workgroup loops
])
def test_csp_tail_fused_kernel_matches_the_two_launches(N, H, W, avg, in_ld, in_off, cuda):
    """st_conv3x3_csp_tail (csrc/wino_csp_tail.hip, tile variant 56): bottleneck conv2 + identity -> final_conv (+ average)
    in one persistent launch == the Winograd launch followed by the 1x1 launch, and == the fp64 module composition
    (reference csp_darknet_disparity_v1.py:145-153 / mmdet CSPLayer)."""
    fused, ref_u, final_u, ref = _csp_tail(cuda, N, H, W, avg, seed=N * 1000 + H + W, in_ld=in_ld, in_off=in_off)
    assert_close(fused, ref)
    assert_close(fused, ref_u, tol=4e-6)      # only the 1x1's own fp32 rounding separates them
    # both halves repeat the arithmetic of the launches they replace (tile variants 43 and 46, the committed plan's choice
    # for these two layers) instruction for instruction: BIT-identical outputs, so fusing moved no float of the pipeline
    assert torch.equal(fused, final_u)


def test_csp_tail_rejects_other_shapes(cuda):
    lib = _lib.load()
    x = torch.zeros(1, 8, 8, 64, device=cuda)
    c2, f = StConvDesc(), StConvDesc()
    c2.in_dev = x.data_ptr(); c2.N, c2.Hi, c2.Wi, c2.Cin, c2.in_ld, c2.in_off = 1, 8, 8, 64, 64, 0
    c2.wgt_dev = x.data_ptr(); c2.bias_dev = x.data_ptr(); c2.wgt_wino_dev = x.data_ptr(); c2.res_dev = x.data_ptr()
    c2.Cout, c2.KH, c2.KW, c2.stride, c2.pad, c2.act = 64, 3, 3, 1, 1, 1
    c2.out1_dev = x.data_ptr(); c2.out1_ld, c2.out1_off, c2.split = 64, 0, 64
    f.in_dev = x.data_ptr(); f.N, f.Hi, f.Wi, f.Cin, f.in_ld, f.in_off = 1, 8, 8, 64, 64, 0
    f.wgt_dev = x.data_ptr(); f.bias_dev = x.data_ptr()
    f.Cout, f.KH, f.KW, f.stride, f.pad, f.act = 64, 1, 1, 1, 0, 1
    f.out1_dev = x.data_ptr(); f.out1_ld, f.out1_off, f.split = 64, 0, 64
    assert lib.st_conv3x3_csp_tail(C.byref(c2), C.byref(f), ptr(x), None) != 0      # 64 -> 64 conv2 is not the tail shape
    assert 'csp tail' in lib.st_last_error().decode()
