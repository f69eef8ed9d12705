"""GPU parity of the stereo module (cost volume, soft-argmin, upsample) and of the per-box depth
kernel against their oracles.  Cost volume and stand-alone soft-argmin / upsample are BIT-EXACT
(same fmaf order as oracle/st_oracle.c); the fused soft-argmin merges disparity groups in a
different order, so it is held to 1e-3 (north_star's float tolerance); box depth: exact decisions
(-1 / NaN / scale clamps), floats within 1e-3."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import depth as odepth
from parity_utils import rel_err
from stereotracking_amd import _lib
from stereotracking_amd._lib import check, current_stream, ptr

pytestmark = pytest.mark.gpu


def run_costvolume(fl, fr, Cc, D, T, cuda, want_cost=True):
    lib = _lib.load()
    N, Hf, Wf, ld = fl.shape
    l, r = torch.from_numpy(fl).to(cuda), torch.from_numpy(fr).to(cuda)
    cost = torch.full((N, Hf, Wf, D), float('nan'), device=cuda) if want_cost else None
    disp = torch.full((N, Hf, Wf), float('nan'), device=cuda)
    check(lib.st_costvolume_softargmin(ptr(l), ptr(r), N, Hf, Wf, Cc, ld, D, T, ptr(cost), ptr(disp),
                                       current_stream()))
    torch.cuda.synchronize()
    return (cost.cpu().numpy() if want_cost else None), disp.cpu().numpy()


@pytest.mark.parametrize('N,Hf,Wf,Cc,ld,D', [(2, 5, 70, 64, 64, 48), (1, 3, 130, 48, 64, 16), (1, 4, 33, 24, 24, 7),
                                             (1, 2, 200, 32, 32, 96)])
def test_costvolume_bit_exact_and_softargmin(N, Hf, Wf, Cc, ld, D, cuda):
    rng = np.random.RandomState(N * 100 + D)
    fl = rng.normal(0, 1, (N, Hf, Wf, ld)).astype(np.float32)
    fr = rng.normal(0, 1, (N, Hf, Wf, ld)).astype(np.float32)
    T = 8.0
    ref_cost = c_oracle.costvolume(fl, fr, Cc, D)
    ref_disp = c_oracle.softargmin(ref_cost, T)
    cost, disp = run_costvolume(fl, fr, Cc, D, T, cuda)
    assert np.array_equal(cost.view(np.uint32), ref_cost.view(np.uint32)), 'cost volume not bit-exact'
    assert rel_err(disp, ref_disp) <= 1e-3
    # streaming form (volume never written) gives the same disparity
    _, disp2 = run_costvolume(fl, fr, Cc, D, T, cuda, want_cost=False)
    assert np.array_equal(disp, disp2)
    # stand-alone soft-argmin on the materialised volume: sequential order => bit-exact
    lib = _lib.load()
    c = torch.from_numpy(ref_cost).to(cuda)
    o = torch.empty(N, Hf, Wf, device=cuda)
    check(lib.st_softargmin(ptr(c), N, Hf, Wf, D, T, ptr(o), current_stream()))
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy().view(np.uint32), ref_disp.view(np.uint32))


AGG_VARIANT = 1    # the aggressor form that made 236 of 240 volumes of the OLD kernel form wrong (tests/helpers/mfma_aggressor.hip)


def _mfma_aggressor():
    """tests/helpers/libmfma_aggressor.so (built by __graft_entry__.build(); rebuilt here when hipcc is at hand and the
    file is missing): a kernel that keeps every SIMD busy with v_mfma_f32_16x16x32_bf16."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers')
    so = os.path.join(here, 'libmfma_aggressor.so')
    if not os.path.exists(so):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC',
                               os.path.join(here, 'mfma_aggressor.hip'), '-o', so])
    lib = C.CDLL(so)
    lib.st_test_bf16_mfma_busy.restype = C.c_int
    lib.st_test_bf16_mfma_busy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return lib


def test_costvolume_beside_bf16_mfma_kernels_equals_serial_run(cuda):
    """Round-4 finding, closed structurally in round 5: a v_pk_fma_f32 whose source is broadcast by op_sel drops single
    16-lane passes of its low result half while bf16 MFMAs of ANY kernel execute on the chip (tools/micro/pkfma_corun.hip
    reproduces it in registers, profiles/r05_pkfma_corun.txt); the cost-volume kernel built that way returned a wrong
    volume in 236 of 240 co-runs (profiles/r05_corun_cv_stress.txt).  The library ships a packed form without such operands and
    contains no bf16 MFMA of its own any more, so the test brings the aggressor along (tests/helpers/mfma_aggressor.hip:
    a v_mfma_f32_16x16x32_bf16 loop with LDS-fed operands; beside it the OLD kernel form returned 236 of 240 volumes
    wrong and fails THIS test - verified with the tools build, profiles/r05_corun_cv_stress.txt).  ONE co-run:
    four cost volumes on four streams beside the aggressor on two more; every volume must equal the one the same call
    gives alone, bit for bit (consumer contract: ocsort_disparity.py:115,132-134 reads this disparity per box)."""
    lib = _lib.load()
    agg = _mfma_aggressor()
    N, Hf, Wf, Cc, D = 8, 184, 320, 64, 48
    NS = 4
    g = torch.Generator(device='cpu').manual_seed(5)
    feats = [(torch.randn(N, Hf, Wf, Cc, generator=g).to(cuda), torch.randn(N, Hf, Wf, Cc, generator=g).to(cuda))
             for _ in range(NS)]
    vols = [torch.full((N, Hf, Wf, D), float('nan'), device=cuda) for _ in range(NS)]

    def cost_volume(i, stream):
        fl, fr = feats[i]
        check(lib.st_costvolume_softargmin(ptr(fl), ptr(fr), N, Hf, Wf, Cc, Cc, D, 32.0, ptr(vols[i]), None,
                                           C.c_void_p(stream.cuda_stream)))

    ref = []
    for i in range(NS):
        cost_volume(i, torch.cuda.current_stream())
        torch.cuda.synchronize()
        ref.append(vols[i].clone())
        vols[i].fill_(float('nan'))
    scratch = torch.zeros(65536, device=cuda)
    streams = [torch.cuda.Stream() for _ in range(NS)]
    extra = [torch.cuda.Stream() for _ in range(2)]
    # the aggressor's FIRST launch loads its code object and starts late: warm it up, or nothing overlaps (the old kernel
    # form passes a cold co-run and fails this one - checked with the tools build, profiles/r05_corun_cv_stress.txt)
    for e in extra:
        assert agg.st_test_bf16_mfma_busy(scratch.data_ptr(), 10, AGG_VARIANT, e.cuda_stream) == 0
    torch.cuda.synchronize()
    for i, s in enumerate(streams):
        if i < len(extra):
            assert agg.st_test_bf16_mfma_busy(scratch.data_ptr(), 3000, AGG_VARIANT, extra[i].cuda_stream) == 0
        cost_volume(i, s)
    torch.cuda.synchronize()
    for i in range(NS):
        assert torch.equal(vols[i], ref[i]), f'cost volume {i} differs from its serial run beside bf16 MFMA kernels'


@pytest.mark.parametrize('N,Hf,Wf,Cc,D', [(1, 3, 300, 8, 192), (2, 2, 210, 16, 160), (1, 2, 260, 24, 256)])
def test_wide_volume_materialised_in_slabs_is_bit_exact(N, Hf, Wf, Cc, D, cuda):
    """SURVEY.md §8(d)'s full-resolution sizing (D = 192 levels): volumes wider than the 128 disparities one launch of the
    tiled kernel covers are MATERIALISED slab by slab (st_costvolume_softargmin with out_disp = NULL); every slab must be
    the same fmaf chain as the oracle, including the x - d < 0 zeros that straddle slab borders."""
    rng = np.random.RandomState(D)
    fl = rng.normal(0, 1, (N, Hf, Wf, Cc)).astype(np.float32)
    fr = rng.normal(0, 1, (N, Hf, Wf, Cc)).astype(np.float32)
    ref = c_oracle.costvolume(fl, fr, Cc, D)
    lib = _lib.load()
    l, r = torch.from_numpy(fl).to(cuda), torch.from_numpy(fr).to(cuda)
    cost = torch.full((N, Hf, Wf, D), float('nan'), device=cuda)
    check(lib.st_costvolume_softargmin(ptr(l), ptr(r), N, Hf, Wf, Cc, Cc, D, 1.0, ptr(cost), None, current_stream()))
    torch.cuda.synchronize()
    assert np.array_equal(cost.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_costvolume_recovers_known_shift(cuda):
    """Right = left shifted by a constant disparity: soft-argmin with a sharp temperature returns it."""
    rng = np.random.RandomState(3)
    N, Hf, Wf, Cc, D, d_true = 1, 4, 96, 32, 24, 9
    fr = rng.normal(0, 1, (N, Hf, Wf, Cc)).astype(np.float32)
    fl = np.zeros_like(fr)
    fl[:, :, d_true:] = fr[:, :, :-d_true]  # left[x] = right[x - d]
    _, disp = run_costvolume(fl, fr, Cc, D, 64.0, cuda)
    assert np.abs(disp[:, :, D:] - d_true).max() < 1e-2


@pytest.mark.parametrize('Hf,Wf,s,cut_h,cut_w', [(24, 40, 4, 16, 0),      # four pixels per thread
                                                 (24, 40, 4, 5, 7),       # valid region ends inside a group of four
                                                 (30, 37, 1, 3, 2),       # scale 1 (the full-resolution mode), odd width
                                                 (16, 33, 2, 0, 1)])      # width 66: the one-pixel kernel
def test_upsample_pack_bit_exact_and_matches_torch(Hf, Wf, s, cut_h, cut_w, cuda):
    rng = np.random.RandomState(4)
    N = 2
    lr = rng.uniform(0, 48, (N, Hf, Wf)).astype(np.float32)
    H, W, vh, vw = Hf * s, Wf * s, Hf * s - cut_h, Wf * s - cut_w
    ref = c_oracle.disp_upsample(lr, s, vh, vw)
    lib = _lib.load()
    out = torch.full((N, 3, H, W), float('nan'), device=cuda)
    lr_dev = torch.from_numpy(lr).to(cuda)  # keep alive: ptr() does not hold a reference
    check(lib.st_disp_upsample_pack(ptr(lr_dev), N, Hf, Wf, s, H, W, vh, vw, ptr(out),
                                    current_stream()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # the oracle itself agrees with torch's bilinear interpolation (align_corners=False)
    t = torch.nn.functional.interpolate(torch.from_numpy(lr)[:, None], scale_factor=s, mode='bilinear',
                                        align_corners=False)[:, 0].numpy() * s
    assert np.abs(ref[:, 0, :vh, :vw] - t[:, :vh, :vw]).max() <= 1e-4 * 48 * s
    assert np.all(ref[:, :, vh:] == 0) and np.all(ref[:, :, :, vw:] == 0)


def make_disp(rng, H, W):
    """Disparity map with structure: background far, a few nearer rectangles, invalid holes."""
    disp = np.full((H, W), 2.0, np.float32) + rng.uniform(0, 0.5, (H, W)).astype(np.float32)
    for _ in range(12):
        y, x = rng.randint(0, H - 40), rng.randint(0, W - 60)
        h, w = rng.randint(6, 40), rng.randint(6, 60)
        disp[y:y + h, x:x + w] = rng.uniform(3, 40) + rng.uniform(0, 0.3, (h, w))
    disp[rng.uniform(size=(H, W)) < 0.05] = 0.0
    return disp


def run_box_depth(disp3, boxes, counts, cuda, baseline=0.25, focal=640.0):
    lib = _lib.load()
    N, _, H, W = disp3.shape
    M = boxes.shape[1]
    d = torch.from_numpy(disp3).to(cuda)
    b = torch.from_numpy(boxes).to(cuda)
    c = torch.from_numpy(counts).to(cuda)
    depth = torch.full((N, M), -7.0, device=cuda)
    scale = torch.full((N, M), -7.0, device=cuda)
    sb = torch.full((N, M, 4), -7.0, device=cuda)
    check(lib.st_box_depth(ptr(d), 3 * H * W, N, H, W, ptr(b), ptr(c), M, baseline, focal, None, 0, current_stream(),
                           ptr(depth), ptr(scale), ptr(sb)))
    torch.cuda.synchronize()
    return depth.cpu().numpy(), scale.cpu().numpy(), sb.cpu().numpy()


def test_box_depth_matches_reference_semantics(cuda):
    rng = np.random.RandomState(11)
    N, H, W, M = 2, 160, 256, 48
    disp = np.stack([make_disp(rng, H, W) for _ in range(N)])
    disp3 = np.repeat(disp[:, None], 3, 1)
    boxes = np.zeros((N, M, 4), np.float32)
    counts = np.array([M, M - 9], np.int32)
    for n in range(N):
        for k in range(M):
            x1, y1 = rng.uniform(0, W - 8), rng.uniform(0, H - 8)
            boxes[n, k] = (x1, y1, min(W, x1 + rng.uniform(1, 90)), min(H, y1 + rng.uniform(1, 70)))
    # edge cases the reference code has branches / quirks for
    boxes[0, 0] = (10.2, 10.7, 10.9, 30.0)        # zero-width after int truncation -> no valid px -> -1
    boxes[0, 1] = (0.0, 0.0, 1.9, 40.0)           # x2-2 < 0: corner slice wraps to an empty slice (NaN mean)
    boxes[0, 2] = (0.0, 5.0, 256.0, 60.0)         # wide box
    boxes[0, 3] = (30.0, 158.0, 60.0, 160.0)      # touches the bottom border
    boxes[0, 4] = (-3.5, 20.0, 40.0, 50.0)        # negative coordinate: Python slice wrap -> empty
    boxes[0, 5] = (100.0, 100.0, 101.5, 101.5)    # 1x1 box: len 1
    hole = disp3.copy()
    hole[0, :, 120:140, 200:230] = 0.0
    boxes[0, 6] = (200.0, 120.0, 230.0, 140.0)    # all-invalid window -> -1
    depth, scale, sb = run_box_depth(hole, boxes, counts, cuda)
    for n in range(N):
        k = int(counts[n])
        tb = torch.from_numpy(boxes[n, :k])
        ref_d, ref_s, ref_sb = odepth.bbox_postp_depth(tb, torch.from_numpy(hole[n:n + 1]))
        ref_d = np.array([float(v) for v in ref_d], np.float32)
        for i in range(k):
            if math.isnan(ref_d[i]):
                assert math.isnan(depth[n, i]), (n, i)
            elif ref_d[i] == -1:
                assert depth[n, i] == -1 and scale[n, i] == 1.0, (n, i, depth[n, i])
            else:
                assert abs(depth[n, i] - ref_d[i]) <= 1e-3 * max(1.0, abs(ref_d[i])), (n, i, depth[n, i], ref_d[i])
        rs, rsb = ref_s.numpy(), ref_sb.numpy()
        ok = ~np.isnan(rs)
        assert np.abs(scale[n, :k][ok] - rs[ok]).max() <= 1e-3
        assert rel_err(sb[n, :k][ok], rsb[ok]) <= 1e-3
        # rows past counts[n] are DEFINED: zero, never stale data of an earlier batch
        assert np.all(depth[n, k:] == 0.0) and np.all(scale[n, k:] == 0.0) and np.all(sb[n, k:] == 0.0)
    assert (depth[0, :7] == -1).sum() >= 3


def test_box_depth_large_box_and_w_gt_800(cuda):
    rng = np.random.RandomState(12)
    H, W = 736, 1280
    disp = (rng.uniform(1.2, 30.0, (H, W))).astype(np.float32)
    disp[720:] = 0.0
    disp3 = np.repeat(disp[None, None], 3, 1)
    boxes = np.array([[[100.0, 50.0, 890.0, 700.0], [100.0, 50.0, 901.5, 700.0], [0.0, 0.0, 1280.0, 720.0],
                       [600.3, 300.9, 640.2, 330.1]]], np.float32)
    counts = np.array([4], np.int32)
    depth, scale, sb = run_box_depth(disp3, boxes, counts, cuda)
    ref_d, ref_s, ref_sb = odepth.bbox_postp_depth(torch.from_numpy(boxes[0]), torch.from_numpy(disp3))
    assert ref_d[1] == -1 and ref_d[2] == -1  # w > 800
    for i in range(4):
        assert abs(depth[0, i] - float(ref_d[i])) <= 1e-3 * max(1.0, abs(float(ref_d[i])))
    assert np.abs(scale[0] - ref_s.numpy()).max() <= 1e-3
    assert rel_err(sb[0], ref_sb.numpy()) <= 1e-3


@pytest.mark.parametrize('h,w', [(50, 70), (48, 72), (64, 96)])   # scalar path, 4-pixel path, 4-pixel path without padding
def test_pack_raw_inputs_matches_reference_pipeline(h, w, cuda):
    """uint8 image + uint16 disparity codes -> the tensors LoadDisparityFromFile._post_processing_v2
    (loading_disparity.py:82-86,129-134), Pad_Disparity (transforms_disparity.py:234-249) and the
    preprocessor (data_preprocessor_disparity_v1.py:38-51) produce; bit-exact."""
    from stereotracking_amd.mot import pack_raw_inputs
    rng = np.random.RandomState(5)
    N = 2
    img = rng.randint(0, 256, (N, 3, h, w)).astype(np.uint8)
    code = rng.randint(0, 48 * 16, (N, h, w)).astype(np.uint16)
    code[rng.uniform(size=code.shape) < 0.1] = 65535
    out = pack_raw_inputs(torch.from_numpy(img).to(cuda), torch.from_numpy(code.view(np.int16)).to(cuda))
    torch.cuda.synchronize()
    H, W = 64, 96
    # numpy restatement of the reference transforms
    disp = code.astype(np.float32)
    disp[code == 65535] = 0
    disp = disp / 16.
    ref_img = np.full((N, 3, H, W), 0, np.float32)
    ref_img[:, :, :h, :w] = img
    ref_img[:, :, h:, :] = 114.0
    ref_img[:, :, :, w:] = 114.0
    ref_disp = np.zeros((N, 3, H, W), np.float32)
    ref_disp[:, :, :h, :w] = disp[:, None]
    ref_mask = np.zeros((N, 1, H, W), np.float32)
    ref_mask[:, 0, :h, :w] = code < 65535
    assert np.array_equal(out['img'].cpu().numpy(), ref_img)
    assert np.array_equal(out['disp_postp'].cpu().numpy(), ref_disp)
    assert np.array_equal(out['disp_mask'].cpu().numpy(), ref_mask)


@pytest.mark.parametrize('h,w,B,n', [(48, 72, 4, 4), (64, 96, 8, 5), (720, 1280, 8, 8)])
def test_raw_frames_chunk_pointer_table(h, w, B, n, cuda):
    """RawFrames.chunk (st_pack_raw_frames: the frames stay in their own allocations, pointers in the kernel
    arguments) == cast + pad-114 of the concatenated frames (data_preprocessor_disparity_v1.py:38-51), bit-exact,
    the last frame repeated up to the batch size; a non-contiguous frame takes the torch.cat route, same values."""
    from stereotracking_amd.mot import RawFrames
    rng = np.random.RandomState(11)
    frames = [torch.from_numpy(rng.randint(0, 256, (1, 3, h, w)).astype(np.uint8)).to(cuda) for _ in range(n)]
    H, W = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    ref = torch.full((B, 3, H, W), 114.0, device=cuda)
    for i in range(B):
        ref[i, :, :h, :w] = frames[min(i, n - 1)][0].float()
    out = RawFrames(frames, (H, W), 114.0).chunk(0, n, B)
    assert torch.equal(out, ref)
    strided = list(frames)
    strided[1] = torch.stack([frames[1][0], frames[1][0]], 1)[:, 0][None]      # a non-contiguous view of frame 1
    assert not strided[1].is_contiguous() and torch.equal(strided[1], frames[1])
    assert torch.equal(RawFrames(strided, (H, W), 114.0).chunk(0, n, B), ref)


@pytest.mark.parametrize('agg_layers,tuned', [(1, False), (2, False), (2, True)])
def test_stereo_module_with_aggregation_matches_oracle(agg_layers, tuned, cuda):
    """Full stereo module (stage-1 features of both views -> cost volume -> `agg_layers` 3x3 convs over
    d-as-channels -> soft-argmin -> x4 upsample) against oracle/stereo.py on the GPU's own features:
    aggregated volume and disparity within 1e-3 (float kernel, north_star tolerance), padding exactly 0."""
    from oracle import stereo as ostereo
    from stereotracking_amd.pipeline import StereoDensePipeline
    from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict
    N, H, W, D = 2, 88, 152, 32            # padded to 96 x 160 by the pipeline
    pipe = StereoDensePipeline(N, (H, W), 0.375, 0.33, 1, stereo=True, max_disp=D, max_det=32,
                               agg_layers=agg_layers)
    names = [n for n, _ in pipe.param_table()]
    assert f'stereo.agg.{agg_layers - 1}.weight' in names and f'stereo.agg.{agg_layers}.weight' not in names
    sd = synthetic_state_dict(pipe.param_table(), seed=1)
    pipe.load_state_dict(sd, autotune=tuned)
    assert (pipe.stereo_module.variant >= 0) == tuned
    batch = synthetic_batch([3, 4], H, W, D)
    img, right = batch['img'].to(cuda), batch['right'].to(cuda)
    Hf, Wf, Dl = pipe.height // 4, pipe.width // 4, D // 4
    vol = torch.full((N, Hf, Wf, Dl), float('nan'), device=cuda)
    lr = torch.full((N, Hf, Wf), float('nan'), device=cuda)
    out = torch.full((N, 3, pipe.height, pipe.width), float('nan'), device=cuda)
    pipe.stereo_module.compute(pipe.det, img, right, (H, W), lr, out, cost_out=vol)
    torch.cuda.synchronize()
    feat = pipe.det.tap('stage1_rgb').cpu().numpy()
    Cf = feat.shape[-1]
    ref_vol, ref_lr, ref_out = ostereo.disparity(feat[:N], feat[N:], Cf, Dl, pipe.temperature, sd, agg_layers,
                                                 valid_hw=(H, W))
    assert rel_err(vol.cpu().numpy(), ref_vol) <= 1e-3       # per element: |a - b| <= 1e-3 * max(1, |b|)
    assert rel_err(lr.cpu().numpy(), ref_lr) <= 1e-3
    got = out.cpu().numpy()
    assert rel_err(got, ref_out) <= 1e-3
    assert (got[:, :, H:, :] == 0).all() and (got[:, :, :, W:] == 0).all()
    assert got.min() >= 0.0 and got.max() <= D          # soft-argmin is a convex combination of the levels


# ---- 3-D aggregation (north_star: "its 3D/2D aggregation"; csrc/agg3d.hip) -----------------------------------------
@pytest.mark.parametrize('N,Hf,Wf,D,act', [(2, 5, 70, 48, 1), (1, 3, 130, 16, 0), (1, 4, 64, 48, 1), (1, 1, 1, 4, 1),
                                           (1, 7, 65, 96, 0), (2, 2, 200, 12, 1),
                                           (1, 37, 70, 48, 1),     # several row bands of the streaming stencil, ragged last band
                                           (1, 9, 40, 192, 1),     # the 16-pixel strips of D = 192
                                           (3, 19, 33, 28, 0)])    # D <= 28: the 2-register staging instance
def test_agg3d_layer_bit_exact(N, Hf, Wf, D, act, cuda):
    """One single-channel 3x3x3 layer over (d, y, x), zero padded in all three dimensions: BIT-EXACT against
    oracle_agg3d (same fmaf order, SiLU through the shared exp polynomial); ragged column strips (Wf not a multiple of
    the strip width), several row bands with a ragged last one, one-pixel volumes and D = 4 (every quad is a border quad)
    included."""
    lib = _lib.load()
    rng = np.random.RandomState(7 * D + Wf)
    vol = rng.normal(0, 1.5, (N, Hf, Wf, D)).astype(np.float32)
    w = rng.normal(0, 0.4, (3, 3, 3)).astype(np.float32)
    bias = 0.125
    ref = c_oracle.agg3d(vol, w, bias, act)
    src = torch.from_numpy(vol).to(cuda)
    dst = torch.full_like(src, float('nan'))
    w27 = (C.c_float * 27)(*w.reshape(-1).tolist())
    check(lib.st_volume_agg3d(ptr(src), ptr(dst), N, Hf, Wf, D, w27, bias, act, current_stream()), 'st_volume_agg3d')
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max()
    # argument checks: in place, D not a multiple of 4
    assert lib.st_volume_agg3d(ptr(src), ptr(src), N, Hf, Wf, D, w27, bias, act, current_stream()) != 0
    assert lib.st_volume_agg3d(ptr(src), ptr(dst), N, Hf, Wf, 6, w27, bias, act, current_stream()) != 0


@pytest.mark.parametrize('N,H,W,Cc,ld,D,act', [(2, 20, 70, 8, 8, 192, 0),    # 16-pixel strips, ragged last strip, all threads busy
                                                (1, 41, 100, 4, 4, 96, 1),    # 32-pixel strips, several bands
                                                (1, 7, 130, 16, 16, 48, 1),   # 64-pixel strips
                                                (1, 9, 45, 8, 12, 64, 0),     # D = 64: idle threads in the workgroup; padded rows
                                                (2, 5, 33, 8, 8, 20, 1),      # D = 20: five quads per pixel
                                                (1, 1, 1, 4, 4, 4, 1),        # one cell column
                                                (1, 3, 300, 8, 8, 192, 1)])   # x < d over the first strips only
def test_costvolume_agg3d_fused_bit_exact(N, H, W, Cc, ld, D, act, cuda):
    """st_costvolume_agg3d (the volume between the cost kernel and the first 3-D layer never reaches memory) against
    oracle_costvolume followed by oracle_agg3d: BIT-EXACT, zero padding of the volume in x, y and d, cells with x < d
    zero, ragged strips and bands, feature rows with padding channels (ld > C)."""
    lib = _lib.load()
    rng = np.random.RandomState(D * 3 + W)
    fl = rng.normal(0, 1.0, (N, H, W, ld)).astype(np.float32)
    fr = rng.normal(0, 1.0, (N, H, W, ld)).astype(np.float32)
    w = rng.normal(0, 0.4, (3, 3, 3)).astype(np.float32)
    bias = -0.0625
    ref = c_oracle.agg3d(c_oracle.costvolume(fl, fr, Cc, D), w, bias, act)
    assert lib.st_costvolume_agg3d_supported(Cc, D) == 1
    gl, gr = torch.from_numpy(fl).to(cuda), torch.from_numpy(fr).to(cuda)
    out = torch.full((N, H, W, D), float('nan'), device=cuda)
    w27 = (C.c_float * 27)(*w.reshape(-1).tolist())
    check(lib.st_costvolume_agg3d(ptr(gl), ptr(gr), N, H, W, Cc, ld, D, w27, bias, act, ptr(out), current_stream()),
          'st_costvolume_agg3d')
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max()
    # and the two-call form it replaces gives the same bits
    vol = torch.empty_like(out)
    out2 = torch.empty_like(out)
    check(lib.st_costvolume_softargmin(ptr(gl), ptr(gr), N, H, W, Cc, ld, D, 1.0, ptr(vol), None, current_stream()))
    check(lib.st_volume_agg3d(ptr(vol), ptr(out2), N, H, W, D, w27, bias, act, current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    # shapes the fused kernel does not take are refused, not computed differently
    assert lib.st_costvolume_agg3d_supported(64, 48) == 0 and lib.st_costvolume_agg3d_supported(8, 196) == 0
    assert lib.st_costvolume_agg3d(ptr(gl), ptr(gr), N, H, W, 5, ld, D, w27, bias, act, ptr(out), current_stream()) != 0


@pytest.mark.parametrize('N,H,W,Cc,ld,D,act', [
    (2, 9, 37, 8, 8, 192, 0),       # the benched level count: 16-column strips, ragged last strip
    (1, 23, 70, 8, 12, 192, 1),     # several bands' worth of rows, padding channels, SiLU in the 3-D layer
    (2, 7, 80, 16, 16, 96, 0),      # 32-column strips
    (1, 12, 150, 4, 4, 48, 1),      # 64-column strips, ragged
    (1, 1, 5, 8, 8, 48, 0),         # one row, narrower than a strip
])
def test_costvolume_agg3d_softargmin_single_kernel_bit_exact(N, H, W, Cc, ld, D, act, cuda):
    """st_costvolume_agg3d_softargmin (round 6): cost volume + ONE 3-D layer + soft-argmin in one kernel - the aggregated
    volume never exists.  BIT-EXACT against oracle_costvolume -> oracle_agg3d -> oracle_softargmin and against the two
    launches st_costvolume_agg3d -> st_softargmin."""
    lib = _lib.load()
    rng = np.random.RandomState(D + 7 * W + H)
    fl = rng.normal(0, 1.0, (N, H, W, ld)).astype(np.float32)
    fr = rng.normal(0, 1.0, (N, H, W, ld)).astype(np.float32)
    w = rng.normal(0, 0.4, (3, 3, 3)).astype(np.float32)
    bias, T = 0.03125, 32.0
    ref = c_oracle.softargmin(c_oracle.agg3d(c_oracle.costvolume(fl, fr, Cc, D), w, bias, act), T)
    gl, gr = torch.from_numpy(fl).to(cuda), torch.from_numpy(fr).to(cuda)
    w27 = (C.c_float * 27)(*w.reshape(-1).tolist())
    disp = torch.full((N, H, W), float('nan'), device=cuda)
    check(lib.st_costvolume_agg3d_softargmin(ptr(gl), ptr(gr), N, H, W, Cc, ld, D, w27, bias, act, T, ptr(disp),
                                             current_stream()), 'st_costvolume_agg3d_softargmin')
    torch.cuda.synchronize()
    got = disp.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.reshape(N, H, W).view(np.uint32)), np.abs(got - ref.reshape(N, H, W)).max()
    vol = torch.empty(N, H, W, D, device=cuda)
    disp2 = torch.full_like(disp, float('nan'))
    check(lib.st_costvolume_agg3d(ptr(gl), ptr(gr), N, H, W, Cc, ld, D, w27, bias, act, ptr(vol), current_stream()))
    check(lib.st_softargmin(ptr(vol), N, H, W, D, T, ptr(disp2), current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(disp, disp2)
    # level counts the single-kernel form is not built for are refused (the caller takes the two calls), never approximated
    assert lib.st_costvolume_agg3d_softargmin(ptr(gl), ptr(gr), N, H, W, Cc, ld, 64, w27, bias, act, T, ptr(disp),
                                              current_stream()) != 0
    assert 'D = 48, 96 or 192' in lib.st_last_error().decode()


def test_costvolume_agg3d_fused_equals_two_call_form_on_random_shapes(cuda):
    """Seeded sweep over 40 shapes (every D multiple of 4 up to 192, widths below / across / beyond a strip, heights from 1 row,
    C in {4, 8, 16} with and without padding channels, both activations): the fused kernel equals the two-call form bit for
    bit (which the tests above pin to the oracle)."""
    lib = _lib.load()
    rng = np.random.RandomState(20260)
    for it in range(40):
        D = 4 * int(rng.randint(1, 49))
        Cc = int(rng.choice([4, 8, 16]))
        ld = Cc + 4 * int(rng.randint(0, 3))
        N, H, W = int(rng.randint(1, 4)), int(rng.randint(1, 30)), int(rng.randint(1, 210))
        act = int(rng.randint(0, 2))
        fl = torch.from_numpy(rng.normal(0, 1.0, (N, H, W, ld)).astype(np.float32)).to(cuda)
        fr = torch.from_numpy(rng.normal(0, 1.0, (N, H, W, ld)).astype(np.float32)).to(cuda)
        w27 = (C.c_float * 27)(*rng.normal(0, 0.4, 27).astype(np.float32).tolist())
        vol = torch.empty(N, H, W, D, device=cuda)
        ref = torch.empty_like(vol)
        out = torch.full_like(vol, float('nan'))
        check(lib.st_costvolume_softargmin(ptr(fl), ptr(fr), N, H, W, Cc, ld, D, 1.0, ptr(vol), None, current_stream()))
        check(lib.st_volume_agg3d(ptr(vol), ptr(ref), N, H, W, D, w27, 0.25, act, current_stream()))
        check(lib.st_costvolume_agg3d(ptr(fl), ptr(fr), N, H, W, Cc, ld, D, w27, 0.25, act, ptr(out), current_stream()))
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (it, N, H, W, Cc, ld, D, act)


def test_costvolume_agg3d_fused_equals_two_call_form_at_full_resolution(cuda):
    """The benched size of the full-resolution mode (736 x 1280 pixels, D = 192, 8 feature channels; 2 pairs here): the fused
    kernel's volume equals the two-call form's bit for bit over all 362 M cells (80 strips x 8 bands of 92 rows per pair: every
    band seam, the ragged d > x corner and the image borders at full size), with and without the activation."""
    lib = _lib.load()
    N, H, W, Cc, D = 2, 736, 1280, 8, 192
    g = torch.Generator(device='cpu').manual_seed(5)
    gl = torch.randn(N, H, W, Cc, generator=g).to(cuda)
    gr = torch.randn(N, H, W, Cc, generator=g).to(cuda)
    w = (torch.randn(27, generator=g) * 0.3).tolist()
    w27 = (C.c_float * 27)(*w)
    vol = torch.empty(N, H, W, D, device=cuda)
    ref = torch.empty_like(vol)
    out = torch.empty_like(vol)
    check(lib.st_costvolume_softargmin(ptr(gl), ptr(gr), N, H, W, Cc, Cc, D, 1.0, ptr(vol), None, current_stream()))
    for act in (0, 1):
        check(lib.st_volume_agg3d(ptr(vol), ptr(ref), N, H, W, D, w27, 0.03125, act, current_stream()))
        out.fill_(float('nan'))
        check(lib.st_costvolume_agg3d(ptr(gl), ptr(gr), N, H, W, Cc, Cc, D, w27, 0.03125, act, ptr(out), current_stream()))
        torch.cuda.synchronize()
        assert torch.equal(out, ref), act
    del vol, ref, out


@pytest.mark.parametrize('N,Hf,Wf,D', [(1, 3, 37, 192), (2, 2, 45, 128), (1, 5, 13, 112), (1, 1, 70, 144), (1, 2, 33, 176),
                                      (1, 3, 37, 48), (2, 2, 45, 16), (1, 5, 13, 96), (1, 1, 70, 80), (1, 2, 33, 32), (1, 3, 20, 64)])
def test_softargmin_wide_volumes_bit_exact(N, Hf, Wf, D, cuda):
    """st_softargmin on volumes of 16 .. 192 levels in steps of 16 (the register kernel: the 48 levels of the benched
    default, the 192 of the full-resolution mode): rows held in registers, split over two lanes when D > 96 and D / 16 is even, the two running sums chained through the lanes in the oracle's order - BIT-EXACT against
    oracle_softargmin; pixel counts that are no multiple of the 32 / 64 pixels of a wave included."""
    lib = _lib.load()
    rng = np.random.RandomState(D + Wf)
    vol = rng.normal(0, 0.6, (N, Hf, Wf, D)).astype(np.float32)
    for T in (4.0, 32.0):
        ref = c_oracle.softargmin(vol, T)
        c = torch.from_numpy(vol).to(cuda)
        o = torch.full((N, Hf, Wf), float('nan'), device=cuda)
        check(lib.st_softargmin(ptr(c), N, Hf, Wf, D, T, ptr(o), current_stream()))
        torch.cuda.synchronize()
        assert np.array_equal(o.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (D, T)


def test_full_resolution_sizing_composition_d192_bit_exact(cuda):
    """north_star's literal sizing as a TESTED composition, not only a timed kernel: a D = 192-level volume at full
    resolution (a 64 x 256 crop; C = 8 features) built by st_costvolume_softargmin in slabs (2 x 96 disparities),
    aggregated by st_volume_agg3d (3x3x3 over d, y, x across the slab boundary), regressed by st_softargmin - each
    stage BIT-EXACT against oracle/st_oracle.c, end to end (consumer contract: loading_disparity.py:85-86,129-134 -
    float32 pixels per full-resolution pixel)."""
    lib = _lib.load()
    N, Hf, Wf, Cc, D, T = 1, 64, 256, 8, 192, 4.0
    rng = np.random.RandomState(192)
    fl = rng.normal(0, 1, (N, Hf, Wf, Cc)).astype(np.float32)
    fr = rng.normal(0, 1, (N, Hf, Wf, Cc)).astype(np.float32)
    w = rng.normal(0, 0.3, (3, 3, 3)).astype(np.float32)
    w[1, 1, 1] += 1.0
    ref_vol = c_oracle.costvolume(fl, fr, Cc, D)
    ref_agg = c_oracle.agg3d(ref_vol, w, 0.0, 0)
    ref_disp = c_oracle.softargmin(ref_agg, T)
    l, r = torch.from_numpy(fl).to(cuda), torch.from_numpy(fr).to(cuda)
    vol = torch.full((N, Hf, Wf, D), float('nan'), device=cuda)
    agg = torch.full((N, Hf, Wf, D), float('nan'), device=cuda)
    disp = torch.full((N, Hf, Wf), float('nan'), device=cuda)
    w27 = (C.c_float * 27)(*w.reshape(-1).tolist())
    check(lib.st_costvolume_softargmin(ptr(l), ptr(r), N, Hf, Wf, Cc, Cc, D, T, ptr(vol), None, current_stream()))
    check(lib.st_volume_agg3d(ptr(vol), ptr(agg), N, Hf, Wf, D, w27, 0.0, 0, current_stream()), 'st_volume_agg3d')
    check(lib.st_softargmin(ptr(agg), N, Hf, Wf, D, T, ptr(disp), current_stream()))
    torch.cuda.synchronize()
    assert np.array_equal(vol.cpu().numpy().view(np.uint32), ref_vol.view(np.uint32))
    assert np.array_equal(agg.cpu().numpy().view(np.uint32), ref_agg.view(np.uint32))
    assert np.array_equal(disp.cpu().numpy().view(np.uint32), ref_disp.view(np.uint32))
    assert np.isfinite(ref_disp).all() and ref_disp.min() >= 0 and ref_disp.max() <= D - 1


@pytest.mark.parametrize('agg3d_layers,agg_layers', [(1, 0), (2, 0), (2, 1)])
def test_stereo_module_with_3d_aggregation_matches_oracle(agg3d_layers, agg_layers, cuda):
    """The stereo module with the 3-D stage (cost volume -> `agg3d_layers` 3x3x3 layers -> `agg_layers` 2-D convs ->
    soft-argmin -> upsample) against oracle/stereo.py on the GPU's own features.  Without 2-D layers every stage is
    bit-exact, so the aggregated volume and both disparities are compared BIT FOR BIT; with a 2-D conv (float MFMA
    kernel) the usual 1e-3 applies.  One identity-initialised 3-D layer leaves the disparity unchanged."""
    from oracle import stereo as ostereo
    from stereotracking_amd.pipeline import StereoDensePipeline
    from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict
    N, H, W, D = 2, 88, 152, 32
    pipe = StereoDensePipeline(N, (H, W), 0.375, 0.33, 1, stereo=True, max_disp=D, max_det=32,
                               agg_layers=agg_layers, agg3d_layers=agg3d_layers)
    table = pipe.param_table()
    names = [n for n, _ in table]
    assert f'stereo.agg3d.{agg3d_layers - 1}.weight' in names and dict(table)['stereo.agg3d.0.weight'] == (1, 1, 3, 3, 3)
    sd = synthetic_state_dict(table, seed=1)
    g = torch.Generator().manual_seed(5)      # generic 3-D taps (the synthetic fan-in rule would make them tiny)
    for l in range(agg3d_layers):
        sd[f'stereo.agg3d.{l}.weight'] = torch.randn(1, 1, 3, 3, 3, generator=g) * 0.25
        sd[f'stereo.agg3d.{l}.bias'] = torch.randn(1, generator=g) * 0.1
    batch = synthetic_batch([3, 4], H, W, D)
    img, right = batch['img'].to(cuda), batch['right'].to(cuda)
    Hf, Wf, Dl = pipe.height // 4, pipe.width // 4, D // 4

    def run():
        vol = torch.full((N, Hf, Wf, Dl), float('nan'), device=cuda)
        lr = torch.full((N, Hf, Wf), float('nan'), device=cuda)
        out = torch.full((N, 3, pipe.height, pipe.width), float('nan'), device=cuda)
        pipe.stereo_module.compute(pipe.det, img, right, (H, W), lr, out, cost_out=vol)
        torch.cuda.synchronize()
        return vol.cpu().numpy(), lr.cpu().numpy(), out.cpu().numpy()

    sd_id = dict(sd)                                          # the same weights with IDENTITY 3-D layers
    for l in range(agg3d_layers):
        ident = torch.zeros(1, 1, 3, 3, 3)
        ident[0, 0, 1, 1, 1] = 1.0
        sd_id[f'stereo.agg3d.{l}.weight'], sd_id[f'stereo.agg3d.{l}.bias'] = ident, torch.zeros(1)
    pipe.load_state_dict(sd_id, autotune=False)
    vol_id, lr_id, _ = run()
    feat = pipe.det.tap('stage1_rgb').cpu().numpy()
    Cf = feat.shape[-1]
    plain = ostereo.disparity(feat[:N], feat[N:], Cf, Dl, pipe.temperature, sd, agg_layers, valid_hw=(H, W))
    if agg_layers == 0 and agg3d_layers == 1:      # ONE identity layer is the identity (more layers put a SiLU between)
        assert np.array_equal(vol_id, plain[0]) and np.array_equal(lr_id, plain[1])
    pipe.load_state_dict(sd, autotune=False)
    vol, lr, out = run()
    ref_vol, ref_lr, ref_out = ostereo.disparity(feat[:N], feat[N:], Cf, Dl, pipe.temperature, sd, agg_layers,
                                                 valid_hw=(H, W), agg3d_layers=agg3d_layers)
    assert not np.array_equal(ref_lr, plain[1])               # the layers do something
    if agg_layers == 0:
        assert np.array_equal(vol.view(np.uint32), ref_vol.view(np.uint32))
        assert np.array_equal(lr.view(np.uint32), ref_lr.view(np.uint32))
        assert np.array_equal(out.view(np.uint32), ref_out.view(np.uint32))
    else:
        assert rel_err(vol, ref_vol) <= 1e-3 and rel_err(lr, ref_lr) <= 1e-3 and rel_err(out, ref_out) <= 1e-3
    assert (out[:, :, H:, :] == 0).all() and (out[:, :, :, W:] == 0).all()


# ---- the stereo module's FULL-RESOLUTION mode (north_star's literal D x H x W sizing as a product path) ------------------
@pytest.mark.parametrize('N,Hf,Wf,C_,ld,scale', [(2, 5, 7, 8, 8, 4), (1, 3, 9, 4, 12, 4), (1, 4, 4, 16, 16, 2)])
def test_feat_upsample_bit_exact(N, Hf, Wf, C_, ld, scale, cuda):
    """st_feat_upsample (bilinear, align_corners=False, NHWC) BIT-EXACT against oracle_feat_upsample, and within float
    rounding of torch's interpolate; a channel slice of wider pixels (ld > C) included."""
    lib = _lib.load()
    rng = np.random.RandomState(Hf * 10 + Wf)
    feat = rng.normal(0, 1, (N, Hf, Wf, ld)).astype(np.float32)
    ref = c_oracle.feat_upsample(feat, scale, C_)
    src = torch.from_numpy(feat).to(cuda)
    dst = torch.full((N, Hf * scale, Wf * scale, C_), float('nan'), device=cuda)
    check(lib.st_feat_upsample(ptr(src), N, Hf, Wf, C_, ld, scale, ptr(dst), current_stream()), 'st_feat_upsample')
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    t = torch.nn.functional.interpolate(torch.from_numpy(feat[..., :C_]).permute(0, 3, 1, 2), scale_factor=scale,
                                        mode='bilinear', align_corners=False).permute(0, 2, 3, 1).numpy()
    assert np.abs(got - t).max() <= 1e-6
    assert lib.st_feat_upsample(ptr(src), N, Hf, Wf, 6, ld, scale, ptr(dst), current_stream()) != 0   # C % 4


@pytest.mark.parametrize('agg3d_layers', [0, 1, 2])
def test_stereo_full_resolution_mode_matches_oracle(agg3d_layers, cuda):
    """StereoCostVolume(full_res=True): reduce (1x1, 48 -> 8) -> bilinear x4 -> a D = max_disp level volume at IMAGE
    resolution (slab-wise for D > 128) -> 3-D aggregation -> soft-argmin in pixels -> disp_postp.  The reduced features
    against a float32 matrix product (MFMA summation order: 1e-5 of scale); everything after them BIT FOR BIT against
    oracle/stereo.py::disparity_fullres fed the GPU's own image-resolution features; and the all-oracle path end to end
    within north_star's 1e-3."""
    from oracle import stereo as ostereo
    from stereotracking_amd.pipeline import StereoDensePipeline
    from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict
    N, H, W, D = 2, 88, 152, 48
    pipe = StereoDensePipeline(N, (H, W), 0.375, 0.33, 1, stereo=True, max_disp=D, max_det=32,
                               agg3d_layers=agg3d_layers, full_res=True)
    sm = pipe.stereo_module
    assert sm.full_res and sm.levels == D and pipe.agg_layers == 0
    table = pipe.param_table()
    assert dict(table)['stereo.reduce.weight'] == (8, 48, 1, 1)
    sd = synthetic_state_dict(table, seed=2)
    g = torch.Generator().manual_seed(6)
    for l in range(agg3d_layers):
        sd[f'stereo.agg3d.{l}.weight'] = torch.randn(1, 1, 3, 3, 3, generator=g) * 0.2
        sd[f'stereo.agg3d.{l}.bias'] = torch.randn(1, generator=g) * 0.05
    pipe.load_state_dict(sd, autotune=False)
    batch = synthetic_batch([5, 6], H, W, D)
    img, right = batch['img'].to(cuda), batch['right'].to(cuda)
    Hp, Wp = pipe.height, pipe.width
    vol = torch.full((N, Hp, Wp, D), float('nan'), device=cuda)
    out = torch.full((N, 3, Hp, Wp), float('nan'), device=cuda)
    sm.compute(pipe.det, img, right, (H, W), None, out, cost_out=vol)
    torch.cuda.synchronize()
    feat = pipe.det.tap('stage1_rgb').cpu().numpy()
    Cf = feat.shape[-1]
    fr = sm.full_res_buffers(cuda, N, Hp // 4, Wp // 4)
    red, up = fr['red'].cpu().numpy(), fr['up'].cpu().numpy()
    # (1) the reduction: 2N images, float matrix product
    ref_red = np.concatenate([ostereo.reduce_features(feat[:N], Cf, sd), ostereo.reduce_features(feat[N:], Cf, sd)])
    assert np.abs(red - ref_red).max() <= 1e-5 * max(1.0, np.abs(ref_red).max())
    # (2) the upsampling of the GPU's own reduced features: bit-exact
    assert np.array_equal(up.view(np.uint32), c_oracle.feat_upsample(red, 4).view(np.uint32))
    # (3) volume, 3-D aggregation, soft-argmin, pack on the GPU's own image-resolution features: bit-exact
    ref_vol, ref_disp, ref_out = ostereo.disparity_fullres(None, None, Cf, D, pipe.temperature, sd, agg3d_layers,
                                                           valid_hw=(H, W), upsampled=(up[:N], up[N:]))
    assert np.array_equal(vol.cpu().numpy().view(np.uint32), ref_vol.view(np.uint32))
    assert np.array_equal(fr['disp'].cpu().numpy().view(np.uint32), ref_disp.view(np.uint32))
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref_out.view(np.uint32))
    assert (got[:, :, H:, :] == 0).all() and (got[:, :, :, W:] == 0).all() and np.isfinite(got).all()
    assert (got[:, 0] == got[:, 1]).all() and (got[:, 0] == got[:, 2]).all() and got.max() <= D - 1
    # (4) end to end from the features with the ORACLE's reduction: north_star's float tolerance
    _, _, ref_all = ostereo.disparity_fullres(feat[:N], feat[N:], Cf, D, pipe.temperature, sd, agg3d_layers,
                                              valid_hw=(H, W))
    assert rel_err(got, ref_all) <= 1e-3
    # (5) the whole pipeline consumes it: detector + decode + depth run on the full-resolution disparity
    res = pipe.run(img, right)
    torch.cuda.synchronize()
    assert torch.equal(res['disp_postp'].cpu(), torch.from_numpy(got)) and torch.isfinite(res['head']).all()
    # (6) which kernels ran: with a 3-D layer the cost volume and the first layer are ONE launch (st_costvolume_agg3d);
    # switched off, the two-call form gives the same bits
    sm.timing = True
    out2 = torch.full_like(out, float('nan'))
    sm.compute(pipe.det, img, right, (H, W), None, out2)
    torch.cuda.synchronize()
    stages = sm.pop_full_res_times()
    assert ('cost_volume_agg3d_first' in stages) == (agg3d_layers > 0) and torch.equal(out2, out)
    if agg3d_layers == 1:
        # opt-in (round 6): cost volume + the ONE 3-D layer + soft-argmin as ONE launch (st_costvolume_agg3d_softargmin: the
        # volume is never allocated, written or read back) - the same bits; off by default because it is slower
        sm.fuse_softargmin = True
        out4 = torch.full_like(out, float('nan'))
        sm.compute(pipe.det, img, right, (H, W), None, out4)
        torch.cuda.synchronize()
        assert 'cost_volume_agg3d_softargmin' in sm.pop_full_res_times() and torch.equal(out4, out)
        sm.fuse_softargmin = False
    sm.fuse_first_layer = False
    out3 = torch.full_like(out, float('nan'))
    sm.compute(pipe.det, img, right, (H, W), None, out3)
    torch.cuda.synchronize()
    assert 'cost_volume' in sm.pop_full_res_times() and torch.equal(out3, out)
    sm.timing, sm.fuse_first_layer = False, True
