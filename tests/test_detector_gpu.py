"""GPU parity of the whole two-branch detector forward (st_detector_forward) against the CPU
PyTorch oracle on the same seeded weights and inputs.  Tolerance: 1e-3 absolute on the raw head
outputs relative to max(1, |ref|) — the bound BASELINE.json's north_star states for floats."""
import pytest
import torch

from oracle.torch_model import OracleDetector, head_to_rows
from stereotracking_amd.engine import HipDetector
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict

pytestmark = pytest.mark.gpu


def build_pair(widen, deepen, N, H, W, seed=0, rgb_only=False):
    det = HipDetector(N, H, W, widen, deepen, 1, rgb_only=rgb_only)
    sd = synthetic_state_dict(det.param_table(), seed=seed)
    det.load_state_dict(sd)
    ora = OracleDetector(deepen, widen, 1, rgb_only=rgb_only).eval()
    missing, unexpected = ora.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing)
    return det, ora


from parity_utils import explain_kept_difference, rel_err  # noqa: E402


@pytest.mark.parametrize('widen,N,H,W', [(0.375, 2, 96, 160), (0.5, 1, 192, 320), (0.5, 2, 64, 96),
                                         (1.0, 1, 64, 96),     # stem 64 wide: two MFMA column blocks in the fused stem
                                         (1.25, 1, 64, 64)])   # stem 80 wide: focus_pack + generic conv fallback
def test_detector_head_parity(widen, N, H, W, cuda):
    det, ora = build_pair(widen, 0.33, N, H, W)
    batch = synthetic_batch(list(range(N)), H - 16, W, 64)
    assert batch['img'].shape[-2:] == (H, W)
    with torch.no_grad():
        ref_rows = head_to_rows(*ora(batch))
        ref64_rows = head_to_rows(*ora.double()({k: v.double() for k, v in batch.items()}))
    head = det.forward(batch['img'].to(cuda), batch['disp_postp'].to(cuda))
    torch.cuda.synchronize()
    for lvl, (rows, ref, ref64) in enumerate(zip(det.head_levels(head), ref_rows, ref64_rows)):
        got = rows[..., :6].cpu()
        e_gpu, e_cpu = rel_err(got.double(), ref64), rel_err(ref.double(), ref64)
        print(f'level {lvl}: gpu-vs-fp64 {e_gpu:.2e}  cpu32-vs-fp64 {e_cpu:.2e}  gpu-vs-cpu32 {rel_err(got, ref):.2e}')
        assert rel_err(got, ref) <= 1e-3
        assert e_gpu <= 1e-3


@pytest.mark.parametrize('widen,N,H,W', [(0.5, 2, 192, 320), (0.375, 1, 96, 160)])
def test_rgb_only_detector_head_and_taps_parity(widen, N, H, W, cuda):
    """The reference's SECOND stereo config (configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone.py:40-42):
    `mmyolo.YOLODetector` over `mmtrack.CSPDarknet` (csp_darknet.py:8-13, forward reads x['img'] only) = the launch plan
    without disp_stem / disp_stage1 and without the average.  Head rows and backbone taps against the oracle with the
    branch disabled (fp32 and fp64); the disparity pointer is NULL for this plan, and a disparity that IS passed changes
    nothing."""
    det, ora = build_pair(widen, 0.33, N, H, W, seed=4, rgb_only=True)
    assert not any('disp_' in n for n, _ in det.param_table())
    batch = synthetic_batch(list(range(N)), H - 16, W, 64)
    with torch.no_grad():
        ref_rows = head_to_rows(*ora(batch))
        feats = ora.backbone(batch)
        ref64_rows = head_to_rows(*ora.double()({k: v.double() for k, v in batch.items()}))
    head = det.forward(batch['img'].to(cuda), None).clone()
    torch.cuda.synchronize()
    for name, ref in zip(('stage2', 'stage3', 'stage4'), feats):
        assert rel_err(det.tap(name).cpu().permute(0, 3, 1, 2), ref) <= 1e-3, name
    for rows, ref, ref64 in zip(det.head_levels(head), ref_rows, ref64_rows):
        got = rows[..., :6].cpu()
        assert rel_err(got, ref) <= 1e-3
        assert rel_err(got.double(), ref64) <= 1e-3
    head2 = det.forward(batch['img'].to(cuda), batch['disp_postp'].to(cuda))
    torch.cuda.synchronize()
    for r1, r2 in zip(det.head_levels(head), det.head_levels(head2)):     # (columns 6, 7 of a head row are padding)
        assert torch.equal(r1[..., :6], r2[..., :6])
    # and it is NOT the two-branch result: the same image-branch weights inside the two-branch plan give another head
    det2, _ = build_pair(widen, 0.33, N, H, W, seed=4)
    head3 = det2.forward(batch['img'].to(cuda), batch['disp_postp'].to(cuda))
    torch.cuda.synchronize()
    assert not torch.equal(det2.head_levels(head3)[0][..., :6], det.head_levels(head)[0][..., :6])


@pytest.mark.parametrize('nc', [3, 5, 80])
def test_detector_head_parity_several_classes(nc, cuda):
    """num_classes > 1: the prediction convs run un-fused into head rows of st_head_row_floats(nc) floats
    (8 up to 3 classes, nc + 5 rounded up to 4 beyond: base config _base_/yolox_s_8x8_mmyolo.py:40-51 with another
    num_classes); class logits, box and objectness columns within 1e-3 of the oracle head."""
    det = HipDetector(2, 96, 160, 0.375, 0.33, nc)
    sd = synthetic_state_dict(det.param_table(), seed=nc)
    det.load_state_dict(sd)
    ora = OracleDetector(0.33, 0.375, nc).eval()
    missing, unexpected = ora.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing)
    batch = synthetic_batch([0, 1], 80, 160, 64)
    with torch.no_grad():
        ref_rows = head_to_rows(*ora(batch))
    head = det.forward(batch['img'].to(cuda), batch['disp_postp'].to(cuda))
    torch.cuda.synchronize()
    assert det.head_row == (8 if nc <= 3 else (nc + 5 + 3) // 4 * 4)
    for rows, ref in zip(det.head_levels(head), ref_rows):
        assert rows.shape[-1] == det.head_row and ref.shape[-1] == nc + 5
        assert rel_err(rows[..., :nc + 5].cpu(), ref) <= 1e-3
    cls, reg, obj = det.head_nchw(head)
    assert cls[0].shape[1] == nc and reg[0].shape[1] == 4 and obj[0].shape[1] == 1


def test_backbone_taps_match_oracle(cuda):
    det, ora = build_pair(0.375, 0.33, 1, 96, 160)
    batch = synthetic_batch([3], 80, 160, 64)
    with torch.no_grad():
        feats = ora.backbone(batch)
    det.forward(batch['img'].to(cuda), batch['disp_postp'].to(cuda))
    torch.cuda.synchronize()
    for name, ref in zip(('stage2', 'stage3', 'stage4'), feats):
        got = det.tap(name).cpu().permute(0, 3, 1, 2)
        assert got.shape == ref.shape
        assert rel_err(got, ref) <= 1e-3, name


def test_forward_rejects_cpu_tensors(cuda):
    det, _ = build_pair(0.375, 0.33, 1, 64, 96)
    x = torch.zeros(1, 3, 64, 96)
    with pytest.raises(RuntimeError, match='CUDA'):
        det.forward(x, x)


def test_full_size_pipeline_parity_and_batch_invariance(cuda):
    """BASELINE.json configs[1] geometry (1280x720 -> 736x1280, D=192, full YOLOX-s, 2 aggregation convs):
    pair 0 of a batch of 2 through the whole HIP pipeline against the oracle composition on the same inputs
    (head + disparity within 1e-3; decode/NMS indices bit-exact on the GPU's own head), and the
    size-independent property the path offers: frames are independent, so the batch-of-2 result of each pair
    is bit-identical to running that pair alone."""
    import numpy as np
    from oracle import c_oracle, stereo as ostereo
    from stereotracking_amd.pipeline import StereoDensePipeline
    H, W, D, AGG = 720, 1280, 192, 2
    pipe = StereoDensePipeline(2, (H, W), 0.5, 0.33, 1, stereo=True, max_disp=D, max_det=300, agg_layers=AGG)
    sd = synthetic_state_dict(pipe.param_table(), seed=0, prior_prob=0.05, logit_std=1.5)
    pipe.load_state_dict(sd, autotune=False)
    batch = synthetic_batch([11, 12], H, W, D)
    img, right = batch['img'].to(cuda), batch['right'].to(cuda)
    out = {k: v.clone() for k, v in pipe.run(img, right).items()}
    torch.cuda.synchronize()
    assert int(out['counts'].min()) > 0

    ora = OracleDetector(0.33, 0.5, 1).eval()
    ora.load_state_dict(sd, strict=False)
    with torch.no_grad():
        fl = ora.backbone.stage1_features(batch['img'][:1]).permute(0, 2, 3, 1).contiguous().numpy()
        fr = ora.backbone.stage1_features(batch['right'][:1]).permute(0, 2, 3, 1).contiguous().numpy()
        disp = torch.from_numpy(ostereo.disparity(fl, fr, fl.shape[-1], D // 4, pipe.temperature, sd, AGG,
                                                  valid_hw=(H, W))[2])
        # stage-wise, like every other parity test here: the detector oracle consumes the GPU's own disparity
        # (soft-argmin at temperature 32 amplifies fp32 rounding of the features; the head would inherit it)
        rows = head_to_rows(*ora(dict(img=batch['img'][:1], disp_postp=out['disp_postp'][:1].cpu())))
    err_d = rel_err(out['disp_postp'][:1].cpu(), disp)      # per element: |a - b| <= 1e-3 * max(1, |b|)
    print(f'full-size disparity max rel err {err_d:.3e}')
    assert err_d <= 1e-3
    assert (out['disp_postp'][:, :, H:] == 0).all()
    for got, ref in zip(pipe.det.head_levels(out['head']), rows):
        assert rel_err(got[:1, :, :6].cpu(), ref) <= 1e-3
    ref = c_oracle.decode_nms(out['head'].cpu().numpy(), 2, pipe.det.levels, pipe.score_thr, pipe.iou_thr,
                              pipe.max_det, (H, W))
    assert np.array_equal(out['counts'].cpu().numpy(), ref[4])
    for n in range(2):
        k = min(int(ref[4][n]), pipe.max_det)
        assert np.array_equal(out['prior_idx'][n, :k].cpu().numpy(), ref[3][n, :k])
        assert np.array_equal(out['boxes'][n, :k].cpu().numpy(), ref[0][n, :k])
        assert np.array_equal(out['scores'][n, :k].cpu().numpy(), ref[1][n, :k])

    # batch invariance: each pair alone (batch of 1, same tile variants are NOT guaranteed -> compare the
    # bit-exact stages on indices, floats within 1e-3)
    solo = StereoDensePipeline(1, (H, W), 0.5, 0.33, 1, stereo=True, max_disp=D, max_det=300, agg_layers=AGG)
    solo.load_state_dict(sd, autotune=False)
    for n in range(2):
        o = solo.run(img[n:n + 1], right[n:n + 1])
        torch.cuda.synchronize()
        assert rel_err(o['disp_postp'][0].cpu(), out['disp_postp'][n].cpu()) <= 1e-3
        # the two launch plans (N=1, N=2) may pick different tiles: kept SETS must agree up to decisions whose
        # margin is inside the float noise of the path (explained by the reference run's own margins)
        ka = o['prior_idx'][0, :int(o['counts'][0])].cpu().numpy()
        kb = out['prior_idx'][n, :int(out['counts'][n])].cpu().numpy()
        if set(ka.tolist()) != set(kb.tolist()):
            rows_n = [r[n:n + 1, :, :6].cpu() for r in pipe.det.head_levels(out['head'])]
            ex = explain_kept_difference(rows_n, pipe.det.levels, ka, kb, pipe.score_thr, pipe.iou_thr)
            assert ex['unexplained'] == [] and ex['affected_frac'] <= 0.10, ex


def test_config0_tiny_pair_against_cpu_oracle(cuda):
    """BASELINE.json configs[0]: ONE synthetic 1280x720 stereo pair, D=64, tiny CSPDarknet (widen 0.375, deepen
    0.33) - the reference's own CPU-runnable case.  The whole HIP pipeline against the CPU
    PyTorch / C oracle composition: disparity and head floats within 1e-3, kept prior indices / boxes / scores
    bit-exact on the GPU's own head, per-box depth decisions equal."""
    import numpy as np
    from oracle import c_oracle, depth as odepth, stereo as ostereo
    from stereotracking_amd.pipeline import StereoDensePipeline
    H, W, D = 720, 1280, 64
    pipe = StereoDensePipeline(1, (H, W), 0.375, 0.33, 1, stereo=True, max_disp=D, max_det=300, agg_layers=2)
    sd = synthetic_state_dict(pipe.param_table(), seed=0, prior_prob=0.05, logit_std=1.5)
    pipe.load_state_dict(sd, autotune=False)
    batch = synthetic_batch([0], H, W, D)
    out = pipe.run(batch['img'].to(cuda), batch['right'].to(cuda))
    torch.cuda.synchronize()
    ora = OracleDetector(0.33, 0.375, 1).eval()
    ora.load_state_dict(sd, strict=False)
    with torch.no_grad():
        fl = ora.backbone.stage1_features(batch['img']).permute(0, 2, 3, 1).contiguous().numpy()
        fr = ora.backbone.stage1_features(batch['right']).permute(0, 2, 3, 1).contiguous().numpy()
        disp = torch.from_numpy(ostereo.disparity(fl, fr, fl.shape[-1], D // 4, pipe.temperature, sd, 2,
                                                  valid_hw=(H, W))[2])
        rows = head_to_rows(*ora(dict(img=batch['img'], disp_postp=out['disp_postp'].cpu())))
    assert rel_err(out['disp_postp'].cpu(), disp) <= 1e-3
    for got, ref in zip(pipe.det.head_levels(out['head']), rows):
        assert rel_err(got[..., :6].cpu(), ref) <= 1e-3
    ref = c_oracle.decode_nms(out['head'].cpu().numpy(), 1, pipe.det.levels, pipe.score_thr, pipe.iou_thr,
                              pipe.max_det, (H, W))
    k = min(int(ref[4][0]), pipe.max_det)
    assert k > 0 and int(out['counts'][0]) == int(ref[4][0])
    assert np.array_equal(out['prior_idx'][0, :k].cpu().numpy(), ref[3][0, :k])
    assert np.array_equal(out['boxes'][0, :k].cpu().numpy(), ref[0][0, :k])
    assert np.array_equal(out['scores'][0, :k].cpu().numpy(), ref[1][0, :k])
    d_ref, s_ref, _ = odepth.bbox_postp_depth(torch.from_numpy(ref[0][0, :k]), out['disp_postp'].cpu())
    d_ref = np.array([float(v) for v in d_ref], np.float64)
    d_got = out['depth'][0, :k].cpu().double().numpy()
    assert np.array_equal(np.isnan(d_got), np.isnan(d_ref)) and np.array_equal(d_got == -1, d_ref == -1)
    ok = ~np.isnan(d_ref) & (d_ref != -1)
    assert rel_err(d_got[ok], d_ref[ok]) <= 1e-3
