"""GPU parity of the whole two-branch detector forward (st_detector_forward) against the CPU
PyTorch oracle on the same seeded weights and inputs.  Tolerance: 1e-3 absolute on the raw head
outputs relative to max(1, |ref|) — the bound BASELINE.json's north_star states for floats."""
import pytest
import torch

from oracle.torch_model import OracleDetector, head_to_rows
from stereotracking_amd.engine import HipDetector
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict

pytestmark = pytest.mark.gpu


def build_pair(widen, deepen, N, H, W, seed=0):
    det = HipDetector(N, H, W, widen, deepen, 1)
    sd = synthetic_state_dict(det.param_table(), seed=seed)
    det.load_state_dict(sd)
    ora = OracleDetector(deepen, widen, 1).eval()
    missing, unexpected = ora.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing)
    return det, ora


def rel_err(got, ref):
    return ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()


@pytest.mark.parametrize('widen,N,H,W', [(0.375, 2, 96, 160), (0.5, 1, 192, 320), (0.5, 2, 64, 96)])
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
