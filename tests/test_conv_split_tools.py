"""Tools build only (`ST_LIBRARY=stereotracking_amd/lib/libstereotrack_hip_ablation.so python -m pytest tests -m split`):
the split-operand (bf16 x 3) instances of the implicit GEMM, variants 50-55.  PARKED since round 5: the
product library refuses them (tests/test_conv_gpu.py::test_product_library_refuses_the_parked_split_instances), so these
tests carry the `split` marker instead of `gpu` and are deselected from every run whose -m expression does not name it
(tests/conftest.py) - no skipped tail under `-m gpu` or `-m "not gpu"`."""
import pytest
import torch
import torch.nn.functional as F

from stereotracking_amd import _lib
from test_conv_gpu import CASES, assert_close, ref_conv, run_conv

pytestmark = pytest.mark.split
SPLIT_BN = {50: 128, 51: 64, 52: 64, 53: 128, 54: 64, 55: 64}


def _need_split_instances():
    if not _lib.load().st_split_instances_available():
        pytest.fail('-m split needs the tools build: ST_LIBRARY=.../libstereotrack_hip_ablation.so')


@pytest.mark.parametrize('variant', [50, 51, 52, 53, 54, 55])
@pytest.mark.parametrize('case', CASES + [(2, 64, 13, 21, 128, 3, 1), (2, 72, 11, 13, 128, 1, 1)])
def test_conv_split_bf16x3_matches_torch(variant, case, cuda):
    """fp32 operands split into three bf16 terms, six exact term products on v_mfma_f32_32x32x16_bf16, fp32 accumulate:
    the same 1e-4-of-scale bar as every exact-fp32 instance (the split is error-free; measured error against float64 is
    BELOW the fp32-input MFMA's, asserted in test_conv_split_is_at_least_as_accurate_as_fp32_mfma)."""
    _need_split_instances()
    N, Cin, H, W, Cout, k, stride = case
    if ((Cout + 31) // 32 * 32) % SPLIT_BN[variant]:
        pytest.skip('tile does not divide Cout')
    torch.manual_seed(variant + sum(case))
    x = torch.randn(N, Cin, H, W) + 1.5   # non-zero mean: a missed zero fill / K tail shows up
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout)
    got, _ = run_conv(x, w, b, stride, k // 2, 1, cuda, variant=variant)
    assert_close(got, ref_conv(x, w, b, stride, k // 2, 1))


def test_conv_split_epilogues(cuda):
    """The split instances share conv_epilogue with the fp32 ones: residual + post_scale, split stores, upsampled store,
    input channel slice."""
    _need_split_instances()
    torch.manual_seed(11)
    x = torch.randn(2, 64, 8, 12)
    w = torch.randn(128, 64, 1, 1) / 8.0
    b = torch.randn(128)
    res = torch.randn(2, 128, 8, 12)
    for v in (50, 51, 52, 53, 54, 55):
        got, _ = run_conv(x, w, b, 1, 0, 1, cuda, variant=v, res=res, post_scale=0.5)
        assert_close(got, ref_conv(x, w, b, 1, 0, 1, res, 0.5))
        got, up = run_conv(x, w, b, 1, 0, 1, cuda, variant=v, split=64, up=True, in_ld=80, in_off=4)
        assert_close(got, ref_conv(x, w, b, 1, 0, 1))
        assert torch.equal(up, F.interpolate(got, scale_factor=2, mode='nearest'))


def test_conv_split_is_at_least_as_accurate_as_fp32_mfma(cuda):
    """The argument for the split instances is accuracy, not only speed: on a head-tower-sized reduction (K = 1152) the
    error against a float64 evaluation must not exceed the exact-fp32 MFMA instance's (measured: about 0.6x)."""
    _need_split_instances()
    torch.manual_seed(3)
    x = torch.randn(2, 128, 24, 40) * torch.randn(2, 128, 24, 40).abs()     # activation-like magnitudes
    w = torch.randn(128, 128, 3, 3) / (128 * 9) ** 0.5
    b = torch.zeros(128)
    ref = ref_conv(x, w, b, 1, 1, 0)
    e32 = (run_conv(x, w, b, 1, 1, 0, cuda, variant=0)[0].double() - ref).abs()
    e3 = (run_conv(x, w, b, 1, 1, 0, cuda, variant=50)[0].double() - ref).abs()
    print(f'fp32 MFMA: max {e32.max():.3e} rms {e32.pow(2).mean().sqrt():.3e}   '
          f'bf16x3: max {e3.max():.3e} rms {e3.pow(2).mean().sqrt():.3e}')
    assert e3.pow(2).mean().sqrt() <= 1.1 * e32.pow(2).mean().sqrt()
    assert e3.max() <= 1.5 * e32.max()
