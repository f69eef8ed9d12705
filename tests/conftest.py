import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line('markers', 'split: parked split-operand (bf16 x 3) conv instances; need a GPU AND the tools build '
                                       '(ST_LIBRARY=...ablation.so); deselected unless -m names the marker')
    config.addinivalue_line('markers', 'yardstick: timing comparisons (record-only under -m gpu; ST_YARDSTICK_ASSERT=1 '
                                       'turns their speed claims into assertions)')


# Collection order of the GPU run (the driver uses `-x`): oracle-parity files first, multi-process rehearsals LAST, so
# that a harness failure in a subprocess launcher can never again hide the parity tests behind it (round 2:
# GPUTEST_r02 stopped at test_multirank_gpu and 27 parity tests did not run).
_ORDER = ['test_decode_nms_gpu', 'test_stereo_depth_gpu', 'test_conv_gpu', 'test_detector_gpu', 'test_batched_assoc_gpu',
          'test_bench_config_parity_gpu', 'test_shell_gpu', 'test_sequence_gpu']
_LAST = ['test_multirank_gpu', 'test_config3_gpu']


def pytest_collection_modifyitems(session, config, items):
    # `split` tests (parked bf16 x 3 instances, tools build only) run only when the -m expression names them
    if 'split' not in (config.getoption('-m') or ''):
        parked = [it for it in items if it.get_closest_marker('split')]
        if parked:
            config.hook.pytest_deselected(items=parked)
            items[:] = [it for it in items if not it.get_closest_marker('split')]

    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _LAST:
            return (2, _LAST.index(name))
        if name in _ORDER:
            return (0, _ORDER.index(name))
        return (1, 0)
    items.sort(key=key)   # stable: the order inside a file is kept


@pytest.fixture(scope='session')
def stlib():
    from stereotracking_amd import _lib
    return _lib.load()


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but torch.cuda.is_available() is False')
    return torch.device('cuda:0')
