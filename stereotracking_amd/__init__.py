"""MI355X-native dense hot path of the stereo tracker."""
import os
import sys
import warnings

WANTED_HW_QUEUES = 8
"""Hardware queues the in-flight contexts want (one per context; bench.py runs 4, the MOT shell 3)."""


def _hip_runtime_is_up():
    """True when the host process initialised the HIP runtime before this import (torch already imported AND its CUDA
    state initialised): environment variables set from here on are no longer read."""
    t = sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


def _setup_runtime_env():
    # One hardware queue per HIP stream: ROCm maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues
    # (default 4), and two in-flight contexts that share a queue serialise behind each other (4 contexts: 1729 pairs/s
    # on 4 queues, 1840 on 8).  The HIP runtime reads it when it initialises, i.e. this import has to come before the
    # first CUDA call of the process; an explicit setting in the environment wins.
    # HIP_FORCE_DEV_KERNARG: kernel arguments in device memory instead of host-coherent memory, ~2 us less per launch
    # (1538 -> 1568 pairs/s with one context, +0.3 % with four); same rule.
    late = _hip_runtime_is_up()
    for name, val in (('GPU_MAX_HW_QUEUES', str(WANTED_HW_QUEUES)), ('HIP_FORCE_DEV_KERNARG', '1')):
        if name not in os.environ:
            if late:
                warnings.warn(
                    f'stereotracking_amd imported after the HIP runtime was initialised: {name}={val} cannot take '
                    f'effect any more (with the default 4 hardware queues a 4th in-flight context shares a queue: '
                    f'~6 % lower throughput).  Import stereotracking_amd before the first torch.cuda call, or export '
                    f'{name}={val}.', RuntimeWarning, stacklevel=3)
            else:
                os.environ[name] = val


def effective_hw_queues():
    """dict(value, late_import): the GPU_MAX_HW_QUEUES value the HIP runtime works with as far as this process can
    tell - the environment's value if it was present when the runtime came up, else the runtime default of 4.
    Reported in bench records."""
    v = os.environ.get('GPU_MAX_HW_QUEUES') if not _LATE_IMPORT else _ENV_AT_IMPORT.get('GPU_MAX_HW_QUEUES')
    return dict(value=int(v) if v and v.isdigit() else 4, late_import=_LATE_IMPORT)


_ENV_AT_IMPORT = {k: os.environ.get(k) for k in ('GPU_MAX_HW_QUEUES', 'HIP_FORCE_DEV_KERNARG')}
_LATE_IMPORT = _hip_runtime_is_up()
_setup_runtime_env()
