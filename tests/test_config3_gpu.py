"""GPU: BASELINE.json configs[3] at its STATED size through the HIP path - a 512-frame synthetic 1280x720 sequence,
D=192, full two-branch YOLOX-s, sharded over the ranks with ONE all-gather of the frame records before the (CPU) tracker
step (sequence.run_sharded_sequence; reference partitioning mmtrack/datasets/samplers/video_sampler.py:25-70, launcher
handling tools/test.py:33-41; the dense path is stateless per frame, mmtrack/models/mot/ocsort_disparity.py:73-83).

What one card can show of it: world 1 against world 2 (two ranks sharing cuda:0, the collective over gloo - the 8-card
RCCL exchange itself is the driver's run): track ids and boxes of all 512 frames EQUAL on every rank, the ragged
509-frame split (255 + 254, both shards padded to 256) alike, and the first 24 frames of the sharded run against the
oracle pipeline + ORACLE tracker rows of tests/golden/config2_sequence.npz (same weights, same sequence, shipped
thresholds)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'config3_worker.py')
GOLD = os.path.join(ROOT, 'tests', 'golden', 'config2_sequence.npz')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(world, T, out_dir):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, WORKER, str(T), out_dir]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), WORKER, str(T), out_dir]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    outs = []
    for r in range(world):
        with open(os.path.join(out_dir, f'c3_T{T}_rank{r}_of{world}.json')) as f:
            outs.append(json.load(f))
    return outs


@pytest.fixture(scope='module')
def single512(cuda, tmp_path_factory):
    return _run(1, 512, str(tmp_path_factory.mktemp('c3_w1')))[0]


def test_config3_512_frames_first_24_equal_the_oracle_tracker_rows(single512):
    """Frames 0..23 of the 512-frame run are the fixture's sequence (the generator is causal, so is the tracker): ids
    identical IN ORDER to the oracle tracker's, scaled boxes within 1e-3 of the oracle pipeline's."""
    g = np.load(GOLD)
    ref = g['tracks_shipped']            # rows [t, id, scaled box (4), score, depth, scale]
    T0 = int(g['T'])
    assert single512['T'] == 512 and len(single512['ids']) == 512 and len(single512['head']) == T0
    rows = 0
    for t in range(T0):
        r = ref[ref[:, 0] == t]
        h = single512['head'][t]
        assert h['ids'] == r[:, 1].astype(np.int64).tolist(), f'frame {t}: track ids differ from the oracle tracker'
        if len(r):
            b, rb = np.asarray(h['scaled_boxes'], np.float64).reshape(-1, 4), r[:, 2:6].astype(np.float64)
            ext = np.maximum(rb[:, 2:4] - rb[:, 0:2], 1.0).max(axis=1, keepdims=True)   # an edge inherits the extent's error
            assert float((np.abs(b - rb) / np.maximum(np.maximum(1.0, np.abs(rb)), ext)).max()) <= 1e-3
            assert float(np.abs(np.asarray(h['scores']) - r[:, 6]).max()) <= 1e-3
        rows += len(r)
    assert rows == len(ref) and rows > 100
    # the sequence goes on: tracks exist well beyond the fixture's horizon
    assert sum(single512['nboxes'][T0:]) > 10 * 24 and max(max(i) for i in single512['ids'] if i) > 50


@pytest.mark.parametrize('T', [512, 509])
def test_config3_sharded_world2_equals_single_process(T, single512, cuda, tmp_path):
    ref = single512 if T == 512 else _run(1, T, str(tmp_path))[0]
    outs = _run(2, T, str(tmp_path))
    assert sorted(o['rank'] for o in outs) == [0, 1]
    for o in outs:      # every rank tracked ALL frames from the gathered records: identical to the unsharded run
        assert o['world'] == 2 and len(o['ids']) == T
        assert o['ids'] == ref['ids'] and o['nboxes'] == ref['nboxes'] and o['box_sum'] == ref['box_sum']
    if T == 509:        # a prefix of the same sequence: the first 509 frames of the 512-frame run
        assert ref['ids'] == single512['ids'][:509] and ref['box_sum'] == single512['box_sum'][:509]
