"""GPU: the multi-rank code paths rehearsed with world_size 2 on the one card of the GPU box ('gloo' backend, the
collective goes through host memory; the 8-GPU RCCL run itself is the driver's).  Covers BASELINE configs[3]'s driver
(run_sharded_sequence: ragged shard, 3 in-flight contexts per rank, ONE all-gather of self-describing frame records,
identical collective order on both ranks) and bench.py's --gpus N path."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(nproc, script_args, env_extra=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(env_extra or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + script_args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout


def _json_lines(stdout):
    """Complete JSON objects printed on lines of their own (anything else on stdout is ignored)."""
    out = []
    for line in stdout.splitlines():
        if line.startswith('{') and line.rstrip().endswith('}'):
            try:
                out.append(json.loads(line))
            except json.JSONDecodeError:
                pass
    return out


def _read_rank(out_dir, rank, world):
    with open(os.path.join(out_dir, f'rank{rank}_of{world}.json')) as f:
        return json.load(f)


def test_sharded_sequence_world2_equals_single_process(cuda, tmp_path):
    T = 13   # ragged: 7 + 6 frames, i.e. 2 batches of 4 per rank with padding on both
    worker = os.path.join(ROOT, 'tests', 'sharded_worker.py')
    out_dir = str(tmp_path)
    single = subprocess.run([sys.executable, worker, str(T), out_dir], cwd=ROOT, capture_output=True, text=True,
                            timeout=600)
    assert single.returncode == 0, single.stderr[-3000:]
    ref = _read_rank(out_dir, 0, 1)
    assert ref['world'] == 1 and sum(ref['nboxes']) > T
    _launch(2, [worker, str(T), out_dir])
    outs = [_read_rank(out_dir, r, 2) for r in (0, 1)]   # one file per rank: ranks never share a pipe for results
    assert sorted(o['rank'] for o in outs) == [0, 1]
    for o in outs:   # every rank tracked ALL frames from the gathered records: identical to the unsharded run
        assert o['ids'] == ref['ids'] and o['nboxes'] == ref['nboxes'] and o['box_sum'] == ref['box_sum']


def test_bench_multirank_path_world2(cuda):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, one rank per process), with the
    collective rehearsed over gloo: the barrier / max-over-ranks timing, the per-step all-gather of the frame records
    from the context streams and the record-vs-local-count check all run; rank 0 prints ONE JSON line."""
    outs = _json_lines(_launch(2, [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
                                   '--no-cpu-baseline', '--no-test-step', '--sustain-seconds', '0.4'],
                               dict(ST_BENCH_BACKEND='gloo')))
    assert len(outs) == 1
    line = outs[0]
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['global_batch'] == 16 and line['config']['detections_overflow'] is False
    assert 'roofline' in line and 'cpu_baseline' not in line


def test_bench_gpus2_without_launcher_starts_two_ranks(cuda):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the form the driver uses for --gpus 1): the parent -
    which makes no GPU call - starts the two ranks itself, relays the ONE JSON line and the exit status.  On this
    one-card box the ranks share cuda:0 over gloo (ST_BENCH_BACKEND=gloo); on the RCCL backend the same command with
    fewer than N devices is refused (checked here too: never a silent 1-rank line).  VERDICT r5 #1."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', ST_BENCH_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    args = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--no-cpu-baseline',
            '--no-test-step', '--sustain-seconds', '0', '--no-secondary-legs']
    p = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1, p.stdout[:600]
    line = _json_lines(p.stdout)[0]
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 16 and line['value'] > 0
    assert sorted(r['rank'] for r in line['config']['ranks_seen']) == [0, 1]
    import torch
    if torch.cuda.device_count() < 2:
        env.pop('ST_BENCH_BACKEND')
        p = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 2 and p.stdout.strip() == '' and 'refusing' in p.stderr


def test_bench_rccl_process_group_of_one_rank_executes_the_collectives(cuda):
    """What one GPU can execute of the RCCL path: bench.py with a process group of ONE rank on the 'nccl' backend
    (ST_BENCH_WORLD1_PG=1) - RCCL communicator init on the device, one `all_gather_into_tensor` of the frame records per
    step issued by dist.DetectionGatherer on its communication stream behind the producing context's event, the barriers
    and the all-reduce of the timing; the gathered records must equal the local counts (bench.py checks it and aborts
    otherwise).  The N > 1 exchange itself needs N devices (the driver's 8-GPU run); this pins that the code path RUNS
    (reference collectives: mmtrack/evaluation/metrics/mot_drone_metrics.py:336-358)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', ST_BENCH_WORLD1_PG='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-cpu-baseline',
                        '--no-test-step', '--sustain-seconds', '0'], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    # stdout is exactly ONE line (RCCL prints a version banner to file descriptor 1 at communicator init: bench.py sends
    # everything but its JSON to stderr, or the driver's parser would meet five lines of banner first)
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1, p.stdout[:600]
    outs = _json_lines(p.stdout)
    assert len(outs) == 1
    line = outs[0]
    assert line['n_gpus'] == 1 and line['value'] > 0
    assert 'nccl' in line['config']['parallelism'] and line['config']['collectives_issued'] >= 8
    assert line['config']['detections_overflow'] is False
