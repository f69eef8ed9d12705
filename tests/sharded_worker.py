"""Worker of tests/test_multirank_gpu.py: one rank of a world-2 'gloo' rehearsal of BASELINE configs[3] (frames of one
sequence sharded over the ranks, ONE all-gather of the frame records, every rank tracks all frames).  Ranks share
cuda:0 here (the GPU box has one card); under 'gloo' DetectionGatherer moves the records through host memory.  Each rank writes its result to <out_dir>/rank{r}_of{world}.json (argv[2])."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from stereotracking_amd.motion import KalmanFilter  # noqa: E402
from stereotracking_amd.pipeline import InflightPipelines  # noqa: E402
from stereotracking_amd.sequence import run_sharded_sequence, synthetic_sequence  # noqa: E402
from stereotracking_amd.synthetic import synthetic_state_dict  # noqa: E402
from stereotracking_amd.trackers import OCSORTTracker_Disparity  # noqa: E402


class _Model:
    motion = KalmanFilter()


def main():
    T = int(sys.argv[1])
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda:0')
    runner = InflightPipelines(3, 4, (80, 160), 0.375, 0.33, 1, stereo=True, max_disp=32, max_det=256, agg_layers=1)
    sd = synthetic_state_dict(runner.param_table(), seed=9, prior_prob=0.2, logit_std=2.5)
    runner.load_state_dict(sd, autotune=False)
    frames = list(synthetic_sequence(T, 4, 80, 160, 32, seed=6))
    trk = OCSORTTracker_Disparity(obj_score_thr=0.02, init_track_thr=0.03, weight_iou_with_det_scores=False,
                                  match_iou_thr=0.1, num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3,
                                  num_frames_retain=30)

    res = run_sharded_sequence(runner, frames, trk, _Model(), dev)
    rec = dict(rank=rank, world=world, ids=[r.instances_id.tolist() for r in res], nboxes=[len(r) for r in res],
               box_sum=[float(r.bboxes.double().sum()) for r in res])
    # One FILE per rank: both ranks share the launcher's stdout pipe, and a write above PIPE_BUF (4096 B) is not
    # atomic, so two long JSON lines can interleave.  Only a short digest goes to stdout.
    out_dir = sys.argv[2]
    tmp = os.path.join(out_dir, f'rank{rank}_of{world}.json.tmp')
    with open(tmp, 'w') as f:
        json.dump(rec, f)
    os.replace(tmp, tmp[:-4])
    print(f'sharded_worker rank {rank}/{world}: {len(res)} frames, {sum(rec["nboxes"])} boxes', flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
