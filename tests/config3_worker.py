"""Worker of tests/test_config3_gpu.py: one rank of BASELINE configs[3] AT ITS STATED SIZE through the HIP path - a
512-frame (or ragged 509-frame) synthetic 1280x720 sequence, D=192, full two-branch YOLOX-s + 2 aggregation convs,
sharded contiguously over the ranks (reference partitioning: mmtrack/datasets/samplers/video_sampler.py:25-70; the dense
path is stateless per frame, mmtrack/models/mot/ocsort_disparity.py:73-83), ONE all-gather of the frame records, every
rank tracks all frames with the SHIPPED tracker thresholds.  On the one-card GPU box the ranks share cuda:0 and the
collective runs over gloo (DetectionGatherer moves the records through host memory); the N-card RCCL run is the driver's.
Weights / sequence / thresholds are those of tests/golden/config2_sequence.npz, so the first 24 frames can be held
against the oracle pipeline + oracle tracker rows of that fixture.
usage: config3_worker.py <num_frames> <out_dir>   -> <out_dir>/c3_T<frames>_rank<r>_of<world>.json"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from stereotracking_amd.config import Config  # noqa: E402
from stereotracking_amd.mot import scale_bbox  # noqa: E402
from stereotracking_amd.motion import KalmanFilter  # noqa: E402
from stereotracking_amd.pipeline import InflightPipelines  # noqa: E402
from stereotracking_amd.sequence import run_sharded_sequence, synthetic_sequence  # noqa: E402
from stereotracking_amd.synthetic import synthetic_state_dict  # noqa: E402
from stereotracking_amd.trackers import OCSORTTracker_Disparity  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'config2_sequence.npz')
CFG = os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'stereo_yolox_s_mot_airdrone_costvolume.py')


class _Model:
    motion = KalmanFilter()


def main():
    T, out_dir = int(sys.argv[1]), sys.argv[2]
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    g = np.load(GOLD)
    H, W, D = int(g['H']), int(g['W']), int(g['D'])
    dev = torch.device('cuda:0')
    cfg = Config.fromfile(CFG)
    runner = InflightPipelines(3, 8, (H, W), 0.5, 0.33, 1, stereo=True, max_disp=D, max_det=int(g['max_det']),
                               agg_layers=int(g['AGG']))
    sd = synthetic_state_dict(runner.param_table(), seed=int(g['weight_seed']), prior_prob=float(g['prior_prob']),
                              logit_std=float(g['logit_std']))
    runner.load_state_dict(sd)          # the committed tuning plan (pipeline.default_tuning_cache()), as bench.py
    tk = dict(cfg.model.tracker)
    tk.pop('type')
    trk = OCSORTTracker_Disparity(**tk)  # shipped thresholds
    t0 = time.perf_counter()
    # the first int(g['T']) frames of this generator ARE the fixture's sequence (one RandomState, advanced frame by frame)
    frames = list(synthetic_sequence(T, int(g['objects']), H, W, D, seed=int(g['seq_seed']), smooth=3))
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = run_sharded_sequence(runner, frames, trk, _Model(), dev)
    torch.cuda.synchronize()
    t_run = time.perf_counter() - t0
    head = []
    for r in res[:int(g['T'])]:          # full rows of the frames the fixture covers
        head.append(dict(ids=r.instances_id.tolist(), scaled_boxes=scale_bbox(r.bboxes, r.scales).double().tolist(),
                         scores=r.scores.double().tolist()))
    rec = dict(rank=rank, world=world, T=T, ids=[r.instances_id.tolist() for r in res], nboxes=[len(r) for r in res],
               box_sum=[float(r.bboxes.double().sum()) for r in res], head=head, seconds_generate=round(t_gen, 2),
               seconds_detect_gather_track=round(t_run, 2), frames_per_s=round(T / t_run, 1))
    tmp = os.path.join(out_dir, f'c3_T{T}_rank{rank}_of{world}.json.tmp')   # one FILE per rank (ranks share a stdout pipe)
    with open(tmp, 'w') as f:
        json.dump(rec, f)
    os.replace(tmp, tmp[:-4])
    print(f'config3_worker rank {rank}/{world}: {len(res)} frames, {sum(rec["nboxes"])} track rows, '
          f'{rec["frames_per_s"]} frames/s', flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
