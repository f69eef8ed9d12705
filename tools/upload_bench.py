"""Where does the configs[2] driver's time go?  Host staging / H2D / pack / dense, per 8-frame batch (GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd.pipeline import InflightPipelines
from stereotracking_amd.sequence import HostSequence, RawFrameUploader, detect_shard, synthetic_sequence
from stereotracking_amd.synthetic import synthetic_state_dict

dev = torch.device('cuda:0')
T, H, W, D = 64, 720, 1280, 192
frames = list(synthetic_sequence(T, 6, H, W, D, seed=3))
runner = InflightPipelines(3, 8, (H, W), 0.5, 0.33, 1, stereo=True, max_disp=D, agg_layers=2, max_det=1000)
runner.load_state_dict(synthetic_state_dict(runner.param_table(), seed=0))
up = RawFrameUploader(8, (H, W), dev, True)
seq = HostSequence(frames, True)
for name, src in (('list', frames), ('pinned', seq), ('pinned', seq), ('list', frames)):
    detect_shard(runner, src if name == 'pinned' else src[:8], dev, uploader=up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    detect_shard(runner, src, dev, uploader=up)
    torch.cuda.synchronize()
    print(f'{name}: {T / (time.perf_counter() - t0):.1f} frames/s')
# pieces
t0 = time.perf_counter()
for i in range(0, T, 8):
    up._stage(up.slots[0], frames[i:i + 8])
print(f'host staging: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms / batch')
t0 = time.perf_counter()
for i in range(0, T, 8):
    b = up.upload((seq, i, i + 8))
torch.cuda.synchronize()
print(f'upload + pack (pinned, no dense): {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms / batch')
b = up.upload((seq, 0, 8))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(8):
    runner.submit(b['img'], right=b['right'])
runner.synchronize()
print(f'dense only: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms / batch')
t0 = time.perf_counter()
for i in range(8):
    runner.submit(b['img'], right=b['right'], post=lambda out, ctx: runner.pipes[0].pack_detections(out, scaled=True, n_real=8))
runner.synchronize()
print(f'dense + pack_detections: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms / batch')
