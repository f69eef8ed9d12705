"""Multi-GPU layer of the dense path: one process per GPU, frames (or whole videos) sharded across
ranks, ONE all-gather of the fixed-size detection buffer per shard (RCCL over xGMI when the backend
is 'nccl'; 'gloo' in the CPU tests), then the CPU tracker consumes the frames in order.

The reference only shards whole videos and never communicates inside the loop
(mmtrack/datasets/samplers/video_sampler.py:25-70: np.array_split of the video list over ranks;
collectives only in evaluation, mot_drone_metrics.py:336-358).  north_star adds frame sharding of
one long sequence; the dense path is stateless per frame (ocsort_disparity.py:73-83), so the
detections — and therefore the track ids — are identical to the sequential run.
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_videos(num_videos, rank=None, world_size=None):
    """Reference sharding: contiguous blocks of whole videos (video_sampler.py:62-70)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return np.array_split(np.arange(num_videos), world_size)[rank].tolist()


def shard_frames(num_frames, rank=None, world_size=None):
    """Contiguous chunk of frames of ONE sequence for this rank (equal sizes, last ranks padded):
    returns (start, stop, chunk) with chunk = ceil(num_frames / world_size)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    chunk = (num_frames + world_size - 1) // world_size
    start = min(rank * chunk, num_frames)
    return start, min(start + chunk, num_frames), chunk


def gather_detections(local, counts=None):
    """local: (F_local, M + 1, 8) frame records of this rank's frames (pack_detections: header row with the
    true count + M rows x1,y1,x2,y2,score,label,depth,scale) - or any fixed-size buffer.
    Returns (world*F_local, ...) [and gathered `counts` when given] on every rank, in rank = frame order.
    One collective per shard, never per frame: the payload is KBs, the cost is launch latency."""
    rank, w = world()
    if w == 1:
        return (local, counts) if counts is not None else local
    out = torch.empty((w * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if counts is None:
        return out
    cout = torch.empty(w * counts.shape[0], dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(cout, counts.contiguous())
    return out, cout


class DetectionOverflow(RuntimeError):
    """A frame kept more boxes than the fixed-size detection buffer holds.  The reference applies no cap
    (yolox_style=True), so dropping the surplus would change the tracker's input: fail loudly instead."""


def unpack_frame(record, frame=None):
    """One frame record (M + 1, 8) (StereoDensePipeline.pack_detections) -> dict of tensors for the tracker.
    Raises DetectionOverflow when the frame kept more boxes than the buffer has rows."""
    k, cap = int(record[0, 0]), int(record[0, 1])
    if k > cap:
        raise DetectionOverflow(f'frame {frame}: {k} detections kept but the detection buffer has {cap} rows; '
                                f'build the pipeline with a larger max_det')
    b = record[1:1 + k]
    return dict(bboxes=b[:, :4], scores=b[:, 4], labels=b[:, 5].long(), depth=b[:, 6], scales=b[:, 7])


def record_counts(records):
    """(F, M + 1, 8) records -> (counts (F,) int64, capacity M)."""
    return records[:, 0, 0].long(), records.shape[1] - 1
