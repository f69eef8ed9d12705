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


class DetectionGatherer:
    """Every detection all-gather of a process goes through ONE object and ONE communication stream.

    The pipeline contexts run on HIP streams of their own and finish in data-dependent order, but a collective must be
    issued in the SAME order on every rank.  The order here is structural: `gather()` is called from the host in
    program order (step i of every rank, whatever context it ran on), the collective is enqueued on the single
    communication stream behind an event recorded on the producing stream, and a sequence number is kept so that
    tests can assert that all ranks issued the same number of collectives.  RCCL then sees one in-order queue per
    rank, independent of how the contexts interleave on the device.  Payload: the fixed-size frame records
    ((max_det + 1) x 8 fp32 per frame, 256 KB per 8-frame shard) - latency-bound over xGMI, so one collective per
    shard, never per frame.  'gloo' (CPU rehearsal) gathers through host memory."""

    def __init__(self, device=None, single_rank_collective=False):
        """single_rank_collective: with a process group of ONE rank, still issue the collective (RCCL all-gather of one
        shard = a device copy through the communicator) instead of returning the records as they are - the way the
        one-GPU box executes the RCCL code path of the N-rank run (tests/test_multirank_gpu.py, bench.py
        ST_BENCH_WORLD1_PG=1)."""
        self.device = torch.device(device) if device is not None else None
        self.stream = None
        self.seq = 0
        _, self.world = world()
        self.single_rank_collective = bool(single_rank_collective) and dist.is_available() and dist.is_initialized()
        self.on_device = ((self.world > 1 or self.single_rank_collective) and dist.is_initialized()
                          and dist.get_backend() == 'nccl')

    def gather(self, records, out=None):
        """records: this rank's (F, M + 1, C) frame records, produced on the CURRENT stream.
        -> (gathered (world * F, M + 1, C), event or None).  world 1: (records, None)."""
        self.seq += 1
        if self.world == 1 and not self.single_rank_collective:
            return records, None
        if not self.on_device:                 # gloo rehearsal: host memory, synchronous
            host = records.cpu()
            if out is None:
                out = torch.empty((self.world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
            dist.all_gather_into_tensor(out, host.contiguous())
            return out, None
        dev = records.device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        if out is None:
            out = torch.empty((self.world * records.shape[0],) + tuple(records.shape[1:]), dtype=records.dtype,
                              device=dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        self.stream.wait_event(ready)
        records.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            dist.all_gather_into_tensor(out, records.contiguous())
            done = torch.cuda.Event()
            done.record(self.stream)
        return out, done


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
