#!/usr/bin/env python
"""bench.py — stereo frame-pairs/s of the dense hot path on MI355X.

One step = one pass of the whole hot path over one batch of 8 synthetic 1280x720 stereo pairs
(BASELINE.json configs[1]: D=192, full YOLOX-s two-branch detector):
  stem+stage1 features of left AND right -> cost volume (D/4 = 48 levels at 1/4 res) + soft-argmin
  -> bilinear x4 -> disp_postp -> disparity branch + fused trunk + PAFPN + head -> decode +
  score filter + sort + NMS -> per-box depth + depth-guided scaling -> detection buffer
  (-> RCCL all-gather of the detection buffers when --gpus > 1).
Inputs are resident in HBM before the timed region.  Weights are seeded random (no checkpoints
offline).  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # one hardware queue per in-flight context (stereotracking_amd/__init__.py)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')   # kernel arguments in device memory (same file)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # RCCL across processes needs dmabuf IPC on this driver stack

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 dense peak
# committed profile records this line reads (tools/profile_round.sh / the GPU tests write them; a missing file = null)
PARITY_ROUND = 'r06' if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                                                      'r06_config2_oracle_blurred_shipped.json')) else 'r05'
TRAFFIC_ROUND = 'r06' if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                                                       'r06_hbm_traffic.json')) else 'r05'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=8, help='stereo pairs per GPU per step')
    ap.add_argument('--max-disp', type=int, default=192)
    ap.add_argument('--agg-layers', type=int, default=2, help='3x3 aggregation convs over the cost volume')
    ap.add_argument('--max-det', type=int, default=1000,
                    help='rows of the fixed-size detection buffer per frame (a capacity: overflow is an error)')
    ap.add_argument('--inflight', type=int, default=4,
                    help='pipeline contexts fed round-robin, one HIP stream each (1 = strictly serial steps)')
    ap.add_argument('--shell-inflight', type=int, default=3,
                    help='contexts of the MOT shell in the test_step leg (a synchronous call fills and drains them: '
                         '3 is faster than 4 there)')
    ap.add_argument('--split-bf16', action='store_true',
                    help='let the autotuner pick the split-operand (bf16x3) conv instances (default: exact-fp32 MFMA only)')
    ap.add_argument('--split-leg', action='store_true',
                    help='also time the workload with the split-operand instances allowed (secondary line)')
    ap.add_argument('--input-batches', type=int, default=4,
                    help='DISTINCT synthetic input batches resident in HBM, fed round-robin in the timed loop (181 MB each: '
                         'one batch alone could partly live in the 256 MB Infinity Cache across steps)')
    ap.add_argument('--agg3d-leg', action='store_true',
                    help='(on by default since round 6; kept for old command lines) the workload with ONE 3-D aggregation '
                         'layer (3x3x3 over d, y, x) in front of the 2-D ones: secondary line, pair 0 checked against the CPU oracle')
    ap.add_argument('--fullres-leg', action='store_true',
                    help='(on by default since round 6; kept for old command lines) the workload with the stereo module in its '
                         'FULL-RESOLUTION mode (a D=192 x 736 x 1280 volume per pair = north_star\'s literal sizing, one 3-D '
                         'aggregation layer, soft-argmin at image resolution): secondary line, pair 0 checked against the CPU oracle')
    ap.add_argument('--no-secondary-legs', action='store_true',
                    help='skip the secondary_agg3d / secondary_full_resolution legs (they never touch `value`)')
    ap.add_argument('--fullres-contexts', type=int, default=2, help='in-flight contexts of the --fullres-leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-test-step', action='store_true',
                    help='skip the second leg through Config.fromfile -> MODELS.build -> model.test_step')
    ap.add_argument('--frames-per-call', type=int, default=64,
                    help='frames of ONE video handed to each model.test_step call of the second leg (BASELINE configs[2] '
                         'is a 64-frame sequence); they run --batch at a time on the model\'s in-flight contexts')
    ap.add_argument('--frames-per-call-long', type=int, default=256,
                    help='a second, longer test_step call size: the per-call fill / drain of the in-flight contexts '
                         '(~5 ms) is amortised over more frames (0 = skip)')
    ap.add_argument('--sustain-seconds', type=float, default=2.0,
                    help='after the K timed steps, keep stepping for this long and report it as `sustained`')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU oracle leg')
    return ap.parse_args()


def conv_roofline(pipe, img, right, steps):
    """Per-op HIP-event timing of the conv kernels over `steps` passes (events are recorded on the
    launch stream inside the library, st_detector_set_timing)."""
    import ctypes as C
    import numpy as np
    from stereotracking_amd._lib import check
    det = pipe.det
    lib = det.lib
    check(lib.st_detector_set_timing(det.handle, 1))
    nops = lib.st_detector_num_ops(det.handle)
    ms = np.zeros(nops, np.float32)
    kind = np.zeros(nops, np.int32)
    var = np.zeros(nops, np.int32)
    macs = np.zeros(nops, np.float64)
    phase = np.zeros(nops, np.int32)
    tot_ms = np.zeros(nops, np.float64)
    other = {'costvolume+softargmin+upsample': 0.0, 'decode_nms': 0.0, 'box_depth': 0.0}
    sm = pipe.stereo_module
    sm.timing = True
    agg = {}   # variant -> [launches, ms]
    cv_ms = 0.0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    # an event pair with nothing in between already measures ~4.6 us on this stack (two barrier packets).  It is
    # measured and REPORTED (event_pair_overhead_us) but NOT subtracted: the per-launch durations below are raw
    # HIP-event times, ~5 % longer than rocprofv3's kernel durations of the same launches (profiles/r02_kernel_stats_
    # inflight1.csv), so `frac` is a lower bound that the committed rocprof summary can only improve on
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(101)]
    for a_, b_ in pairs:
        a_.record()
        b_.record()
    torch.cuda.synchronize()
    null_ms = sorted(a_.elapsed_time(b_) for a_, b_ in pairs)[50]
    for _ in range(steps):
        b = pipe._buffers(img.device)
        ev[0].record()
        disp = pipe.disparity(img, right)          # phase-0 convs + cost volume + upsample
        ev[1].record()
        pipe.det.forward_phase(1, disp=disp, head_out=b['head'])
        ev[2].record()
        boxes, scores, labels, prior, counts = pipe.det.decode_nms(b['head'], pipe.score_thr, pipe.iou_thr,
                                                                   pipe.max_det, (pipe.ori_h, pipe.ori_w))
        ev[3].record()
        pipe.box_depth(disp, boxes, counts)
        ev[4].record()
        torch.cuda.synchronize()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib.st_detector_op_times(det.handle, nops, p(ms), p(kind), p(var), p(macs), p(phase)))
        ph0 = float(ms[phase == 0].sum())     # raw: what the bracketing events of the stereo module saw
        tot_ms += ms
        cv_ms += sm.pop_costvolume_time()
        for v, t in sm.pop_times():   # aggregation convs: the same conv kernel, launched by the stereo module
            a = agg.setdefault(int(v), [0, 0.0])
            a[0] += 1
            a[1] += t
            ph0 += t
        other['costvolume+softargmin+upsample'] += ev[0].elapsed_time(ev[1]) - ph0
        other['decode_nms'] += ev[2].elapsed_time(ev[3])
        other['box_depth'] += ev[3].elapsed_time(ev[4])
    check(lib.st_detector_set_timing(det.handle, 0))
    sm.timing = False
    Hf, Wf = pipe.height // pipe.feat_stride, pipe.width // pipe.feat_stride
    agg_macs = sm.agg_macs(pipe.batch, Hf, Wf)   # per aggregation layer
    VARIANT_TILES = {v: lib.st_conv_variant_name(v).decode() for v in range(64)}
    VARIANT_TILES[-1] = 'skipped'
    # One row per LAUNCH kind: the ops a fused / grouped launch computes are summed into the row of the launch that
    # computes them, time and flops together ('wino2x2g+' = the riders of a grouped Winograd launch: their time is
    # inside the leading 'wino2x2g' op's event pair, so both names form ONE row; the fused front kernel's three ops
    # carry one variant name already).  Rounds 3-4 printed the riders as a row of their own with flops but no time.
    ROW = {'wino2x2g+': 'wino2x2g'}
    per_variant = {}
    for v in sorted(set(var[kind == 1].tolist()) | set(agg)):
        sel = (kind == 1) & (var == v)
        n_agg, ms_agg = agg.get(v, [0, 0.0])
        t = (tot_ms[sel].sum() + ms_agg) / steps
        fl = 2.0 * (macs[sel].sum() + agg_macs * n_agg / steps)
        n_launch = int(sel.sum()) + n_agg // steps
        n_events = n_launch      # every op of the plan is bracketed by its own event pair, fused-away ops included
        if VARIANT_TILES[v] == 'front3x3s2':
            n_launch //= 3      # one launch computes three ops of the plan (3x3/s2 + main|short + conv1)
        if VARIANT_TILES[v] == 'wino2x2g+':
            n_launch = 0        # riders of a grouped Winograd launch: computed by the 'wino2x2g' op in front of them
        if VARIANT_TILES[v] == 'wino32tail':
            n_launch //= 2      # one launch computes two ops of the plan (bottleneck conv2 + CSP final_conv)
        name = ROW.get(VARIANT_TILES[v], VARIANT_TILES[v])
        e = per_variant.setdefault(name, dict(launches=0, ops=0, event_pairs=0, ms_per_step=0.0, gflop_per_step=0.0))
        e['launches'] += n_launch
        e['ops'] += int(sel.sum()) + n_agg // steps
        e['event_pairs'] += n_events
        e['ms_per_step'] += float(t)
        e['gflop_per_step'] += fl / 1e9
    for e in per_variant.values():
        e['tflops'] = round(e['gflop_per_step'] / e['ms_per_step'], 3) if e['ms_per_step'] > 0 else 0.0
        e['ms_per_step'] = round(e['ms_per_step'], 4)
        e['gflop_per_step'] = round(e['gflop_per_step'], 3)
    # Kernel families of the MFMA work.  `roofline` describes the DOMINANT one = the family with the largest summed
    # duration per step.  conv_igemm_kernel's tile instances are one kernel template (which instance a layer uses is
    # an autotune outcome that varies from run to run) and are aggregated; the Winograd, direct 3x3, streaming 1x1
    # fused-stem, fused-front and LDS-resident 1x1 kernels are families of their own (a chained 1x1 pair counts as two
    # `launches` here: ops of the plan, the second with only the event overhead as duration).  `tflops` / `frac` = flops the
    # matrix pipes EXECUTE / summed duration (<= peak by construction); `algorithmic_tflops` = 2 x MACs of the direct
    # convolution / the same duration: the Winograd kernel executes 2.25x fewer multiplies than that,
    # so its algorithmic rate may exceed the MFMA peak (`algorithmic_speedup`).
    FAMILY = {'stem6x6s2': 'st::stem_focus_conv_kernel', 'pw128': 'st::pw_conv_kernel', 'dc4x32': 'st::direct_conv3x3_kernel',
              'wino2x2': 'st::wino_conv3x3_kernel', 'wino2x2n': 'st::wino_conv3x3_kernel', 'skipped': None,
              'wino2x2g': 'st::wino_conv3x3_kernel', 'wino2x2g+': 'st::wino_conv3x3_kernel',   # grouped launches
              'front3x3s2': 'st::front_s2_csp_kernel', 'pwres': 'st::pw_resident_kernel',
              'wino32tail': 'st::wino_csp_tail_kernel',   # round 6: stage-1 bottleneck conv2 (Winograd) + final_conv in one launch
              'headpred': 'st::head_pred_kernel'}   # (a VALU reduction, listed with the conv ops it replaces)
    fam = {}
    for name, v in per_variant.items():
        f = FAMILY.get(name, 'st::conv_igemm_kernel')
        if f is None:
            continue
        e = fam.setdefault(f, dict(launches=0, event_pairs=0, ms_per_step=0.0, gflop_per_step=0.0, instances=[]))
        e['launches'] += v['launches']
        e['event_pairs'] += v['event_pairs']
        e['ms_per_step'] += v['ms_per_step']
        e['gflop_per_step'] += v['gflop_per_step']
        e['instances'].append(name)
    # Durations: raw HIP-event times carry the event-pair overhead (two barrier packets, measured above on this stream:
    # `event_pair_overhead_us`); rocprofv3's kernel durations do not.  The overhead is subtracted per launch so that
    # `achieved` / `avg_launch_us` reproduce from profiles/r06_kernel_stats_inflight1.csv (the raw figures are kept
    # next to them).  `frac` is a fraction of the INSTRUCTION peak: flops the matrix pipes execute / time / 157.3 - for
    # the Winograd family that is the direct-convolution count / 2.25 (F(2x2,3x3) issues 16 of every 36 multiplies),
    # which goes into `algorithmic_speedup`, never into `frac`.
    WINO = 'st::wino_conv3x3_kernel'
    # the fused CSP tail executes its 3x3 conv (9 x 32 x 32 MACs per pixel) in Winograd form (/ 2.25 = 4096 MACs) and its 1x1
    # conv (64 x 64 = 4096 MACs) directly: 13312 direct MACs per pixel, 8192 executed
    SPEED = {WINO: 2.25, 'st::wino_csp_tail_kernel': 13312.0 / 8192.0}
    for f, e in fam.items():
        speed = SPEED.get(f, 1.0)
        raw = e['ms_per_step']
        e['ms_per_step_raw_events'] = round(raw, 4)
        e['ms_per_step'] = max(raw - e['event_pairs'] * null_ms, 1e-6)
        e['algorithmic_tflops'] = round(e['gflop_per_step'] / e['ms_per_step'], 3)
        e['algorithmic_speedup'] = speed
        e['tflops'] = round(e['gflop_per_step'] / speed / e['ms_per_step'], 3)       # executed by the matrix pipes
        e['frac'] = round(e['tflops'] / PEAK_FP32_MFMA_TFLOPS, 4)
        e['ms_per_step'] = round(e['ms_per_step'], 4)
        e['gflop_per_step'] = round(e['gflop_per_step'], 3)
        e['executed_gflop_per_step'] = round(e['gflop_per_step'] / speed, 3)
    dom = max(fam, key=lambda k: fam[k]['ms_per_step'])
    D = fam[dom]
    n_conv_launches = sum(e['event_pairs'] for e in fam.values())
    conv_ms = float((tot_ms[kind == 1].sum() + sum(a[1] for a in agg.values())) / steps) - n_conv_launches * null_ms
    conv_fl = 2.0 * float(macs[kind == 1].sum() + agg_macs * sum(a[0] for a in agg.values()) / steps)
    conv_exec = sum(e['executed_gflop_per_step'] for e in fam.values()) * 1e9
    # HBM bytes per launch of the dominant family from this round's rocprofv3 PMC passes of this command (FETCH_SIZE x2
    # gfx950 correction + WRITE_SIZE, separate passes; tools/profile_round.sh writes profiles/r06_hbm_traffic.json from
    # the SAME commit's library).  null when that file is absent: never a number from another round.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, 'profiles', TRAFFIC_ROUND + '_hbm_traffic.json')
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        tot_b, cov = 0.0, 0
        for kname, row in tj.items():
            if (kname.startswith(dom) or (dom == WINO and 'wino_conv3x3_group_kernel' in kname)) and row.get('fetch_bytes_corrected_per_launch') is not None:
                n = row.get('fetch_calls') or 0
                tot_b += n * (row['fetch_bytes_corrected_per_launch'] + (row.get('write_bytes_per_launch') or 0))
                cov += n
        if cov:
            traffic = int(tot_b / cov)
            traffic_src = ('profiles/' + TRAFFIC_ROUND + '_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command '
                           '(separate runs of tools/profile_round.sh), launch-weighted over %s*' % dom)
    roof = dict(bound='mfma', kernel=dom + ('<...> (all tile instances)' if dom == 'st::conv_igemm_kernel' else ''),
                achieved=D['tflops'], peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s', frac=D['frac'],
                traffic=traffic, traffic_source=traffic_src,
                algorithmic_tflops=D['algorithmic_tflops'], algorithmic_speedup=D['algorithmic_speedup'],
                flop_per_launch=round(D['executed_gflop_per_step'] * 1e9 / max(D['launches'], 1)),
                algorithmic_flop_per_launch=round(D['gflop_per_step'] * 1e9 / max(D['launches'], 1)),
                avg_launch_us=round(D['ms_per_step'] * 1e3 / max(D['launches'], 1), 2),
                avg_launch_us_raw_events=round(D['ms_per_step_raw_events'] * 1e3 / max(D['launches'], 1), 2),
                launches_per_step=D['launches'], event_pair_overhead_us=round(null_ms * 1e3, 2),
                event_overhead_subtracted=True,
                definition='achieved = flops EXECUTED by the matrix pipes (direct-convolution count / algorithmic_speedup) '
                           '/ summed kernel duration of the family in a serialized pass (HIP events on the launch '
                           'stream minus the measured event-pair overhead; an event-bracketed launch also carries its '
                           'dispatch latency, so these durations read ~4 % above the rocprofv3 kernel-trace durations '
                           'of the same launches in profiles/r06_kernel_stats_inflight1.csv: `achieved` is a lower '
                           'bound); frac = achieved / peak <= 1',
                families=fam,
                all_mfma_kernels=dict(ms_per_step=round(conv_ms, 4),
                                      algorithmic_tflops=round(conv_fl / (conv_ms * 1e-3) / 1e12, 3),
                                      tflops=round(conv_exec / (conv_ms * 1e-3) / 1e12, 3),
                                      frac=round(conv_exec / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)),
                per_variant=per_variant,
                other_kernels_ms_per_step={k: round(v / steps, 4) for k, v in other.items()},
                focus_spp_ms_per_step=round(float(tot_ms[kind != 1].sum() / steps), 4))
    # secondary roofline (SURVEY.md §8d "cost volume: HBM-bound scan", materialised form at 1/4 resolution):
    # algorithmic bytes of one costvolume launch = both feature maps read once + the volume written once
    Cf = pipe.det.tap('stage1_rgb').shape[-1]
    cv_bytes = pipe.batch * Hf * Wf * 4.0 * (2 * Cf + (sm.levels if sm.agg_layers else 0)) + pipe.batch * Hf * Wf * 4.0
    cv_us = cv_ms / steps * 1e3
    roof['secondary_costvolume'] = dict(bound='hbm', kernel='st::costvolume_tiled_kernel', unit='GB/s', peak=8000.0,
                                        bytes_per_launch=int(cv_bytes), avg_launch_us=round(cv_us, 2),
                                        achieved=round(cv_bytes / (cv_us * 1e-6) / 1e9, 1) if cv_us > 0 else None,
                                        frac=round(cv_bytes / (cv_us * 1e-6) / 8e12, 4) if cv_us > 0 else None)
    roof['secondary_costvolume_fullres'] = costvolume_fullres(lib)
    return roof


def costvolume_fullres(lib, reps=5):
    """SURVEY.md §8(d)'s OTHER cost-volume sizing, as a kernel measurement: the full-resolution materialised volume
    D=192 x 720 x 1280 (176.9 M cells, 708 MB fp32) from C=8 synthetic feature maps, one pair per launch.  The product
    pipeline correlates stage-1 features at 1/4 resolution (48 levels, 2.83 M cells); this line only shows what the
    same kernel family does at the full-resolution sizing.  Algorithmic bytes = both feature maps read once + the
    volume written once (the soft-argmin then reads it once more: st_softargmin, not timed here)."""
    import ctypes as C
    try:
        Hf, Wf, Cf, D = 720, 1280, 8, 192
        dev = torch.device('cuda', torch.cuda.current_device())
        g = torch.Generator(device='cpu').manual_seed(0)
        fl = torch.randn(1, Hf, Wf, Cf, generator=g).to(dev)
        fr = torch.randn(1, Hf, Wf, Cf, generator=g).to(dev)
        vol = torch.empty(1, Hf, Wf, D, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = []
        for _ in range(reps + 1):
            e0.record()
            rc = lib.st_costvolume_softargmin(C.c_void_p(fl.data_ptr()), C.c_void_p(fr.data_ptr()), 1, Hf, Wf, Cf, Cf, D,
                                              1.0, C.c_void_p(vol.data_ptr()), None, stream)
            e1.record()
            e1.synchronize()
            if rc != 0:
                return dict(error=lib.st_last_error().decode())
            ms.append(e0.elapsed_time(e1))
        us = sorted(ms[1:])[len(ms[1:]) // 2] * 1e3
        nbytes = 4.0 * Hf * Wf * (2 * Cf + D)
        return dict(bound='hbm', kernel='st::costvolume_tiled_kernel<24> x 2 slabs of 96 disparities', unit='GB/s',
                    peak=8000.0, cells=Hf * Wf * D,
                    bytes_per_launch=int(nbytes), launch_us=round(us, 1), achieved=round(nbytes / (us * 1e-6) / 1e9, 1),
                    frac=round(nbytes / (us * 1e-6) / 8e12, 4),
                    workload='1 pair, full-resolution volume D=192 x 720 x 1280 from C=8 features (kernel only)')
    except Exception as e:   # a secondary line must never cost the headline
        return dict(error=repr(e))


def cpu_baseline(sd, batch_cpu, max_disp, seconds, agg_layers, max_det=1000):
    """The CPU oracle (kind 'port': this repo's restatement of the reference path; the reference itself
    cannot be imported) timed on the host cores over whole stereo pairs."""
    import numpy as np
    from oracle import c_oracle, depth as odepth, stereo as ostereo
    from oracle.torch_model import OracleDetector, head_to_rows
    # north_star: "the reference CPU path timed on the same box's host cores (core count stated)".  Three thread counts
    # are timed (torch intra-op threads and the C oracle's OpenMP loops bound alike): 16 (a 1-GPU box's nominal CPU
    # share, the figure of rounds 2-5), 64, and every physical core this process may run on; `value` = the best one,
    # `cores` = its count, `by_threads` lists all three.  Method: mmtrack/utils/benchmark.py:195-228 (warm-up, then a
    # fixed sample under a wall clock).
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    physical = max(1, allowed // 2) if allowed >= 32 else allowed    # SMT siblings add nothing to fp32 FMA loops
    thread_counts = sorted({min(16, allowed), min(64, physical), physical})
    ora = OracleDetector(0.33, 0.5, 1).eval()
    ora.load_state_dict(sd, strict=False)
    img, right, = batch_cpu['img'][:1], batch_cpu['right'][:1]
    H, W = img.shape[-2:]
    ori_h, ori_w = 720, 1280
    D = max_disp // 4

    keep = {}

    def one_pair():
        with torch.no_grad():
            fl = ora.backbone.stage1_features(img).permute(0, 2, 3, 1).contiguous().numpy()
            fr = ora.backbone.stage1_features(right).permute(0, 2, 3, 1).contiguous().numpy()
            disp = torch.from_numpy(ostereo.disparity(fl, fr, fl.shape[-1], D, 32.0, sd, agg_layers,
                                                      valid_hw=(ori_h, ori_w))[2])
            rows = head_to_rows(*ora(dict(img=img, disp_postp=disp)))
        levels, off, flat = [], 0, []
        for r, s in zip(rows, (8, 16, 32)):
            hw = r.shape[1]
            h = H // s
            pad = torch.zeros(1, hw, 8)
            pad[..., :6] = r
            levels.append((h, hw // h, s, off))
            off += hw * 8
            flat.append(pad.reshape(-1))
        head = torch.cat(flat).numpy()
        boxes, scores, labels, prior, counts = c_oracle.decode_nms(head, 1, levels, 0.01, 0.5, max_det, (ori_h, ori_w))
        k = min(int(counts[0]), max_det)      # the same capacity as the GPU's detection buffer: no box dropped
        odepth.bbox_postp_depth(torch.from_numpy(boxes[0, :k]), disp)
        keep['disp'] = disp
        return k

    by_threads = []
    per = max(2.0, seconds / len(thread_counts))
    best = None
    for nt in thread_counts:
        torch.set_num_threads(nt)
        c_oracle.set_threads(nt)
        one_pair()  # warm-up (oneDNN primitive caches, thread pools of this size)
        t0 = time.perf_counter()
        n = 0
        while True:
            one_pair()
            n += 1
            if time.perf_counter() - t0 >= per or n >= 512:
                break
        dt = time.perf_counter() - t0
        by_threads.append(dict(threads=nt, value=round(n / dt, 4), pairs=n, seconds=round(dt, 2)))
        if best is None or n / dt > best[0]:
            best = (n / dt, nt, n, dt)
    torch.set_num_threads(1)          # SURVEY.md §8d also asks for the 1-thread figure (one pair)
    c_oracle.set_threads(1)
    t1 = time.perf_counter()
    one_pair()
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(min(16, allowed))
    c_oracle.set_threads(min(16, allowed))
    v, nt, n, dt = best
    return dict(oracle_disp_pair0=keep['disp'], value=round(v, 4), unit='stereo frame-pairs/s', cores=nt, kind='port',
                by_threads=by_threads, value_1_thread=round(1.0 / dt1, 4),
                sample=f'{n} x 1 synthetic 1280x720 pair (D={max_disp}, {agg_layers} aggregation convs, full YOLOX-s '
                       'two-branch), CPU oracle '
                       f'(PyTorch fp32 + C oracle) on {nt} threads, {dt:.1f} s; every thread count of `by_threads` timed the same way',
                host_cpus=os.cpu_count(), cpus_allowed=allowed, physical_cores_assumed=physical)


def split_leg(args, sd, img, right, headline):
    """SECONDARY line, never `value`: the same workload with the split-operand conv instances allowed in the plan (fp32
    operands as three bf16 terms, six exact products on v_mfma_f32_32x32x16_bf16, fp32 accumulate; off by default).
    A plan of its own is autotuned (and remembered in the user cache), the K steps are timed exactly like the headline."""
    from stereotracking_amd.pipeline import InflightPipelines
    try:
        runner = InflightPipelines(max(1, args.inflight), args.batch, (720, 1280), 0.5, 0.33, 1, stereo=True,
                                   max_disp=args.max_disp, max_det=args.max_det, agg_layers=args.agg_layers,
                                   split_bf16=True)
        runner.load_state_dict(sd)
        for _ in range(args.warmup):
            runner.submit(img, right)
        runner.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.submit(img, right)
        runner.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        plan = runner.pipes[0].det.get_tuning()
        v = args.batch * args.steps / dt
        return dict(value=round(v, 3), unit='stereo frame-pairs/s', ms_per_step=round(dt / args.steps * 1e3, 4),
                    vs_headline=round(v / headline, 4), ops_on_split_instances=sum(1 for t in plan if 50 <= t <= 55),
                    conv_ops=sum(1 for t in plan if t >= 0),
                    note='NOT the headline: plan with split-operand (bf16x3) conv instances; outputs differ from the '
                         'exact-fp32 plan by fp32-level noise (head 3e-4 relative), parity record in DESIGN.md')
    except Exception as e:   # a secondary line must never cost the headline
        return dict(error=repr(e))


def parity_records():
    """The parity truth, read from the COMMITTED records of this round's GPU tests (tests/test_config2_oracle_gpu.py writes
    them; profiles/r06_config2_oracle_<sequence>_<thresholds>.json): configs[2] end to end through model.test_step against
    the CPU fp32 oracle pipeline AND a float64 evaluation of the same arithmetic.  north_star asks for floats within 1e-3
    of the CPU path and bit-exact indices: per sequence the line says whether that literal bar holds
    (`within_1e3_of_cpu_path`) and what was measured where it does not.  null when the records are absent."""
    out = {}
    for seq in ('blurred', 'white_noise'):
        for thr in ('shipped', 'stress'):
            path = os.path.join(ROOT, 'profiles', f'{PARITY_ROUND}_config2_oracle_{seq}_{thr}.json')
            if not os.path.exists(path):
                continue
            r = json.load(open(path))
            t, w, e, dist = r['totals'], r['worst'], r['vs_fp64'], r['box_vs_fp64_distribution']
            frames = len(r.get('frames', [])) or None
            out[f'{seq}/{thr}'] = dict(
                boxes_gpu_vs_cpu32=w['box'], boxes_gpu_vs_fp64=e['gpu_box'], boxes_cpu32_vs_fp64=e['cpu_box'],
                scores_gpu_vs_cpu32=w['score'], within_1e3_of_cpu_path=bool(w['box'] <= 1e-3 and w['score'] <= 1e-3),
                boxes_compared=dist['gpu']['boxes'], boxes_over_1e3_vs_fp64=dict(gpu=dist['gpu']['over_1e3'], cpu32=dist['cpu32']['over_1e3']),
                kept_set_differences=t['det_sym_diff'], frames=frames,
                frames_with_equal_detection_order=t.get('frames_with_equal_det_order'),
                frames_with_ids_equal_in_order=t['frames_with_equal_ids_in_order'],
                track_rows=t['track_rows'], track_rows_outside_the_id_bijection=t['inconsistent'] + t['only_gpu'] + t['only_oracle'],
                ids_seen=r['ids_seen'], ids_relabeled=r['ids_relabeled'])
    if not out:
        return None
    out['source'] = ('COMMITTED RECORD, not measured by this run: profiles/%s_config2_oracle_*.json, written by '
                     'tests/test_config2_oracle_gpu.py on MI355X with the committed plan (the driver\'s `pytest -m gpu` asserts '
                     'the same bounds live)' % PARITY_ROUND)
    out['reading'] = ('"bit-exact indices" holds as an equivalence class: equal kept sets up to detections inside the measured '
                      'fp32 noise of a threshold, ONE id bijection over all track rows; the detection ORDER (float score '
                      'order) is not reproduced by any fp32 evaluation, the CPU oracle included (vs float64)')
    return out


def agg3d_leg(args, inputs, batch_cpu, headline, dev):
    """SECONDARY line, never `value`: the same workload with ONE 3-D aggregation layer (single-channel 3x3x3 over d, y, x:
    csrc/agg3d.hip) in front of the 2-D aggregation convs - north_star's "3D/2D aggregation" as a benched form.  The K
    steps are timed exactly like the headline; pair 0's disparity is checked against the CPU oracle with the same
    layer (oracle/stereo.py, agg3d_layers=1); the layer's own launch is timed with HIP events on its stream."""
    import ctypes as C
    from oracle import stereo as ostereo
    from oracle.torch_model import OracleDetector
    from stereotracking_amd._lib import check, ptr
    from stereotracking_amd.pipeline import InflightPipelines
    from stereotracking_amd.synthetic import synthetic_state_dict
    try:
        runner = InflightPipelines(max(1, args.inflight), args.batch, (720, 1280), 0.5, 0.33, 1, stereo=True,
                                   max_disp=args.max_disp, max_det=args.max_det, agg_layers=args.agg_layers,
                                   agg3d_layers=1)
        sd = synthetic_state_dict(runner.param_table(), seed=0)     # keyed by NAME: the detector / 2-D weights of the headline
        g = torch.Generator().manual_seed(3)
        w3 = torch.randn(1, 1, 3, 3, 3, generator=g) * 0.15
        w3[0, 0, 1, 1, 1] += 1.0                                     # a smoothing-like kernel around the identity
        sd['stereo.agg3d.0.weight'], sd['stereo.agg3d.0.bias'] = w3, torch.zeros(1)
        runner.load_state_dict(sd)
        nb = len(inputs)
        for i in range(args.warmup):
            runner.submit(*inputs[i % nb])
        runner.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = runner.submit(*inputs[i % nb])[0]
        runner.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        v = args.batch * args.steps / dt
        out = runner.submit(*inputs[0])[0]
        torch.cuda.synchronize()
        disp0 = out['disp_postp'][0, 0].cpu()
        # the layer alone, at the bench volume
        pipe = runner.pipes[0]
        lib = pipe.det.lib
        Hf, Wf, D = pipe.height // pipe.feat_stride, pipe.width // pipe.feat_stride, pipe.D
        vin = torch.randn(args.batch, Hf, Wf, D, device=dev)
        vout = torch.empty_like(vin)
        w27 = (C.c_float * 27)(*w3.reshape(-1).tolist())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            check(lib.st_volume_agg3d(ptr(vin), ptr(vout), args.batch, Hf, Wf, D, w27, 0.0, 0, None))
        e0.record()
        for _ in range(20):
            check(lib.st_volume_agg3d(ptr(vin), ptr(vout), args.batch, Hf, Wf, D, w27, 0.0, 0, None))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nbytes = 2.0 * vin.numel() * 4
        # oracle, pair 0
        ora = OracleDetector(0.33, 0.5, 1).eval()
        ora.load_state_dict(sd, strict=False)
        img, right = batch_cpu['img'][:1], batch_cpu['right'][:1]
        with torch.no_grad():
            fl = ora.backbone.stage1_features(img).permute(0, 2, 3, 1).contiguous().numpy()
            fr = ora.backbone.stage1_features(right).permute(0, 2, 3, 1).contiguous().numpy()
        ref = torch.from_numpy(ostereo.disparity(fl, fr, fl.shape[-1], D, 32.0, sd, args.agg_layers, valid_hw=(720, 1280),
                                                 agg3d_layers=1)[2])[0, 0]
        ad = (disp0 - ref).abs()
        return dict(value=round(v, 3), unit='stereo frame-pairs/s', ms_per_step=round(dt / args.steps * 1e3, 4),
                    vs_headline=round(v / headline, 4), agg3d_layers=1, agg_layers=args.agg_layers,
                    roofline=dict(kernel='st::vol_agg3d_kernel', volume=[args.batch, Hf, Wf, D], avg_launch_us=round(us, 2),
                                  bytes_per_launch=int(nbytes), bound='hbm', peak=8000.0, unit='GB/s',
                                  achieved=round(nbytes / (us * 1e-6) / 1e9, 1), frac=round(nbytes / (us * 1e-6) / 8e12, 4),
                                  definition='the 3-D layer alone at the bench volume: volume read once + written once / '
                                             'average of 20 back-to-back launches between two HIP events on the launch stream'),
                    disparity_vs_oracle_pair0=dict(l1_px=float(ad.mean()), max_abs_px=float(ad.max()),
                                                   max_rel=float((ad / ref.abs().clamp(min=1.0)).max())),
                    note='NOT the headline: one 3x3x3 aggregation layer over (d, y, x) added in front of the 2-D convs; the '
                         'frozen spec of the benched module (SURVEY 8 a-7) aggregates with 2-D convs only')
    except Exception as e:   # a secondary line must never cost the headline
        return dict(error=repr(e))


def fullres_leg(args, inputs, batch_cpu, headline, plan, dev):
    """SECONDARY line, never `value`: north_star's literal sizing as a product path - StereoCostVolume(full_res=True):
    stage-1 features reduced to 8 channels and brought to image resolution, a D = 192 level volume per pair at 736 x 1280
    (181 M cells, 694 MB; materialised in two slabs of 96), ONE 3x3x3 aggregation layer over (d, y, x), soft-argmin in
    pixels; then the same detector / decode / depth as the headline.  Two contexts in flight (11.6 GB of volumes each).
    Pair 0's disparity is checked against the CPU oracle (oracle/stereo.py::disparity_fullres)."""
    from oracle import stereo as ostereo
    from oracle.torch_model import OracleDetector
    from stereotracking_amd.pipeline import InflightPipelines
    from stereotracking_amd.synthetic import synthetic_state_dict
    try:
        B, D = args.batch, args.max_disp
        nctx = max(1, args.fullres_contexts)
        runner = InflightPipelines(nctx, B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=D, max_det=args.max_det,
                                   agg_layers=0, agg3d_layers=1, full_res=True)
        sd = synthetic_state_dict(runner.param_table(), seed=0)
        g = torch.Generator().manual_seed(3)
        w3 = torch.randn(1, 1, 3, 3, 3, generator=g) * 0.15
        w3[0, 0, 1, 1, 1] += 1.0
        sd['stereo.agg3d.0.weight'], sd['stereo.agg3d.0.bias'] = w3, torch.zeros(1)
        runner.load_state_dict(sd, autotune=False)
        for p in runner.pipes:                     # the detector graph is the headline's: reuse its committed plan
            p.det.set_tuning(plan)
        nb = len(inputs)
        steps, warm = max(4 * nctx, args.steps // 5), 2 * nctx
        for i in range(warm):
            runner.submit(*inputs[i % nb])
        runner.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            runner.submit(*inputs[i % nb])
        runner.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        v = B * steps / dt
        pipe = runner.pipes[0]
        pipe.stereo_module.timing = True
        out = pipe.run(*inputs[0])
        torch.cuda.synchronize()
        stages = pipe.stereo_module.pop_full_res_times()
        pipe.stereo_module.timing = False
        disp0 = out['disp_postp'][0, 0].cpu()
        H, W = pipe.height, pipe.width
        vol_bytes = 4.0 * B * H * W * D
        cv_b = vol_bytes + 2 * 4.0 * B * H * W * 8
        def hbm(ms, nbytes):
            return dict(ms=round(ms, 3), bytes=int(nbytes), frac_of_8TBs=round(nbytes / (ms * 1e-3) / 8e12, 4))
        kern = dict(features_reduce_upsample_ms=round(stages['features_reduce_upsample'], 3))
        if 'softargmin_pack' in stages:
            kern['softargmin_pack'] = hbm(stages['softargmin_pack'], vol_bytes)
        if 'cost_volume_agg3d_softargmin' in stages:
            # round 6: cost volume + the 3-D layer + soft-argmin in ONE kernel (st_costvolume_agg3d_softargmin): features in,
            # disparity out; the D-level volume is neither written nor read back.  `bytes` keeps the algorithmic traffic of the
            # materialised form it replaces (volume written once + read once) so the fraction compares with earlier rounds;
            # `bytes_moved` is what this kernel itself must move (features + disparity)
            fb = 2 * 4.0 * B * H * W * 8 + 4.0 * B * H * W
            kern['cost_volume_agg3d_softargmin_fused'] = dict(
                hbm(stages['cost_volume_agg3d_softargmin'], cv_b + vol_bytes), bytes_moved=int(fb),
                valu_tflops=round(2.0 * (27 + 8) * B * H * W * D / (stages['cost_volume_agg3d_softargmin'] * 1e-3) / 1e12, 1))
            kern['pack_ms'] = round(stages['pack'], 3)
        elif 'cost_volume_agg3d_first' in stages:     # one pass: features in, aggregated volume out (st_costvolume_agg3d)
            kern['cost_volume_agg3d_fused'] = dict(hbm(stages['cost_volume_agg3d_first'], cv_b),
                                                   valu_tflops=round(2.0 * (27 + 8) * B * H * W * D / (stages['cost_volume_agg3d_first'] * 1e-3) / 1e12, 1))
        else:
            kern['cost_volume'] = hbm(stages['cost_volume'], cv_b)
            kern['agg3d'] = hbm(stages['agg3d'], 2 * vol_bytes)
        # the leg's roofline object: its dominant kernel = the fused cost volume + first 3-D layer (features read once,
        # aggregated volume written once); the other kernels of the stage listed beside it with their own fractions
        dom_name = ('cost_volume_agg3d_softargmin_fused' if 'cost_volume_agg3d_softargmin_fused' in kern else
                    'cost_volume_agg3d_fused' if 'cost_volume_agg3d_fused' in kern else 'cost_volume')
        dom = kern[dom_name]
        leg_roof = dict(bound='hbm', kernel='st::cv_agg3d_kernel<..., SA>' if dom_name == 'cost_volume_agg3d_softargmin_fused' else
                        'st::cv_agg3d_kernel' if dom_name == 'cost_volume_agg3d_fused' else 'st::costvolume_tiled_kernel',
                        unit='GB/s', peak=8000.0, bytes_per_launch=dom['bytes'], avg_launch_us=round(dom['ms'] * 1e3, 1),
                        achieved=round(dom['bytes'] / (dom['ms'] * 1e-3) / 1e9, 1), frac=dom['frac_of_8TBs'],
                        valu_tflops=dom.get('valu_tflops'),
                        note='VALU-bound by its counters (35 fp32 FMAs per cell, vector ALU busy 0.86): the HBM fraction is the '
                             'contract\'s figure, not its bound; for the single-kernel form `bytes_per_launch` is the algorithmic '
                             'traffic of the materialised form it replaces (features + volume written + volume read), kept for '
                             'comparison across rounds - the kernel itself moves `bytes_moved`',
                        bytes_moved=dom.get('bytes_moved'),
                        stage_bytes_per_step=int(cv_b + vol_bytes),
                        stage_ms_per_step=round(sum(v['ms'] if isinstance(v, dict) else v for v in kern.values()), 3),
                        definition='one serialized pass of the stereo stage on one context, HIP events on the launch stream '
                                   'around each kernel (st::softargmin_reg_kernel + pack in `softargmin_pack`)')
        leg_roof['stage_frac_of_8TBs'] = round(leg_roof['stage_bytes_per_step'] / (leg_roof['stage_ms_per_step'] * 1e-3) / 8e12, 4)
        # oracle, pair 0 (the C oracle walks 181 M cells three times: tens of seconds)
        from oracle import c_oracle
        try:
            c_oracle.set_threads(min(64, max(1, len(os.sched_getaffinity(0)) // 2)))
        except AttributeError:
            pass
        ora = OracleDetector(0.33, 0.5, 1).eval()
        ora.load_state_dict(sd, strict=False)
        img, right = batch_cpu['img'][:1], batch_cpu['right'][:1]
        t1 = time.perf_counter()
        with torch.no_grad():
            fl = ora.backbone.stage1_features(img).permute(0, 2, 3, 1).contiguous().numpy()
            fr = ora.backbone.stage1_features(right).permute(0, 2, 3, 1).contiguous().numpy()
        ref = torch.from_numpy(ostereo.disparity_fullres(fl, fr, fl.shape[-1], D, 32.0, sd, 1, valid_hw=(720, 1280))[2])[0, 0]
        ad = (disp0 - ref).abs()
        return dict(value=round(v, 3), unit='stereo frame-pairs/s', ms_per_step=round(dt / steps * 1e3, 3), steps=steps,
                    vs_headline=round(v / headline, 4), inflight_contexts=nctx,
                    volume=dict(levels=D, height=H, width=W, cells_per_pair=D * H * W, bytes_per_pair=int(vol_bytes / B)),
                    roofline=leg_roof, stage_ms_per_step_serialized=kern,
                    disparity_vs_oracle_pair0=dict(l1_px=float(ad.mean()), max_abs_px=float(ad.max()),
                                                   max_rel=float((ad / ref.abs().clamp(min=1.0)).max()),
                                                   oracle_seconds=round(time.perf_counter() - t1, 1)),
                    note='NOT the headline: the stereo module in full-resolution mode (reduce 64 -> 8, bilinear x4, D=192 levels at '
                         '736x1280, one 3x3x3 aggregation layer, soft-argmin in pixels); the benched default correlates 64-channel '
                         'features at 1/4 resolution (48 levels = the same 192 px range)')
    except Exception as e:   # a secondary line must never cost the headline
        return dict(error=repr(e))


def test_step_leg(args, sd, batch_cpu, dev, pairs_target):
    """The SAME workload through the reference's plugin surface (SURVEY.md §8b "Callers"): Config.fromfile of the
    stereo config -> MODELS.build -> model.test_step(data), data = what a dataset pipeline yields (lists of (1,3,h,w)
    frames of ONE video + TrackDataSamples), inputs resident in HBM.  One call carries --frames-per-call frames (64 =
    the sequence length of BASELINE configs[2]); inside the call they run `batch` at a time on the model's in-flight
    contexts while the CPU association step (shipped thresholds) consumes the finished chunks in frame order."""
    from stereotracking_amd import mot  # noqa: F401  (registers the classes)
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    from stereotracking_amd.structures import TrackDataSample
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort',
                                       'stereo_yolox_s_mot_airdrone_costvolume.py'))
    cfg.model.stereo['max_disp'] = args.max_disp
    cfg.model.stereo['agg_layers'] = args.agg_layers
    B, F = args.batch, max(args.batch, args.frames_per_call)
    model = MODELS.build(dict(cfg.model, dense_batch=B, inflight=max(1, args.shell_inflight), max_det=args.max_det,
                              tuning_cache=os.environ.get('ST_TUNE_CACHE'),
                              split_bf16=True if args.split_bf16 else None))
    model.detector.load_state_dict({k: v for k, v in sd.items() if not k.startswith('stereo.')}, strict=False)
    model.stereo.load_state_dict({k[len('stereo.'):]: v for k, v in sd.items() if k.startswith('stereo.')})
    left = [batch_cpu['img'][i % B:i % B + 1, :, :720].to(torch.uint8).to(dev) for i in range(F)]
    right = [batch_cpu['right'][i % B:i % B + 1, :, :720].to(torch.uint8).to(dev) for i in range(F)]
    frame = [0]

    def data(nf=None):
        nf = nf or F
        samples = [TrackDataSample(dict(frame_id=frame[0] + i, ori_shape=(720, 1280), img_shape=(720, 1280),
                                        scale_factor=(1.0, 1.0))) for i in range(nf)]
        frame[0] += nf
        return dict(inputs=dict(img=[left[i % F] for i in range(nf)], right=[right[i % F] for i in range(nf)]),
                    data_samples=samples)

    def call(nf=None):
        return model.test_step(data(nf))

    for _ in range(2):
        outs = call()
    torch.cuda.synchronize()
    model.timings.update(frames=0, tracker_s=0.0, host_s=0.0, wait_s=0.0, pre_s=0.0, tail_s=0.0, submit_s=0.0, depth_s=0.0)
    calls = max(2, -(-pairs_target // F))
    t0 = time.perf_counter()
    for _ in range(calls):
        outs = call()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = dict(model.timings)
    long_call = None
    FL = args.frames_per_call_long
    if FL > F:
        call(FL)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            call(FL)
        torch.cuda.synchronize()
        long_call = dict(frames_per_call=FL, calls=2, value=round(2 * FL / (time.perf_counter() - t0), 3),
                         unit='stereo frame-pairs/s')
    # the same calls through model.test_steps(iterable): the loop form of test_step that keeps the contexts primed across
    # calls (call k+1's first chunks are submitted while call k drains); every call's results are what test_step returns
    t0 = time.perf_counter()
    n_loop = 0
    for outs_l in model.test_steps(data() for _ in range(calls)):
        n_loop += len(outs_l)
    torch.cuda.synchronize()
    loop = dict(value=round(n_loop / (time.perf_counter() - t0), 3), unit='stereo frame-pairs/s', calls=calls,
                frames_per_call=F, path='for outs in model.test_steps(dataloader)')
    return dict(value=round(calls * F / dt, 3), primed_loop=loop, long_call=long_call, unit='stereo frame-pairs/s', calls=calls, frames_per_call=F,
                ms_per_call=round(dt / calls * 1e3, 3),
                path='Config.fromfile(stereo_yolox_s_mot_airdrone_costvolume.py) -> MODELS.build -> model.test_step',
                tracker_ms_per_frame=round(tm['tracker_s'] / max(tm['frames'], 1) * 1e3, 4),
                host_ms_per_call=dict(preprocessor=round(tm['pre_s'] / calls * 1e3, 3), predict=round(tm['host_s'] / calls * 1e3, 3),
                                      of_which_waiting_for_gpu=round(tm['wait_s'] / calls * 1e3, 3),
                                      submitting_chunks=round(tm['submit_s'] / calls * 1e3, 3),
                                      association=round(tm['tracker_s'] / calls * 1e3, 3),
                                      track_depth_launches=round(tm['depth_s'] / calls * 1e3, 3),
                                      after_last_chunk_left_the_gpu=round(tm['tail_s'] / calls * 1e3, 3)),
                tracks_last_frame=int(len(outs[-1].pred_track_instances)),
                detections_last_frame=int(len(outs[-1].pred_det_instances)),
                note='includes the preprocessor (uint8 frames; the cast + pad happens inside the stem kernels), the dense path on the model\'s in-flight '
                     'contexts, one D2H of the detection records per 8-frame chunk, the CPU OC-SORT association with the '
                     'shipped thresholds and one batched depth launch for the tracks per chunk')


def tracker_cost(seconds=1.0):
    """Host cost of the association step alone on a realistic load (6 objects, dropped
    detections, an occlusion): ms per frame of OCSORTTracker_Disparity with the shipped thresholds.  Decides whether
    SURVEY §8 f-4 (batched GPU association) is needed: the dense path delivers a frame every ~0.75 ms."""
    from stereotracking_amd.motion import KalmanFilter
    from stereotracking_amd.structures import InstanceData, TrackDataSample
    from stereotracking_amd.synthetic import synthetic_detection_stream
    from stereotracking_amd.trackers import OCSORTTracker_Disparity

    class _Model:
        motion = KalmanFilter()

    T = 64
    det = synthetic_detection_stream(51, T)
    frames = []
    for t in range(T):
        d = det[det[:, 0] == t]
        frames.append(dict(bboxes=torch.from_numpy(d[:, 1:5].copy()), scores=torch.from_numpy(d[:, 5].copy()),
                           labels=torch.zeros(len(d), dtype=torch.long), scales=torch.from_numpy(d[:, 7].copy()),
                           depth=torch.from_numpy(d[:, 6].copy())))
    trk = OCSORTTracker_Disparity(obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=False,
                                  match_iou_thr=0.1, num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3,
                                  num_frames_retain=30)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for t in range(T):
            s = TrackDataSample(dict(frame_id=t))
            s.pred_det_instances = InstanceData(**frames[t])
            trk.track(_Model(), None, None, s)
        n += T
    return dict(ms_per_frame=round((time.perf_counter() - t0) / n * 1e3, 4), objects=6, frames=n,
                workload='64-frame synthetic detection stream (6 objects, 10 % dropped detections, 8-frame occlusion)')


def batched_association_line(dev, B=1024, T=32, M=16):
    """the association step of B independent 6-object sequences per step on the device (one wave per
    sequence, csrc/batched_assoc.hip) next to the native host tracker on one of them."""
    import numpy as np
    from stereotracking_amd.batched_assoc import BatchedGpuTracker
    from stereotracking_amd.synthetic import synthetic_detection_stream
    from stereotracking_amd.trackers import OCSORTTracker_Disparity
    try:
        cfg = dict(obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=False, match_iou_thr=0.1,
                   num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3, num_frames_retain=30)
        base = []
        for s in range(16):
            det = synthetic_detection_stream(200 + s, T=T, K=6)
            d, c = np.zeros((T, M, 8), np.float32), np.zeros(T, np.int32)
            for t in range(T):
                r = det[det[:, 0] == t]
                k = len(r)
                d[t, :k, 0:4], d[t, :k, 4], d[t, :k, 6], d[t, :k, 7], c[t] = r[:, 1:5], r[:, 5], r[:, 6], r[:, 7], k
            base.append((d, c))
        rec = np.zeros((T, M + 1, 13), np.float32)
        rec[:, 0, 0], rec[:, 0, 1], rec[:, 0, 2] = base[0][1], M, 1
        rec[:, 1:, 8:12], rec[:, 1:, 4:8] = base[0][0][:, :, 0:4], base[0][0][:, :, 4:8]
        host = OCSORTTracker_Disparity(**cfg)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            host.track_records(list(range(T)), rec)
            n += T
        host_us = (time.perf_counter() - t0) / n * 1e6
        dets = torch.from_numpy(np.stack([base[b % 16][0] for b in range(B)], 1)).to(dev)
        counts = torch.from_numpy(np.stack([base[b % 16][1] for b in range(B)], 1)).to(dev)
        fids = [torch.full((B,), t, dtype=torch.int32, device=dev) for t in range(T)]
        g = BatchedGpuTracker(B, max_tracks=32, max_dets=M, device=dev, **cfg)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(T):
                g.step(fids[t], dets[t], counts[t], check_status=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        return dict(sequences_per_step=B, ms_per_step=round(dt / T * 1e3, 4),
                    us_per_sequence_frame=round(dt / T / B * 1e6, 4), host_native_us_per_sequence_frame=round(host_us, 3),
                    overflow=bool(int(g.status.max()) != 0),
                    workload=f'{B} independent 6-object sequences x {T} frames, shipped tracker thresholds; ids equal '
                             'to the host tracker (tests/test_batched_assoc_gpu.py)')
    except Exception as e:   # a secondary line must never cost the headline
        return dict(error=repr(e))


def launch_ranks(args, argv, json_fd):
    """`bench.py --gpus N` started WITHOUT torch.distributed.run: start the N ranks from here.  The parent makes no GPU
    call (counting devices does not initialise one), runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` as a child, relays rank 0's single JSON
    line to its own stdout and returns the child's exit status (non-zero if ANY rank failed; a run whose ranks printed
    no line, or more than one, is a failure too).  Fewer than N visible devices is an error before anything starts -
    except under ST_BENCH_BACKEND=gloo, the one-card rehearsal of the N-rank code path.
    (reference: launcher handling tools/test.py:33-41, one process per GPU.)"""
    import socket
    import subprocess
    backend = os.environ.get('ST_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and ndev < args.gpus:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but only {ndev} device(s) visible; refusing to measure fewer '
                         'ranks than asked for\n')
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write('bench.py: no launcher in the environment, starting %d ranks: %s\n' % (args.gpus, ' '.join(cmd)))
    sys.stderr.flush()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=2, env=env, cwd=os.getcwd())
    out, _ = p.communicate()
    lines = [l for l in out.decode(errors='replace').splitlines() if l.startswith('{') and l.rstrip().endswith('}')]
    if p.returncode != 0:
        sys.stderr.write(f'bench.py: the {args.gpus}-rank run failed with status {p.returncode}\n')
        return p.returncode
    if len(lines) != 1:
        sys.stderr.write(f'bench.py: expected ONE JSON line from rank 0, got {len(lines)}\n')
        return 3
    if json.loads(lines[0]).get('n_gpus') != args.gpus:
        sys.stderr.write('bench.py: the line does not report n_gpus = %d\n' % args.gpus)
        return 4
    os.write(json_fd, (lines[0] + '\n').encode())
    return 0


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON of rank 0: libraries that print to file descriptor 1 (RCCL writes a
    # version banner there when its communicator initialises) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    launched = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if args.gpus > 1 and not launched:
        # `python bench.py --gpus N` without a launcher (the form the driver uses for N = 1): this process has made no
        # GPU call yet and never will - it starts the N ranks itself, relays their ONE JSON line and their exit status.
        # A --gpus N run can therefore never come back as a silent 1-rank measurement.
        raise SystemExit(launch_ranks(args, sys.argv[1:], json_fd))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or drop the launcher: '
                         'bench.py --gpus N starts its own ranks)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path is the only product path)')
    # one rank per GPU; ST_BENCH_BACKEND=gloo is a single-GPU REHEARSAL of the multi-rank code path (ranks share
    # cuda:0, the detection buffers are gathered through host memory) - never used for reported numbers
    backend = os.environ.get('ST_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and torch.cuda.device_count() < world:
        raise SystemExit(f'--gpus {world} but only {torch.cuda.device_count()} device(s) visible: one rank per GPU, '
                         'never two ranks on one card (ST_BENCH_BACKEND=gloo is the one-card rehearsal)')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    # ST_BENCH_WORLD1_PG=1: a process group of ONE rank - the one-GPU box then executes the RCCL branches of the N-rank
    # run for real (communicator init, one all-gather of the frame records per step on the communication stream,
    # barriers, the all-reduce of the timing); a rehearsal like ST_BENCH_BACKEND=gloo, never used for reported numbers
    pg = world > 1 or os.environ.get('ST_BENCH_WORLD1_PG') == '1'
    if pg:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == 'nccl' else torch.device('cpu')  # where collectives run

    from stereotracking_amd.pipeline import InflightPipelines
    from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict

    B = args.batch
    # `inflight` contexts (own workspace + HIP stream each): step i runs on context i % inflight, so the launch
    # tails and the latency-bound decode / NMS / depth kernels of one batch overlap the convs of the next
    runner = InflightPipelines(max(1, args.inflight), B, (720, 1280), 0.5, 0.33, 1, stereo=True,
                               max_disp=args.max_disp, max_det=args.max_det, agg_layers=args.agg_layers,
                               split_bf16=True if args.split_bf16 else None)
    pipe = runner.pipes[0]
    sd = synthetic_state_dict(runner.param_table(), seed=0)
    runner.load_state_dict(sd)   # plan from pipeline.default_tuning_cache() (committed), measured when absent
    # every rank gets its own 8 pairs (weak scaling: frames shard across ranks)
    # --input-batches DISTINCT batches, all resident in HBM before the timed region, fed round-robin: step i reads
    # batch i % nb (the timed loop of rounds 1-4 re-read ONE 181 MB batch, which the Infinity Cache can partly hold)
    nb = max(1, args.input_batches)
    batch_cpu = synthetic_batch([rank * B + i for i in range(B)], 720, 1280, args.max_disp)
    inputs = [(batch_cpu['img'].to(dev), batch_cpu['right'].to(dev))]
    for j in range(1, nb):
        bj = synthetic_batch([100000 * j + rank * B + i for i in range(B)], 720, 1280, args.max_disp)
        inputs.append((bj['img'].to(dev), bj['right'].to(dev)))
        del bj
    img, right = inputs[0]
    step_no = [0]
    from stereotracking_amd.dist import DetectionGatherer
    gathered = [torch.empty(world * B, pipe.max_det + 1, 8, device=cdev) for _ in runner.pipes] if pg else None
    # ONE communication stream for every all-gather of this rank: collectives are issued in host program order (step
    # i on every rank), behind an event of the producing context's stream - never from inside a context stream
    gatherer = DetectionGatherer(dev, single_rank_collective=pg and world == 1)

    def post(out, ctx):   # runs under the context's stream
        dets = pipe.pack_detections(out)   # self-describing frame records: header row (true count) + max_det rows
        if pg:  # ONE collective per shard of frames: the fixed-size records (8 x 32 KB / rank)
            out['records'], out['records_ready'] = gatherer.gather(dets, gathered[ctx])
        else:
            out['records'] = dets
        return out

    def step():
        l, r = inputs[step_no[0] % nb]
        step_no[0] += 1
        return runner.submit(l, r, post=post)[0]

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if pg:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if pg:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # self-validation for the day a multi-GPU node runs this: every rank reports (rank, local device index, device
    # uuid, name); N ranks must have used N DISTINCT devices
    import stereotracking_amd
    hw_queues = stereotracking_amd.effective_hw_queues()
    props = torch.cuda.get_device_properties(dev)
    me = dict(rank=rank, device_index=dev_index, uuid=str(getattr(props, 'uuid', '')), name=props.name)
    ranks_seen = [me]
    if pg:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
        if backend == 'nccl' and len({(r['uuid'] or r['device_index']) for r in ranks_seen}) != world:
            raise SystemExit(f'ranks share devices: {ranks_seen}')

    sustained = None
    if args.sustain_seconds > 0:     # a longer region for the eye of a GPU-busy sampler; `value` stays the K-step figure
        n_s, t1 = 0, time.perf_counter()
        while True:
            for _ in range(args.steps):
                out = step()
            n_s += args.steps
            torch.cuda.synchronize()
            go_on = time.perf_counter() - t1 < args.sustain_seconds
            if pg:   # every step carries a collective: the ranks must agree on how many more rounds they run
                flag = torch.tensor([1 if go_on else 0], device=cdev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                go_on = bool(flag.item())
            if not go_on:
                break
        dts = time.perf_counter() - t1
        if pg:
            t = torch.tensor([dts], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        sustained = dict(seconds=round(dts, 3), steps=n_s, value=round(world * B * n_s / dts, 3))

    if (step_no[0] - 1) % nb != 0:     # the record checks below compare against batch 0's CPU oracle: end on batch 0
        step_no[0] = 0
        out = step()
        torch.cuda.synchronize()
    counts = out['counts'].cpu().tolist()
    disp_pair0 = out['disp_postp'][0, 0].cpu()                   # BASELINE metric's "disparity L1 vs ref" (rank 0, pair 0)
    rec_counts = out['records'][:, 0, 0].cpu().long().tolist()   # what the tracker side of the all-gather sees
    if rec_counts[rank * B:(rank + 1) * B] != counts:
        raise SystemExit(f'gathered frame records disagree with the local counts: {rec_counts} vs {counts}')
    if max(rec_counts) > pipe.max_det:
        raise SystemExit(f'detection buffer overflow: kept {rec_counts} > max_det={pipe.max_det} (raise --max-det); '
                         'the reference applies no cap, a truncated run is not a valid measurement')
    line = {
        'metric': 'stereo frame-pairs/sec @1280x720 D=192', 'value': round(world * B * args.steps / dt, 3),
        'unit': 'stereo frame-pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'configs[1]: batch={B} synthetic 1280x720 stereo pairs per GPU, D={args.max_disp}, '
                               'full YOLOX-s two-branch backbone+PAFPN+head, cost volume at 1/4 res '
                               f'({args.max_disp // 4} levels) + {args.agg_layers} 3x3 aggregation convs + soft-argmin, '
                               'decode+NMS, per-box depth',
                   'global_batch': world * B, 'inflight_contexts': len(runner), 'distinct_input_batches': nb, 'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                   'hip_force_dev_kernarg': os.environ.get('HIP_FORCE_DEV_KERNARG'),
                   'parallelism': (f'frames sharded x{world}, one all-gather of detections per step ({backend})'
                                   if world > 1 else
                                   (f'ONE rank with a process group ({backend}): the all-gather of the step runs through the '
                                    'communicator (rehearsal of the collective code path)' if pg else
                                    'single process, no process group (no collective in the step)')),
                   'collectives_issued': gatherer.seq if pg else 0,
                   'ranks_seen': ranks_seen, 'hw_queues': hw_queues,
                   'tuning_plan': os.path.relpath(pipe.tuning_source, ROOT) if os.path.isabs(str(getattr(pipe, 'tuning_source', ''))) else str(getattr(pipe, 'tuning_source', None)),
                   'detections_kept_rank0': counts, 'max_det': pipe.max_det, 'detections_overflow': False},
        'sustained': sustained,
    }
    if rank == 0:
        roof = conv_roofline(pipe, img, right, max(3, min(args.steps, 10)))
        Hf, Wf = pipe.height // pipe.feat_stride, pipe.width // pipe.feat_stride
        roof['gflop_per_pair_conv'] = round(
            2.0 * (pipe.det.macs + pipe.agg_layers * pipe.stereo_module.agg_macs(B, Hf, Wf)) / B / 1e9, 3)
        # the whole detector against the fp32 MFMA roof = 66.96 GFLOP (direct-convolution count) per
        # pair x pairs/s / 157.3 TFLOP/s, from the TIMED region's throughput (all kernels, 3 contexts in flight)
        roof['pipeline_achieved_tflops'] = round(66.96e9 * line['value'] / world / 1e12, 3)
        roof['pipeline_frac'] = round(66.96e9 * line['value'] / world / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4)
        roof['pipeline_definition'] = ('SURVEY.md 8(d): 66.96 GFLOP of direct convolution per pair x pairs/s per GPU / '
                                       '157.3 TFLOP/s (Winograd layers counted at their direct-convolution flops: an '
                                       'algorithmic rate, may exceed what the pipes execute; the disparity stem executes K=36 of its 108 '
                                       'because disp_postp\'s three planes are identical: 1.09 of the 66.96 GFLOP are not executed)')
        roof['measured'] = ('separate serialized pass on one context after the timed region: with inflight > 1 the '
                            'timed region overlaps kernels of consecutive batches, which inflates per-launch durations '
                            '(compare profiles/*_inflight1 for the serialized rocprof summary)')
        line['roofline'] = roof
        # north_star's literal sizing (D=192 x 736 x 1280 per pair) and its "3D/2D aggregation" under the SAME clock as the
        # headline: both legs run by default (SURVEY 8 a-7 "two sizings to report"); neither ever touches `value`
        if world == 1 and not args.no_secondary_legs:
            line['secondary_agg3d'] = agg3d_leg(args, inputs, batch_cpu, line['value'], dev)
            line['secondary_full_resolution'] = fullres_leg(args, inputs, batch_cpu, line['value'], pipe.det.get_tuning(), dev)
        if world == 1 and not args.no_test_step:
            del runner   # its three workspaces are not needed any more
            line['test_step'] = test_step_leg(args, sd, batch_cpu, dev, B * args.steps)
            line['test_step']['vs_pipeline'] = round(line['test_step']['value'] / line['value'], 4)
            line['test_step']['primed_loop']['vs_pipeline'] = round(line['test_step']['primed_loop']['value'] / line['value'], 4)
            if line['test_step'].get('long_call'):
                line['test_step']['long_call']['vs_pipeline'] = round(line['test_step']['long_call']['value'] / line['value'], 4)
        line['parity'] = parity_records()
        # north_star's literal float bar ("floats within 1e-3 of the CPU path"), said at the TOP level per input sequence
        # of configs[2] (shipped thresholds): true on the blurred sequence, FALSE on the white-noise one (1.51e-3; a
        # property of the module spec - temperature-32 soft-argmin on texture-less matches - the fp32 oracle itself is
        # 1.03e-3 from float64 there; its consequence for the tracks is bounded in tests/test_config2_oracle_gpu.py)
        if line['parity']:
            line['within_1e3_of_cpu_path'] = {k.split('/')[0]: v['within_1e3_of_cpu_path']
                                              for k, v in line['parity'].items() if k.endswith('/shipped')}
            line['parity_source'] = 'committed record'
        line['tracker_cpu'] = tracker_cost()
        if world == 1 and args.split_leg and not args.split_bf16:
            line['secondary_split_bf16x3'] = split_leg(args, sd, img, right, line['value'])
        if world == 1:
            line['batched_gpu_association'] = batched_association_line(dev)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(sd, batch_cpu, args.max_disp, args.cpu_seconds, args.agg_layers, args.max_det)
            # BASELINE.json's metric string ends in "disparity L1 vs ref": the GPU's disparity of pair 0 (from the last
            # timed step) against the CPU oracle's disparity of the same pair (computed by the cpu_baseline leg above,
            # OUTSIDE the timed region).  "ref" = the oracle: the reference ships no stereo matcher.
            ref_d = line['cpu_baseline'].pop('oracle_disp_pair0')[0, 0]
            ad = (disp_pair0 - ref_d).abs()
            line['disparity_l1_vs_oracle'] = dict(
                l1_px=float(ad.mean()), max_abs_px=float(ad.max()),
                max_rel=float((ad / ref_d.abs().clamp(min=1.0)).max()), mean_disp_px=float(ref_d.mean()),
                pair='rank 0 pair 0 of the timed workload', ref='oracle (CPU fp32 restatement of the stereo module)')
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    if pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
