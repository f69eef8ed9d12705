#!/usr/bin/env python
"""Concurrency stress of the cost-volume kernel and the aggregation conv (Winograd K-tail instance): several streams,
white-noise-like features, every result compared with the serial run."""
import ctypes as C
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the split-operand (bf16x3) instances are parked in the TOOLS build of the library (round 5)
os.environ.setdefault('ST_LIBRARY', os.path.join(_ROOT, 'stereotracking_amd', 'lib', 'libstereotrack_hip_ablation.so'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
N, Hf, Wf, Cc, D = 8, 184, 320, 64, 48
NS = 4
feats = [(torch.randn(N, Hf, Wf, Cc, device=dev), torch.randn(N, Hf, Wf, Cc, device=dev)) for _ in range(NS)]
vols = [torch.empty(N, Hf, Wf, D, device=dev) for _ in range(NS)]
aggs = [torch.empty(N, Hf, Wf, D, device=dev) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
# aggregation conv 48 -> 48 (Winograd instance 43)
torch.manual_seed(0)
w = torch.randn(D, D, 3, 3) * 0.05
b = torch.randn(D) * 0.01
wp = torch.empty(lib.st_conv_packed_floats(D, D, 3, 3)); bp = torch.empty(64)
check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, D, D, 3, 3, ptr(wp), ptr(bp)))
wn = torch.empty(lib.st_wino_packed_floats(D, D)); check(lib.st_wino_pack_weights(ptr(wp), D, D, ptr(wn)))
wpd, bpd, wnd = wp.to(dev), bp.to(dev), wn.to(dev)


def agg_desc(src, dst):
    d = StConvDesc()
    d.in_dev = src.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, Hf, Wf, D, D, 0
    d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr(); d.wgt_wino_dev = wnd.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = D, 3, 3, 1, 1
    d.out1_dev = dst.data_ptr(); d.out1_ld, d.out1_off, d.split = D, 0, D
    d.act, d.post_scale = 1, 1.0
    return d


def launch(i, s):
    fl, fr = feats[i]
    st = C.c_void_p(s.cuda_stream)
    check(lib.st_costvolume_softargmin(ptr(fl), ptr(fr), N, Hf, Wf, Cc, Cc, D, 32.0, ptr(vols[i]), None, st))
    src = refvol[i] if refvol else vols[i]       # victim 2: the Winograd conv on a CONSTANT input
    check(lib.st_conv2d_nhwc_variant(C.byref(agg_desc(src, aggs[i])), st, 43))


# a layer on the split-operand conv instance, co-running on extra streams
xs = torch.randn(8, 92, 160, 256, device=dev)
ws = torch.randn(128, 256, 1, 1) / 16
wps = torch.empty(lib.st_conv_packed_floats(128, 256, 1, 1)); bps = torch.zeros(128)
check(lib.st_conv_pack_weights(ptr(ws), ptr(bps), None, None, None, None, 0.0, 128, 256, 1, 1, ptr(wps), ptr(bps)))
wpsd, bpsd = wps.to(dev), bps.to(dev)
outs_s = torch.empty(8, 92, 160, 128, device=dev)
ds = StConvDesc()
ds.in_dev = xs.data_ptr(); ds.N, ds.Hi, ds.Wi, ds.Cin, ds.in_ld, ds.in_off = 8, 92, 160, 256, 256, 0
ds.wgt_dev = wpsd.data_ptr(); ds.bias_dev = bpsd.data_ptr()
ds.Cout, ds.KH, ds.KW, ds.stride, ds.pad = 128, 1, 1, 1, 0
ds.out1_dev = outs_s.data_ptr(); ds.out1_ld, ds.out1_off, ds.split = 128, 0, 128
ds.act, ds.post_scale = 1, 1.0
extra = [torch.cuda.Stream() for _ in range(2)]
# aggressor: the library's split instances (default), or `torch`: bf16 GEMMs of ANOTHER library in the process
# (torch.matmul -> hipBLASLt / rocBLAS kernels issuing bf16 MFMAs) - the case the product library has to survive
# or `micro`: the register-only v_mfma_f32_16x16x32_bf16 loop of tests/helpers/mfma_aggressor.hip (what the -m gpu co-run
# test uses)
TORCH_AGG = len(sys.argv) > 1 and sys.argv[1] == 'torch'
MICRO_AGG = len(sys.argv) > 1 and sys.argv[1] == 'micro'
SPLITV = [] if (TORCH_AGG or MICRO_AGG) else [int(v) for v in (sys.argv[1:] or ['53', '54', '50', '51'])]
if MICRO_AGG:
    agg_lib = C.CDLL(os.path.join(_ROOT, 'tests', 'helpers', 'libmfma_aggressor.so'))
    agg_lib.st_test_bf16_mfma_busy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    agg_scratch = torch.zeros(65536, device=dev)
    AGG_VARIANT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ga = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
gb = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)

ref = []
refvol = []
for i in range(NS):
    launch(i, torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref.append((vols[i].clone(), aggs[i].clone()))
refvol = [r[0].clone() for r in ref]
bad_v = bad_a = 0
for rep in range(60):
    for i in range(NS):
        vols[i].fill_(float('nan')); aggs[i].fill_(float('nan'))
    torch.cuda.synchronize()
    for i, s in enumerate(streams):
        if i < len(extra):
            for v in SPLITV:
                check(lib.st_conv2d_nhwc_variant(C.byref(ds), C.c_void_p(extra[i].cuda_stream), v))
            if MICRO_AGG:
                agg_lib.st_test_bf16_mfma_busy(agg_scratch.data_ptr(), 3000, AGG_VARIANT, extra[i].cuda_stream)
            if TORCH_AGG:
                with torch.cuda.stream(extra[i]):
                    for _ in range(3):
                        torch.matmul(ga, gb)
        launch(i, s)
    torch.cuda.synchronize()
    for i in range(NS):
        ev = not torch.equal(vols[i], ref[i][0])
        ea = not torch.equal(aggs[i], ref[i][1])
        bad_v += ev; bad_a += ea
        if ev or ea:
            dv = (vols[i] - ref[i][0]).abs().nan_to_num(9e9); da = (aggs[i] - ref[i][1]).abs().nan_to_num(9e9)
            nz = torch.nonzero(dv.amax(-1) > 0)
            print(f'rep {rep} stream {i}: volume differs {ev} (max {dv.max().item():.2e}, {len(nz)} px, first {nz[:2].tolist()} last {nz[-2:].tolist()}) agg differs {ea} (max {da.max().item():.2e})', flush=True)
print('bad volume', bad_v, 'bad agg', bad_a)
