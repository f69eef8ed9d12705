#!/usr/bin/env python
"""One conv layer shape x a few tile variants, a handful of launches each: run under
   rocprofv3 --kernel-trace --pmc <counters> to account for where the SIMD cycles of the conv kernel go."""
import os
import sys

sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location('conv_ablation_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_ablation.py'))
src = open(spec.origin).read().split('\nfor shape in')[0]
ns = {'__file__': spec.origin, '__name__': 'conv_ablation_mod'}
exec(compile(src, spec.origin, 'exec'), ns)
for v in (0, 14, 7, 100 + 0):   # 128x128, 64x128dma, 64x128, and the no-global-load ablation of 128x128
    ms, tf = ns['bench'](8, 92, 160, 128, 256, 3, 1, v, reps=5)
    print(f'variant {v}: {ms * 1e3:.1f} us {tf:.1f} TF/s')
