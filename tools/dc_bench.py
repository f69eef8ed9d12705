#!/usr/bin/env python
"""Direct 3x3 kernel (variant 42) vs the best generic tiles on the narrow 3x3 shapes of the path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_ablation.py')).read().split('\nfor shape in')[0]
ns = {'__file__': os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_ablation.py'), '__name__': 'm'}
exec(compile(src, ns['__file__'], 'exec'), ns)
for shape in [(8, 184, 320, 48, 48, 3, 1), (16, 184, 320, 32, 32, 3, 1), (8, 92, 160, 64, 64, 3, 1), (8, 184, 320, 64, 64, 3, 1)]:
    for v in (3, 13, 18, 5, 42):
        try:
            ms, tf = ns['bench'](*shape, v)
            print(f'shape {shape} variant {v:3d}: {ms * 1e3:8.1f} us  {tf:7.1f} TF/s', flush=True)
        except Exception as e:  # noqa: BLE001
            print(f'shape {shape} variant {v}: {str(e)[:80]}')
