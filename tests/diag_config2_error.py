"""Diagnostic (not a test): where does the end-to-end float error of a configs[2] frame come from?  Compares, for a few
frames of the config2 fixture's sequence, GPU vs oracle: disparity; head end to end; head with the oracle detector fed
the GPU's disparity (conv error alone); the same for the untuned (implicit-GEMM only) plan.  Run on the GPU box:
    python tests/diag_config2_error.py [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

from oracle import stereo as ostereo  # noqa: E402
from oracle.torch_model import head_to_rows  # noqa: E402
from parity_utils import make_oracle, rel_err  # noqa: E402
from stereotracking_amd.pipeline import StereoDensePipeline  # noqa: E402
from stereotracking_amd.sequence import synthetic_sequence  # noqa: E402
from stereotracking_amd.synthetic import pad_to_divisor, synthetic_batch, synthetic_state_dict  # noqa: E402


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device('cuda:0')
    H, W, D, AGG = 720, 1280, 192, 2
    torch.set_num_threads(16)
    frames = list(synthetic_sequence(nf, 6, H, W, D, seed=3))
    smooth = synthetic_batch(list(range(nf)), H, W, D)
    for tuned in (True, False):
        pipe = StereoDensePipeline(1, (H, W), 0.5, 0.33, 1, stereo=True, max_disp=D, agg_layers=AGG)
        sd = synthetic_state_dict(pipe.param_table(), seed=0)
        pipe.load_state_dict(sd, autotune=tuned, tuning_cache=False)
        ora = make_oracle(sd)
        for kind in ('noise sequence', 'smoothed pairs'):
            for t in range(nf):
                if kind == 'noise sequence':
                    img = torch.from_numpy(pad_to_divisor(frames[t]['left'].astype(np.float32), 32, 114.0))[None]
                    right = torch.from_numpy(pad_to_divisor(frames[t]['right'].astype(np.float32), 32, 114.0))[None]
                else:
                    img, right = smooth['img'][t:t + 1], smooth['right'][t:t + 1]
                out = pipe.run(img.to(dev), right.to(dev))
                torch.cuda.synchronize()
                with torch.no_grad():
                    fl = ora.backbone.stage1_features(img).permute(0, 2, 3, 1).contiguous().numpy()
                    fr = ora.backbone.stage1_features(right).permute(0, 2, 3, 1).contiguous().numpy()
                    cost, lr, disp = ostereo.disparity(fl, fr, fl.shape[-1], D // 4, pipe.temperature, sd, AGG,
                                                       valid_hw=(H, W))
                    disp = torch.from_numpy(disp)
                    rows_e2e = head_to_rows(*ora(dict(img=img, disp_postp=disp)))
                    rows_stage = head_to_rows(*ora(dict(img=img, disp_postp=out['disp_postp'].cpu())))
                    ora64 = ora.double()
                    rows64 = head_to_rows(*ora64(dict(img=img.double(), disp_postp=out['disp_postp'].cpu().double())))
                    ora.float()
                gfeat = pipe.det.tap('stage1_rgb').cpu().numpy()
                e_feat = rel_err(gfeat[:1], fl)
                dd = (out['disp_postp'].cpu() - disp).abs()
                got = [r[..., :6].cpu() for r in pipe.det.head_levels(out['head'])]
                e_e2e = max(rel_err(a, b) for a, b in zip(got, rows_e2e))
                e_stage = max(rel_err(a, b) for a, b in zip(got, rows_stage))
                e_gpu64 = max(rel_err(a, b) for a, b in zip(got, rows64))
                e_cpu64 = max(rel_err(a, b) for a, b in zip(rows_stage, rows64))
                print(f'tuned={tuned} {kind} frame {t}: feat {e_feat:.2e}  disp max|d| {dd.max():.3e} px '
                      f'(rel {rel_err(out["disp_postp"].cpu(), disp):.2e}, >1e-3px: {(dd > 1e-3).float().mean():.2e})  '
                      f'head e2e {e_e2e:.2e}  head stagewise {e_stage:.2e}  gpu-vs-fp64 {e_gpu64:.2e}  '
                      f'cpu32-vs-fp64 {e_cpu64:.2e}  |head|max {max(float(b.abs().max()) for b in rows_stage):.1f}',
                      flush=True)


if __name__ == '__main__':
    main()
