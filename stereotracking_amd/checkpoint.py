"""Checkpoint loading for the plugin surface: what `mmengine.runner.load_checkpoint` (the runner's `load_from`,
reference tools/test.py) and the detector's `ColorPretrained` init (reference
mmtrack/models/detectors/yolo_detector_disparity_v1.py:144-166) do, restated for an offline host: local files only."""
import re

import torch

_REMOTE = ('http://', 'https://', 'open-mmlab://', 'openmmlab://', 'mmcls://', 'torchvision://', 's3://', 'petrel://')


def read_checkpoint(filename, map_location='cpu', trusted=False):
    """-> the checkpoint dict (or bare state_dict) of a LOCAL file.  `trusted=False` unpickles tensors and plain
    containers only (torch.load(weights_only=True)); pass trusted=True for checkpoints that carry other objects."""
    if str(filename).startswith(_REMOTE):
        raise RuntimeError(f'{filename}: remote checkpoints cannot be fetched here; download the file and pass its path '
                           f'(the shipped config names a URL: configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:46)')
    return torch.load(filename, map_location=map_location, weights_only=not trusted)


def _state_dict_of(ckpt):
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    if not isinstance(sd, dict):
        raise RuntimeError('checkpoint holds no state_dict')
    return sd


def load_matching(module, state_dict, strict=False):
    """module.load_state_dict restricted to the entries whose name AND shape match (mmengine's load_state_dict reports a
    size mismatch and goes on where torch raises) -> dict(missing, unexpected, mismatched)."""
    own = module.state_dict()
    ok, mismatched = {}, []
    for k, v in state_dict.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            mismatched.append((k, tuple(v.shape), tuple(own[k].shape)))
        else:
            ok[k] = v
    if strict and mismatched:
        raise RuntimeError(f'size mismatch for {mismatched[:5]}')
    res = module.load_state_dict(ok, strict=False)
    missing = [k for k in res.missing_keys]
    unexpected = list(res.unexpected_keys)
    if strict and (missing or unexpected):
        raise RuntimeError(f'missing keys {missing[:5]}, unexpected keys {unexpected[:5]}')
    return dict(missing=missing, unexpected=unexpected, mismatched=mismatched)


def load_checkpoint(model, filename, map_location='cpu', strict=False, revise_keys=((r'^module\.', ''),), trusted=False):
    """mmengine.runner.load_checkpoint's call shape: load `filename` into `model` (non-strict by default, DataParallel's
    `module.` prefix stripped) and return the checkpoint.  The report of what did not load is in ckpt['_load_report']."""
    ckpt = read_checkpoint(filename, map_location, trusted)
    sd = dict(_state_dict_of(ckpt))
    for pat, rep in revise_keys:
        sd = {re.sub(pat, rep, k): v for k, v in sd.items()}
    report = load_matching(model, sd, strict)
    if isinstance(ckpt, dict):
        ckpt['_load_report'] = report
    return ckpt


def color_pretrained_state_dict(state_dict):
    """The reference's `ColorPretrained` rule (yolo_detector_disparity_v1.py:155-163): the disparity branch starts from the
    RGB branch's weights - every entry whose name contains `stem` is ALSO loaded under that name with `stem` ->
    `disp_stem`, every entry containing `stage1` also under `stage1` -> `disp_stage1`."""
    out = dict(state_dict)
    for name, param in state_dict.items():
        if 'stem' in name:
            out[name.replace('stem', 'disp_stem')] = param
        if 'stage1' in name:
            out[name.replace('stage1', 'disp_stage1')] = param
    return out
