"""Output containers with the reference's field names: a small InstanceData look-alike (mmengine
is not installed) and TrackDataSample (reference mmtrack/structures/track_data_sample.py:5-107:
gt_instances, pred_det_instances, pred_track_instances, dense_depth_map + metainfo such as
frame_id, ori_shape, img_shape, scale_factor, pad_param)."""
import copy

import numpy as np
import torch


class InstanceData:
    """Fields of equal first dimension; supports len(), data['bboxes'], data.bboxes, data[mask],
    `'key' in data`, clone(), and item assignment like mmengine.structures.InstanceData."""

    def __init__(self, metainfo=None, **fields):
        object.__setattr__(self, '_fields', {})
        object.__setattr__(self, '_meta', dict(metainfo or {}))
        for k, v in fields.items():
            self[k] = v

    # field access ------------------------------------------------------------------------------
    def __setattr__(self, name, value):
        self[name] = value

    def __getattr__(self, name):
        fields = object.__getattribute__(self, '_fields')
        if name in fields:
            return fields[name]
        meta = object.__getattribute__(self, '_meta')
        if name in meta:
            return meta[name]
        raise AttributeError(name)

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise TypeError('InstanceData item assignment needs a string key')
        n = len(self)
        if self._fields and hasattr(value, '__len__') and len(value) != n:
            raise ValueError(f'field {key} has length {len(value)}, expected {n}')
        self._fields[key] = value

    def __getitem__(self, item):
        if isinstance(item, str):
            return self._fields[item]
        out = InstanceData(metainfo=self._meta)
        for k, v in self._fields.items():
            if isinstance(v, (torch.Tensor, np.ndarray)):
                idx = item
                if isinstance(v, np.ndarray) and isinstance(item, torch.Tensor):
                    idx = item.cpu().numpy()
                sel = v[idx]
                if sel.ndim == v.ndim - 1:  # integer index keeps the instance dimension
                    sel = sel[None]
                out._fields[k] = sel
            elif isinstance(v, list):
                ids = torch.arange(len(v))[item.cpu() if isinstance(item, torch.Tensor) else item]
                out._fields[k] = [v[int(i)] for i in ids.reshape(-1)]
            else:
                out._fields[k] = v
        return out

    def __contains__(self, key):
        return key in self._fields or key in self._meta

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def get(self, key, default=None):
        return self._fields.get(key, self._meta.get(key, default))

    def keys(self):
        return list(self._fields.keys())

    def items(self):
        return list(self._fields.items())

    def clone(self):
        out = InstanceData(metainfo=copy.deepcopy(self._meta))
        for k, v in self._fields.items():
            out._fields[k] = v.clone() if isinstance(v, torch.Tensor) else copy.deepcopy(v)
        return out

    def to(self, *args, **kwargs):
        out = InstanceData(metainfo=self._meta)
        for k, v in self._fields.items():
            out._fields[k] = v.to(*args, **kwargs) if isinstance(v, torch.Tensor) else v
        return out

    def cpu(self):
        return self.to('cpu')

    def set_metainfo(self, meta):
        self._meta.update(meta)

    @property
    def metainfo(self):
        return dict(self._meta)

    def __repr__(self):
        return f'InstanceData(n={len(self)}, fields={self.keys()})'


class TrackDataSample:
    """Per-frame container.  Metainfo keys are attributes too (sample.frame_id, sample.ori_shape)."""

    _DATA_KEYS = ('gt_instances', 'ignored_instances', 'proposals', 'pred_det_instances',
                  'pred_track_instances', 'pred_instances', 'dense_depth_map')

    def __init__(self, metainfo=None):
        object.__setattr__(self, '_meta', dict(metainfo or {}))
        object.__setattr__(self, '_data', {})

    def set_metainfo(self, meta):
        self._meta.update(meta)

    @property
    def metainfo(self):
        return dict(self._meta)

    def get(self, key, default=None):
        if key in self._data:
            return self._data[key]
        return self._meta.get(key, default)

    def __contains__(self, key):
        return key in self._data or key in self._meta

    def __getattr__(self, name):
        data = object.__getattribute__(self, '_data')
        if name in data:
            return data[name]
        meta = object.__getattribute__(self, '_meta')
        if name in meta:
            return meta[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self._DATA_KEYS:
            if name != 'dense_depth_map' and not isinstance(value, InstanceData):
                raise TypeError(f'{name} must be an InstanceData')
            self._data[name] = value
        else:
            self._meta[name] = value

    def __delattr__(self, name):
        self._data.pop(name, None)
        self._meta.pop(name, None)

    def __repr__(self):
        return f'TrackDataSample(meta={list(self._meta)}, data={list(self._data)})'
