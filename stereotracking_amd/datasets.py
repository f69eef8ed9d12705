"""AirDrone reader for BASELINE configs[4]: CocoVID json -> per-video frame lists with left / right / disparity /
depth paths and the GT `instances`, and the test-time loaders that turn the PNG files into the RAW bytes the device
input pipeline consumes (uint8 pixels, uint16 disparity codes -> RawFrameUploader / st_pack_raw_inputs).

Mirrors, for the test-time path only:
  MOTDispDataset.parse_data_info        reference mmtrack/datasets/mot_disp_dataset.py:38-97 (paths, instance filter rules)
  BaseVideoDataset._load_video_data_list  mmtrack/datasets/base_video_dataset.py:104-148 (CocoVID traversal, video_length)
  VideoSampler                          mmtrack/datasets/samplers/video_sampler.py:25-70 (whole videos per rank)
  LoadDisparityFromFile                 mmtrack/datasets/transforms/loading_disparity.py:71-134 (uint16, 65535 invalid, /16)
  LoadDepthFromFile                     :197-260 ('airsim' in the path: value / 100 = metres)
  PackTrackInputs_Disparity             mmtrack/datasets/transforms/formatting_disparity.py:139-338 (HWC -> (T,C,H,W), metainfo)
mmcv / OpenCV are not available, so PNG files are decoded here: chunk parsing + inflate with the
standard library, scanline un-filtering by the native st_png_unfilter.  What is NOT mirrored: training pipelines,
augmentation, Resize to another scale (the shipped test pipeline resizes 1280x720 to itself).
"""
import json
import os
import struct
import zlib
from collections import defaultdict

import numpy as np

from .registry import DATASETS, DATA_SAMPLERS, TRANSFORMS

_PNG_SIG = b'\x89PNG\r\n\x1a\n'
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}       # PNG colour type -> samples per pixel


# ---- PNG ---------------------------------------------------------------------------------------------------------
def read_png(path_or_bytes):
    """Decode a non-interlaced 8- or 16-bit PNG (gray / gray+alpha / RGB / RGBA) -> numpy (H, W) or (H, W, C), dtype
    uint8 / uint16, channel order as stored (RGB).  Equivalent of cv2.imdecode(..., IMREAD_UNCHANGED) up to OpenCV's
    BGR order (see LoadImageFromFile)."""
    import ctypes as C
    from . import _lib
    data = path_or_bytes
    if not isinstance(data, (bytes, bytearray, memoryview)):
        with open(path_or_bytes, 'rb') as f:
            data = f.read()
    if bytes(data[:8]) != _PNG_SIG:
        raise ValueError('not a PNG file')
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, typ = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif typ == b'IDAT':
            idat.append(bytes(body))
        elif typ == b'IEND':
            break
    if hdr is None:
        raise ValueError('PNG without IHDR')
    w, h, depth, ctype, _, _, interlace = hdr
    if depth not in (8, 16) or ctype not in _CHANNELS or interlace != 0:
        raise NotImplementedError(f'PNG bit depth {depth} / colour type {ctype} / interlace {interlace} is not supported')
    ch = _CHANNELS[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    raw = zlib.decompress(b''.join(idat))
    if len(raw) != h * (stride + 1):
        raise ValueError(f'PNG payload has {len(raw)} bytes, expected {h * (stride + 1)}')
    out = np.empty((h, stride), np.uint8)
    lib = _lib.load()
    _lib.check(lib.st_png_unfilter(C.c_char_p(raw), h, stride, bpp, C.c_void_p(out.ctypes.data)), 'st_png_unfilter')
    if depth == 16:
        out = out.view('>u2').astype(np.uint16)       # PNG stores big-endian samples
    out = out.reshape(h, w, ch)
    return out[..., 0] if ch == 1 else out


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def write_png(path, arr, filters=None, level=6):
    """Encode uint8 / uint16 (H,W) or (H,W,C in 1..4) as PNG.  `filters`: one filter type (0-4) or a per-row sequence
    (the tests cycle through all five so that the decoder's every branch runs); default 0."""
    arr = np.asarray(arr)
    if arr.ndim == 2:
        arr = arr[..., None]
    h, w, ch = arr.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    if arr.dtype == np.uint16:
        depth, rows = 16, arr.astype('>u2').view(np.uint8).reshape(h, w * ch * 2)
    elif arr.dtype == np.uint8:
        depth, rows = 8, arr.reshape(h, w * ch)
    else:
        raise TypeError('write_png takes uint8 or uint16 arrays')
    bpp = ch * depth // 8
    if filters is None:
        filters = 0
    ftypes = [int(filters)] * h if np.isscalar(filters) else [int(f) for f in filters]
    zero = np.zeros_like(rows[0])
    out = bytearray()
    for y in range(h):
        cur = rows[y]
        prev = rows[y - 1] if y else zero
        left = np.concatenate([np.zeros(bpp, np.uint8), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.uint8), prev[:-bpp]])
        ft = ftypes[y]
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left.astype(np.int32) + prev) >> 1).astype(np.uint8)
        elif ft == 4:
            f = cur - _paeth(left, prev.astype(np.int32), upleft.astype(np.int32))
        else:
            raise ValueError('PNG filter types are 0..4')
        out.append(ft)
        out += f.astype(np.uint8).tobytes()

    def chunk(typ, body):
        return struct.pack('>I', len(body)) + typ + body + struct.pack('>I', zlib.crc32(typ + body) & 0xFFFFFFFF)
    blob = (_PNG_SIG + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, ctype, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(bytes(out), level)) + chunk(b'IEND', b''))
    with open(path, 'wb') as f:
        f.write(blob)


# ---- CocoVID annotation file -> per-video data lists ---------------------------------------------------------------
@DATASETS.register_module(name=['MOTDispDataset', 'mmtrack.MOTDispDataset'])
class MOTDispDataset:
    """Test-time view of the reference dataset class (mot_disp_dataset.py:11-104 on base_video_dataset.py:104-148):
    `data_list` in the reference's order (videos by id, images of a video by frame_id), each entry carrying
    img_path / disp_path / depth_path (+ right_path: the right image the stereo module consumes; the reference never
    reads it), frame_id, video_length, height / width and the filtered GT `instances`."""

    METAINFO = {'CLASSES': ('drone',)}

    def __init__(self, ann_file, data_root='', data_prefix=None, disparity_dir_name='disparity', depth_dir_name=None,
                 right_dir_name='right', metainfo=None, pipeline=None, test_mode=True, load_as_video=True,
                 ref_img_sampler=None, detection_file=None, **kwargs):
        if not load_as_video:
            raise NotImplementedError('image-mode loading is a training feature (out of scope)')
        self.data_root = data_root
        self.ann_file = ann_file if os.path.isabs(ann_file) else os.path.join(data_root, ann_file)
        self.data_prefix = {k: (v if os.path.isabs(v) else os.path.join(data_root, v))
                            for k, v in (data_prefix or {}).items()}
        self.disparity_dir_name, self.depth_dir_name, self.right_dir_name = disparity_dir_name, depth_dir_name, right_dir_name
        self.metainfo = dict(self.METAINFO, **({k.upper() if k == 'classes' else k: v for k, v in (metainfo or {}).items()}))
        self.pipeline = [TRANSFORMS.build(t) if isinstance(t, dict) else t for t in (pipeline or [])]
        self.test_mode = test_mode
        self._load()

    def _load(self):
        with open(self.ann_file) as f:
            coco = json.load(f)
        names = tuple(self.metainfo['CLASSES'])
        self.cat_ids = [c['id'] for c in coco['categories'] if c['name'] in names]   # CocoVID.get_cat_ids order
        self.cat2label = {cid: i for i, cid in enumerate(self.cat_ids)}
        by_video, anns_of = defaultdict(list), defaultdict(list)
        for img in coco['images']:
            by_video[img['video_id']].append(img)
        for ann in coco.get('annotations', []):
            anns_of[ann['image_id']].append(ann)
        self.videos = {v['id']: v['name'] for v in coco['videos']}
        self.data_list, self.video_first = [], []
        for vid in sorted(self.videos):                                   # coco.get_vid_ids()
            imgs = sorted(by_video[vid], key=lambda im: im['frame_id'])   # get_img_ids_from_vid: by frame_id
            self.video_first.append(len(self.data_list))
            for img in imgs:
                raw = dict(img, img_id=img['id'], video_length=len(imgs))
                anns = [a for a in anns_of[img['id']] if a['category_id'] in self.cat_ids]
                self.data_list.append(self.parse_data_info(dict(raw_img_info=raw, raw_ann_info=anns)))

    def parse_data_info(self, raw_data_info):
        """mot_disp_dataset.py:38-97, statement for statement in behaviour (path rewriting by replacing 'left' in the
        file name; instances dropped when ignored / outside the image / degenerate / of another category)."""
        img_info, ann_info = raw_data_info['raw_img_info'], raw_data_info['raw_ann_info']
        info = dict(img_info)
        fname = img_info['file_name']
        prefix = self.data_prefix.get('img_path')
        img_path = os.path.join(prefix, fname) if prefix is not None else fname
        info['img_path'] = img_path
        info['disp_path'] = img_path.replace(fname, fname.replace('left', self.disparity_dir_name))
        if self.depth_dir_name is not None:
            info['depth_path'] = img_path.replace(fname, fname.replace('left', self.depth_dir_name))
        info['right_path'] = img_path.replace(fname, fname.replace('left', self.right_dir_name))
        instances = []
        for ann in ann_info:
            if ann.get('ignore', False):
                continue
            x1, y1, w, h = ann['bbox']
            inter_w = max(0, min(x1 + w, img_info['width']) - max(x1, 0))
            inter_h = max(0, min(y1 + h, img_info['height']) - max(y1, 0))
            if inter_w * inter_h == 0:
                continue
            if ann['area'] <= 0 or w < 1 or h < 1:
                continue
            if ann['category_id'] not in self.cat_ids:
                continue
            instances.append(dict(ignore_flag=1 if ann.get('iscrowd', False) else 0, instance_id=ann['instance_id'],
                                  category_id=ann['category_id'], bbox_label=self.cat2label[ann['category_id']],
                                  bbox=[x1, y1, x1 + w, y1 + h], location=ann['location'], mot_conf=ann['mot_conf'],
                                  visibility=ann['visibility']))
        info['instances'] = instances
        return info

    def __len__(self):
        return len(self.data_list)

    def get_data_info(self, idx):
        return dict(self.data_list[idx], cat2label=self.cat2label)

    def __getitem__(self, idx):
        results = self.get_data_info(idx)
        for t in self.pipeline:
            results = t(results)
            if results is None:
                return None
        return results

    def video_indices(self):
        """[(video name, [data_list indices in frame order])] - what VideoSampler iterates."""
        ends = self.video_first[1:] + [len(self.data_list)]
        names = [self.videos[v] for v in sorted(self.videos)]
        return [(n, list(range(a, b))) for n, a, b in zip(names, self.video_first, ends)]


@DATA_SAMPLERS.register_module(name=['VideoSampler', 'mmtrack.VideoSampler'])
class VideoSampler:
    """video_sampler.py:25-70: the list of first-frame indices is np.array_split over the ranks; a rank iterates its
    videos one after the other, frame by frame."""

    def __init__(self, dataset, rank=None, world_size=None, seed=None):
        from . import dist as sdist
        r, w = sdist.world()
        self.rank = r if rank is None else rank
        self.world_size = w if world_size is None else world_size
        vids = dataset.video_indices()
        chunks = np.array_split(np.arange(len(vids)), self.world_size)
        self.videos = [vids[i] for i in chunks[self.rank]]
        self.indices = [i for _, idx in self.videos for i in idx]

    def __iter__(self):
        return iter(self.indices)

    def __len__(self):
        return len(self.indices)


# ---- test-time transforms ------------------------------------------------------------------------------------------
@TRANSFORMS.register_module(name=['LoadImageFromFile', 'mmcv.LoadImageFromFile'])
class LoadImageFromFile:
    """mmcv LoadImageFromFile [upstream-memory]: uint8 (h,w,3) in BGR order (OpenCV's), img_shape / ori_shape set."""

    def __init__(self, to_float32=False, key='img_path', out='img', **kwargs):
        self.to_float32, self.key, self.out = to_float32, key, out

    def __call__(self, results):
        img = read_png(results[self.key])
        if img.ndim == 2:
            img = np.repeat(img[..., None], 3, -1)
        img = np.ascontiguousarray(img[..., 2::-1])       # stored RGB(A) -> BGR
        results[self.out] = img.astype(np.float32) if self.to_float32 else img
        if self.out == 'img':
            results['img_shape'] = results['ori_shape'] = img.shape[:2]
        return results


@TRANSFORMS.register_module(name=['LoadDisparityFromFile', 'mmtrack.LoadDisparityFromFile'])
class LoadDisparityFromFile:
    """loading_disparity.py:13-134.  `raw_codes=True` (this package's device path) additionally keeps the uint16 PNG
    codes under 'disp_codes': they cross PCIe as 2 bytes per pixel and st_pack_raw_inputs applies the same
    post-processing on the device (65535 -> 0, / 16, x3 channels, mask)."""

    def __init__(self, to_float32=True, to_3channel=False, post_processing=None, ignore_empty=False, raw_codes=True,
                 **kwargs):
        self.to_float32, self.to_3channel, self.post_processing = to_float32, to_3channel, post_processing
        self.ignore_empty, self.raw_codes = ignore_empty, raw_codes

    def __call__(self, results):
        try:
            disp = read_png(results['disp_path'])
        except Exception:
            if self.ignore_empty:
                return None
            raise
        if disp.ndim != 2:
            disp = disp[..., 0]
        if self.raw_codes:
            results['disp_codes'] = disp.astype(np.uint16)
        results['disp_mask'] = (disp < 65535).astype(np.uint8)[:, :, None]            # :82-83
        disp = np.repeat(disp[:, :, None], 3, axis=-1) if self.to_3channel else disp[:, :, None]
        if self.to_float32:
            disp = disp.astype(np.float32)
        results['disp'] = disp
        if results.get('img_shape') is None:
            results['img_shape'] = results['ori_shape'] = disp.shape[:2]
        if self.post_processing is not None:                                              # _post_processing_v2 :129-134
            dp = disp.copy()
            dp[dp == 65535] = 0
            results['disp_postp'] = dp.astype(np.float32) / 16.
        else:
            results['disp_postp'] = disp
        return results


@TRANSFORMS.register_module(name=['LoadDepthFromFile', 'mmtrack.LoadDepthFromFile'])
class LoadDepthFromFile:
    """loading_disparity.py:147-286, the AirSim branch: depth = value / 100 (metres) when 'airsim' is in the path."""

    def __init__(self, to_float32=True, to_3channel=False, post_processing=None, ignore_empty=False, **kwargs):
        self.to_float32, self.to_3channel, self.ignore_empty = to_float32, to_3channel, ignore_empty
        if post_processing is not None:
            raise NotImplementedError('depth post_processing (visualisation scaling) is out of scope')

    def __call__(self, results):
        path = results['depth_path']
        try:
            depth = read_png(path)
        except Exception:
            if self.ignore_empty:
                return None
            raise
        if 'airsim' in path.lower():
            depth = depth / 100.
        else:
            raise NotImplementedError('only the AirSim depth encoding is restated (the SELMA branch is unused by the '
                                      'stereo configs)')
        depth = np.repeat(depth[:, :, None], 3, axis=-1) if self.to_3channel else depth[:, :, None]
        if self.to_float32:
            depth = depth.astype(np.float32)
        results['depth'] = results['depth_postp'] = depth
        return results


def resize_on_device(a, h2, w2, bilinear):
    """One array of a dataset sample -> (h2, w2) through the HIP resampler (st_resize_planes, csrc/pack_pool.hip): uint8
    images by OpenCV's 8-bit INTER_LINEAR, everything else (uint16 PNG codes, fp32 maps, uint8 masks) by INTER_NEAREST.
    `a`: numpy (h, w) / (h, w, c) or a CUDA tensor of that layout; the result comes back in the same container.  There is
    no CPU implementation in the product: without the HIP library and a GPU this raises."""
    import torch
    from ._lib import check, current_stream, load, ptr
    if not torch.cuda.is_available():
        raise RuntimeError('Resize_Disparity with a non-identity scale resamples on the GPU (st_resize_planes); no GPU here')
    if isinstance(a, torch.Tensor):
        t = a
    elif a.dtype == np.uint16:         # PNG codes: moved as their int16 bit pattern (nearest sampling copies elements)
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.int16))
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    if bilinear and t.dtype != torch.uint8:
        raise TypeError('the bilinear path is the 8-bit image path (decoded frames are uint8)')
    h, w = t.shape[:2]
    P = 1 if t.dim() == 2 else int(t.shape[2])
    dev = t.device if t.is_cuda else torch.device('cuda', torch.cuda.current_device())
    src = t.to(dev).contiguous()
    dst = torch.empty((h2, w2) + tuple(t.shape[2:]), dtype=t.dtype, device=dev)
    check(load().st_resize_planes(ptr(src), P, h, w, 1, ptr(dst), h2, w2, t.element_size(), 1 if bilinear else 0,
                                  current_stream()), 'st_resize_planes')
    if isinstance(a, torch.Tensor):
        return dst
    out = dst.cpu().numpy()
    return out.view(a.dtype) if out.dtype != a.dtype else out


@TRANSFORMS.register_module(name=['Resize_Disparity', 'mmtrack.Resize_Disparity'])
class Resize_Disparity:
    """transforms_disparity.py:23-137 at test time: rescale to `scale` (w, h) - keep_ratio: mmcv.rescale_size, else exactly
    `scale`.  The shipped pipeline rescales 1280x720 frames to (1280, 720): the identity, nothing is touched.  Any other
    size resamples ON THE DEVICE (st_resize_planes): img / right by cv2's 8-bit INTER_LINEAR, disp_postp / disp_codes /
    disp_mask / depth_postp by INTER_NEAREST (:52-112), and records img_shape / scale_factor = (new_w / w, new_h / h) the
    way mmdet's Resize does - predict() divides the boxes by it again (rescale=True)."""

    BILINEAR = ('img', 'right')
    NEAREST = ('disp_postp', 'disp_codes', 'disp_mask', 'depth_postp', 'depth')

    def __init__(self, scale, keep_ratio=True, **kwargs):
        self.scale, self.keep_ratio = tuple(scale), bool(keep_ratio)

    def new_size(self, h, w):
        if self.keep_ratio:
            f = min(max(self.scale) / max(h, w), min(self.scale) / min(h, w))      # mmcv.rescale_size
            return int(w * f + 0.5), int(h * f + 0.5)
        return int(self.scale[0]), int(self.scale[1])

    def __call__(self, results):
        h, w = results['img_shape'][:2]
        nw, nh = self.new_size(h, w)
        results['scale'] = self.scale
        results['keep_ratio'] = self.keep_ratio
        if (nw, nh) == (w, h):
            results['scale_factor'] = (1.0, 1.0)
            return results
        done = {}
        for keys, bil in ((self.BILINEAR, True), (self.NEAREST, False)):
            for k in keys:
                a = results.get(k)
                if a is None:
                    continue
                if id(a) not in done:      # 'depth' and 'depth_postp' may be one array
                    done[id(a)] = resize_on_device(a, nh, nw, bil)
                results[k] = done[id(a)]
        results['img_shape'] = (nh, nw) + tuple(results['img_shape'][2:])
        results['scale_factor'] = (nw / w, nh / h)
        return results


@TRANSFORMS.register_module(name=['Pad_Disparity', 'mmtrack.Pad_Disparity'])
class Pad_Disparity:
    """transforms_disparity.py:140-249: right / bottom padding to a multiple of size_divisor (img 114, disp 0, mask 0).
    On the device path the padding is applied by st_pack_raw_inputs; here only the padded shape is recorded."""

    def __init__(self, size_divisor=32, pad_val=None, **kwargs):
        self.size_divisor, self.pad_val = size_divisor, pad_val or dict(img=114.0, disp=0, disp_mask=0)

    def __call__(self, results):
        h, w = results['img_shape'][:2]
        d = self.size_divisor
        results['pad_shape'] = ((h + d - 1) // d * d, (w + d - 1) // d * d)
        results['pad_size_divisor'] = d
        return results


@TRANSFORMS.register_module(name=['PackTrackInputs_Disparity', 'mmtrack.PackTrackInputs_Disparity'])
class PackTrackInputs_Disparity:
    """formatting_disparity.py:139-338 at test time (pack_single_img=True): arrays HWC -> tensors (T=1,C,H,W) under
    'inputs', metainfo + GT instances on a TrackDataSample.  RAW bytes are kept raw (uint8 image, uint16 codes as
    int16 storage): the cast / pad / x3 repeat is the device pre-processor's job (st_pack_raw_inputs)."""

    def __init__(self, pack_single_img=True, meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor'),
                 default_meta_keys=('frame_id', 'video_length', 'instances'), **kwargs):
        self.meta_keys = tuple(meta_keys) + tuple(k for k in default_meta_keys if k not in meta_keys)

    def __call__(self, results):
        import torch
        from .structures import TrackDataSample
        inputs = {}
        for key, out in (('img', 'img'), ('right', 'right'), ('disp_postp', 'disp_postp'), ('disp_mask', 'disp_mask'),
                         ('depth_postp', 'depth_postp')):
            if key in results:
                a = results[key]
                a = a[:, :, None] if a.ndim == 2 else a
                inputs[out] = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))[None]
        if 'disp_codes' in results:
            inputs['disp_codes'] = torch.from_numpy(results['disp_codes'].view(np.int16))[None]
        sample = TrackDataSample({k: results[k] for k in self.meta_keys if k in results})
        return dict(inputs=inputs, data_samples=sample)


# ---- whole videos as page-locked raw sequences (the configs[4] driver's input) --------------------------------------
def load_video(dataset, indices, use_right, with_depth=False, pin=True):
    """The frames of ONE video -> (HostSequence of raw bytes, per-frame GT instance lists, per-frame metainfo[, depth
    maps (T,h,w) float32 metres])."""
    from .sequence import HostSequence
    left, second, gts, metas, depth = [], [], [], [], []
    load_img = LoadImageFromFile()
    load_right = LoadImageFromFile(key='right_path', out='right')
    load_disp = LoadDisparityFromFile(to_float32=False, raw_codes=True)
    load_depth = LoadDepthFromFile()
    for i in indices:
        info = dataset.get_data_info(i)
        r = load_img(dict(info))
        left.append(np.ascontiguousarray(r['img'].transpose(2, 0, 1)))
        if use_right:
            second.append(np.ascontiguousarray(load_right(dict(info))['right'].transpose(2, 0, 1)))
        else:
            second.append(load_disp(dict(info))['disp_codes'])
        if with_depth and 'depth_path' in info:
            depth.append(load_depth(dict(info))['depth'][..., 0])
        gts.append(info['instances'])
        metas.append({k: info[k] for k in ('frame_id', 'video_length', 'img_path', 'img_id', 'height', 'width') if k in info})
    seq = HostSequence.from_raw(np.stack(left), right=np.stack(second) if use_right else None,
                                codes=None if use_right else np.stack(second), gt=gts, pin=pin)
    out = (seq, gts, metas)
    return out + (np.stack(depth),) if depth else out


def load_videos(dataset, use_right, sampler=None):
    """dict video name -> HostSequence and dict name -> GT lists for this rank's videos (VideoSampler split)."""
    sampler = sampler or VideoSampler(dataset)
    videos, gts = {}, {}
    for name, idx in sampler.videos:
        seq, gt, _ = load_video(dataset, idx, use_right)[:3]
        videos[name], gts[name] = seq, gt
    return videos, gts
