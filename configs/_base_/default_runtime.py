# Runtime defaults for the inference entry points of this repo (no training hooks: out of scope).
default_scope = 'mmtrack'
env_cfg = dict(dist_cfg=dict(backend='nccl'))  # 'nccl' is RCCL on ROCm
log_level = 'INFO'
load_from = None
