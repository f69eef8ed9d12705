"""MI355X-native dense hot path of the stereo tracker."""
import os

# One hardware queue per HIP stream: ROCm maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default
# 4), and two in-flight contexts that share a queue serialise behind each other (4 contexts: 1729 pairs/s on 4 queues,
# 1840 on 8).  The HIP runtime reads it when it initialises, i.e. this import has to come before the first CUDA call
# of the process; an explicit setting in the environment wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
# Kernel arguments in device memory instead of host-coherent memory: ~2 us less per launch (1538 -> 1568 pairs/s with one
# context, +0.3 % with four); same rule - read at runtime initialisation, an explicit setting wins.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
