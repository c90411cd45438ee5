"""distributeddeeplearning_amd — MI355X-native synchronous data-parallel CNN training.

``import distributeddeeplearning_amd.torch as hvd`` gives the Horovod-compatible API; see
README.md and SURVEY.md for the layer map.
"""
__version__ = "0.1.0"

import os as _os

# Multi-process GPU work on this ROCm stack (RCCL peer buffers, the direct xGMI all-reduce's IPC
# handles) needs the dmabuf IPC mode; it has to be chosen before the HIP runtime starts, i.e.
# before the first device call of the process.  An explicit setting in the environment wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
