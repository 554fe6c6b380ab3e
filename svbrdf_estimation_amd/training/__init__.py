"""what sits either side of the hot path in a training run -- the
Deschaintre U-Net (stock PyTorch-ROCm modules, no custom kernels), the tiled-PNG sample reader,
and a one-process-per-GPU DDP harness (RCCL all-reduce of the U-Net gradients; the rendering
loss itself shards by batch with no collective).  See train.py at the repository root."""
from . import data, models  # noqa: F401
