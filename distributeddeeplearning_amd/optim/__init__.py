from .fused import FusedSGD, FusedLARS, default_weight_decay_filter
from .lr import scaled_lr, staircase_multiplier, goyal_lr, poly_lr, LRSchedule
