from .loop import (train_epoch, validate, log_summary, to_device_batch, build_data, build_optimizer,
                   build_graph_step)
from .trainer import Trainer
from .estimator import Estimator, EstimatorSpec, ModeKeys, RunConfig, BroadcastGlobalVariablesHook
