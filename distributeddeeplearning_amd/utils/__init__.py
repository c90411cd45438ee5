from .timer import Timer, timer, DeviceTimer
from .logging import get_logger, RankAdapter, MetricsWriter
from .config import TrainConfig, EngineConfig, str_to_bool
