from . import backend, collectives, functions, bucket, compression, timeline
from .optimizer import DistributedOptimizer
