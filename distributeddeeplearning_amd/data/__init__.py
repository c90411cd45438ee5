from .fake import FakeData, FakeBatchIterator
from .sampler import ShardedSampler, EpochRandomSampler, get_sampler
from .synthetic import DeviceSyntheticLoader
