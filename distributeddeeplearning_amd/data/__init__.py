from .fake import FakeData, FakeBatchIterator, PrebatchedFakeStream
from .sampler import ShardedSampler, EpochRandomSampler, get_sampler
from .synthetic import DeviceSyntheticLoader
from .prefetch import DevicePrefetcher
