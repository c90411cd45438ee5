"""distributeddeeplearning_amd — MI355X-native synchronous data-parallel CNN training.

``import distributeddeeplearning_amd.torch as hvd`` gives the Horovod-compatible API; see
README.md and SURVEY.md for the layer map.
"""
__version__ = "0.1.0"
