"""Typed configuration from the reference's environment-variable contract.

The reference has no argparse and no config files: module-level ``os.getenv`` reads are the
whole flag system (``imagenet_pytorch_horovod.py:35-53``).  This module keeps the
same variable names but parses them with types, fixing the reference's quirks instead of
replicating them: ``EPOCHS`` is an int (#1), booleans accept only real truthy
spellings rather than "any string containing t" (#2), validation data comes from the TEST
directory (#3).

Engine tunables use ``HVD_*`` names and also honour Horovod's ``HOROVOD_*`` spellings
(SURVEY §5.6).
"""
from __future__ import annotations

import argparse
import os
from dataclasses import dataclass, fields
from typing import Optional

__all__ = ["str_to_bool", "env_int", "env_float", "env_bool", "env_str", "TrainConfig",
           "EngineConfig", "IMAGENET_TRAIN_LEN", "RGB_MEAN", "RGB_SD"]

IMAGENET_TRAIN_LEN = 1281167  # reference default FAKE_DATA_LENGTH (:50-52)
RGB_MEAN = (0.485, 0.456, 0.406)  # reference :41
RGB_SD = (0.229, 0.224, 0.225)  # reference :42

_TRUE = {"1", "true", "t", "yes", "y", "on"}
_FALSE = {"0", "false", "f", "no", "n", "off", ""}


def str_to_bool(value) -> bool:
    """Strict boolean parsing (the reference's ``_str_to_bool`` :28-32 is a substring test)."""
    if isinstance(value, bool):
        return value
    s = str(value).strip().lower()
    if s in _TRUE:
        return True
    if s in _FALSE:
        return False
    raise ValueError("cannot interpret %r as a boolean" % (value,))


def env_str(name, default=None, *aliases):
    for key in (name,) + aliases:
        if key in os.environ:
            return os.environ[key]
    return default


def env_int(name, default, *aliases) -> int:
    raw = env_str(name, None, *aliases)
    return int(default) if raw is None or raw == "" else int(float(raw))


def env_float(name, default, *aliases) -> float:
    raw = env_str(name, None, *aliases)
    return float(default) if raw is None or raw == "" else float(raw)


def env_bool(name, default=False, *aliases) -> bool:
    raw = env_str(name, None, *aliases)
    return bool(default) if raw is None else str_to_bool(raw)


@dataclass
class TrainConfig:
    """Training-script settings; defaults are the reference's constants (:35-53)."""

    width: int = 224
    height: int = 224
    channels: int = 3
    num_classes: int = 1000
    lr: float = 0.001  # scaled by world size at optimizer construction (:333)
    momentum: float = 0.9
    weight_decay: float = 0.0  # reference defines 5e-5 (:47) but never applies it in PyTorch
    warmup_epochs: int = 0  # reference defines 5 (:46) but only Keras uses it
    epochs: int = 1
    batch_size: int = 64  # per GPU (:40)
    seed: int = 42
    distributed: bool = False
    fake: bool = False  # reference default (imagenet_pytorch_horovod.py:49)
    fake_data_length: int = IMAGENET_TRAIN_LEN
    validation: bool = False
    train_dir: Optional[str] = None
    test_dir: Optional[str] = None
    output_dir: Optional[str] = None  # rank-0 checkpoint directory
    temp_dir: Optional[str] = None  # other ranks' scratch directory
    num_workers: int = 5
    log_interval: int = 100
    model: str = "resnet50"
    dtype: str = "bf16"
    optimizer: str = "sgd"  # sgd | lars
    device_data: bool = True  # on-GPU synthetic generator instead of the host FakeData loader
    prefetch_depth: int = 2  # host batches uploaded + converted ahead on a copy stream (0 = off)
    uint8_transport: bool = True  # real data: ship uint8 pixels, normalise on the GPU (K20)
    hip_graph: str = "auto"  # whole step as one HIP graph: on | off | auto (one GPU, batch <= 32)
    resume: bool = True
    metrics_path: Optional[str] = None
    max_steps: int = 0  # 0 = whole epoch

    @classmethod
    def from_env(cls, **overrides) -> "TrainConfig":
        cfg = cls(
            epochs=env_int("EPOCHS", 1),
            batch_size=env_int("BATCHSIZE", 64, "BATCH_SIZE"),
            lr=env_float("LR", 0.001),
            momentum=env_float("MOMENTUM", 0.9),
            weight_decay=env_float("WEIGHT_DECAY", 0.0),
            warmup_epochs=env_int("WARMUP_EPOCHS", 0),
            distributed=env_bool("DISTRIBUTED", False),
            fake=env_bool("FAKE", False),
            fake_data_length=env_int("FAKE_DATA_LENGTH", IMAGENET_TRAIN_LEN),
            validation=env_bool("VALIDATION", False),
            train_dir=env_str("AZ_BATCHAI_INPUT_TRAIN", None, "INPUT_TRAIN_DIR"),
            test_dir=env_str("AZ_BATCHAI_INPUT_TEST", None, "INPUT_TEST_DIR"),
            output_dir=env_str("AZ_BATCHAI_OUTPUT_MODEL", None, "OUTPUT_MODEL_DIR"),
            temp_dir=env_str("AZ_BATCHAI_JOB_TEMP_DIR", None, "JOB_TEMP_DIR"),
            num_workers=env_int("NUM_WORKERS", 5),
            model=env_str("MODEL", "resnet50"),
            dtype=env_str("DTYPE", "bf16"),
            optimizer=env_str("OPTIMIZER", "sgd"),
            device_data=env_bool("DEVICE_DATA", True),
            prefetch_depth=env_int("PREFETCH_DEPTH", 2),
            uint8_transport=env_bool("UINT8_TRANSPORT", True),
            hip_graph=env_str("HIP_GRAPH", "auto"),
            resume=env_bool("RESUME", True),
            metrics_path=env_str("METRICS_PATH", None),
            max_steps=env_int("MAX_STEPS", 0),
            log_interval=env_int("LOG_INTERVAL", 100),
        )
        for k, v in overrides.items():
            if v is not None:
                setattr(cfg, k, v)
        return cfg

    @classmethod
    def add_arguments(cls, parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
        """One ``--flag`` per field; unset flags fall back to the environment value."""
        for f in fields(cls):
            flag = "--" + f.name.replace("_", "-")
            if f.type in ("bool", bool):
                parser.add_argument(flag, type=str_to_bool, default=None, metavar="BOOL")
            elif f.type in ("int", int):
                parser.add_argument(flag, type=int, default=None)
            elif f.type in ("float", float):
                parser.add_argument(flag, type=float, default=None)
            else:
                parser.add_argument(flag, type=str, default=None)
        return parser

    @classmethod
    def from_args(cls, argv=None) -> "TrainConfig":
        parser = cls.add_arguments(argparse.ArgumentParser(description=cls.__doc__))
        ns = parser.parse_args(argv)
        return cls.from_env(**vars(ns))


@dataclass
class EngineConfig:
    """Gradient-exchange engine tunables."""

    # bytes per bucket cap.  Measured with a one-rank RCCL communicator (HVD_FORCE_COLLECTIVES=1,
    # profiles/r1_notes.md): every extra bucket costs ~0.065 ms per step (collective launch on the
    # host while the late layers' short kernels wait for it), so 32 MB (3 bf16 buckets + 1 fp32 for
    # ResNet-50; the un-overlappable tail is the last 11.5 MB) beats 8 MB (7 + 1, 6 MB tail)
    fusion_threshold: int = 32 * 1024 * 1024
    first_bucket_bytes: int = 8 * 1024 * 1024  # small first bucket: comm starts early in backward
    # the last bucket (first layers, ready only when backward ends) holds at most this much, so the
    # bucket before it is exchanged while the stem's backward still runs
    tail_bucket_bytes: int = 256 * 1024
    # dtype of the gradient arenas = dtype on the wire and of multi-pass accumulation:
    # auto / bf16 = the parameter's dtype (bf16 conv/fc gradients, 51 MB per ResNet-50 step),
    # fp32 = fp32 arenas for every parameter (102 MB; the split-K reduce writes fp32 directly)
    comm_dtype: str = "auto"
    backend: str = "auto"  # auto | rccl | gloo
    # auto = native on GPUs (in-tree C++ engine: RCCL called directly from its comm thread),
    # torch.distributed's gloo on CPU | torch (c10d process group) | native
    engine: str = "auto"
    # rccl | twoshot | oneshot | auto: which all-reduce serves the gradient buckets (native engine
    # on GPUs; twoshot/oneshot = the direct xGMI peer-access kernel, auto = by bucket size)
    allreduce_algo: str = "rccl"
    num_buckets: int = 0  # > 0: split every arena into this many equal buckets instead of the byte caps
    timeline: Optional[str] = None
    debug: bool = False
    collective_timeout_s: float = 300.0
    stall_warning_s: float = 60.0  # Horovod's stall-check default [EXT]
    fault_inject: Optional[str] = None  # "rank:step"
    # Horovod's negotiation cycle.  Static buckets need none: a non-zero value is accepted for
    # command-line compatibility and reported once at init as having no effect.
    cycle_time_ms: float = 0.0
    # run the gradient exchange even with one rank (rehearses the RCCL path on a one-GPU box)
    force_collectives: bool = False
    high_priority_comm: bool = True   # RCCL collectives on a high-priority HIP stream (own hardware queue)
    comm_metrics: bool = True         # per-step exposed-communication time from HIP events (never synchronises)
    per_bucket_update: bool = True    # FusedSGD: update bucket k as soon as the stream has waited for bucket k
    # one checked all-reduce per enabled algorithm / wire dtype + broadcast + all-gather at init()
    # (parallel/selftest.py); a wrong word or a timeout ends the job with the engine's diagnosis
    init_selftest: bool = True
    # "" = a native engine that fails to start or fails its self-test ends the job (default);
    # "torch" = every rank then agrees (through the TCP store) to continue on torch.distributed's
    # RCCL process group and says so on stderr and in hvd.engine_name()
    engine_fallback: str = ""
    # after each of the first N optimizer steps (every step with HVD_DEBUG=1) the ranks all-gather a
    # checksum of their reduced gradients and updated parameters and raise if replicas differ
    replica_check_steps: int = 2
    fault_corrupt: Optional[str] = None  # "rank:step": flip one reduced-gradient word there (tests the check)

    @classmethod
    def from_env(cls) -> "EngineConfig":
        return cls(
            fusion_threshold=env_int("HVD_FUSION_THRESHOLD", 32 * 1024 * 1024,
                                     "HOROVOD_FUSION_THRESHOLD"),
            first_bucket_bytes=env_int("HVD_FIRST_BUCKET_BYTES", 8 * 1024 * 1024),
            tail_bucket_bytes=env_int("HVD_TAIL_BUCKET_BYTES", 256 * 1024),
            comm_dtype=env_str("HVD_COMM_DTYPE", "auto"),
            backend=env_str("HVD_BACKEND", "auto"),
            engine=env_str("HVD_ENGINE", "auto"),
            allreduce_algo=env_str("HVD_ALLREDUCE_ALGO", "rccl"),
            num_buckets=env_int("HVD_NUM_BUCKETS", 0),
            timeline=env_str("HVD_TIMELINE", None, "HOROVOD_TIMELINE"),
            debug=env_bool("HVD_DEBUG", False),
            collective_timeout_s=env_float("HVD_COLLECTIVE_TIMEOUT_S", 300.0),
            stall_warning_s=env_float("HVD_STALL_WARNING_S", 60.0,
                                      "HOROVOD_STALL_CHECK_TIME_SECONDS"),
            fault_inject=env_str("HVD_FAULT_INJECT", None),
            cycle_time_ms=env_float("HVD_CYCLE_TIME", 0.0, "HOROVOD_CYCLE_TIME"),
            force_collectives=env_bool("HVD_FORCE_COLLECTIVES", False),
            high_priority_comm=env_bool("HVD_HIGH_PRIORITY_COMM", True),
            comm_metrics=env_bool("HVD_COMM_METRICS", True),
            per_bucket_update=env_bool("HVD_PER_BUCKET_UPDATE", True),
            init_selftest=env_bool("HVD_INIT_SELFTEST", True),
            engine_fallback=(env_str("HVD_ENGINE_FALLBACK", "") or "").lower(),
            replica_check_steps=env_int("HVD_REPLICA_CHECK_STEPS", 2),
            fault_corrupt=env_str("HVD_FAULT_CORRUPT", None),
        )
