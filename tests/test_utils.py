"""L0 utilities: Timer, logger, typed config, dotenv helpers (SURVEY §4 unit list)."""
import io
import logging
import os

import pytest

from distributeddeeplearning_amd.utils import env as envmod
from distributeddeeplearning_amd.utils.config import (EngineConfig, TrainConfig, env_int,
                                                      str_to_bool)
from distributeddeeplearning_amd.utils.logging import LOG_FORMAT, MetricsWriter, RankAdapter
from distributeddeeplearning_amd.utils.timer import Timer, timer


class FakeClock:
    def __init__(self):
        self.t = 0.0

    def __call__(self):
        return self.t


def test_timer_live_then_frozen():
    clk = FakeClock()
    t = Timer(timer=clk)
    t.start()
    clk.t = 2.0
    assert t.elapsed == 2.0
    clk.t = 3.5
    t.stop()
    clk.t = 10.0
    assert t.elapsed == 3.5
    assert str(t) == "3.500"


def test_timer_factor_and_context_output():
    clk = FakeClock()
    out = []
    with Timer(timer=clk, factor=1000, output=out.append, prefix="Training",
               fmt="took {:.1f} ms") as t:
        clk.t = 0.25
        assert t.elapsed == 250.0
    assert out == ["Training took 250.0 ms"]


def test_timer_prefix_implies_print(capsys):
    with Timer(timer=FakeClock(), prefix="X"):
        pass
    assert capsys.readouterr().out.strip() == "X took 0.000 seconds"


def test_timer_silent_without_output(capsys):
    with Timer(timer=FakeClock()):
        pass
    assert capsys.readouterr().out == ""


def test_timer_restart_resets_end():
    clk = FakeClock()
    t = Timer(timer=clk)
    t.start(); clk.t = 1.0; t.stop()
    t.start(); clk.t = 1.5
    assert t.elapsed == 0.5


def test_timer_decorator_bare_and_with_logger(capsys):
    @timer
    def f(x):
        return x + 1

    assert f(1) == 2
    assert "function f execution time" in capsys.readouterr().out
    records = []

    class H(logging.Handler):
        def emit(self, r):
            records.append(r)

    lg = logging.getLogger("timer-test")
    lg.setLevel(logging.DEBUG)
    lg.addHandler(H())

    @timer(logger=lg, level=logging.DEBUG)
    def g():
        return 7

    assert g() == 7
    assert records and records[0].function_name == "g" and records[0].execution_time >= 0


def test_logger_format_matches_reference():
    stream = io.StringIO()
    base = logging.getLogger("fmt-test")
    base.setLevel(logging.INFO)
    h = logging.StreamHandler(stream)
    h.setFormatter(logging.Formatter(LOG_FORMAT))
    base.addHandler(h)
    base.propagate = False
    lg = RankAdapter(base)
    lg.info("hello")
    lg.set_epoch(3)
    lg.info("again")
    lines = stream.getvalue().splitlines()
    assert lines[0] == "INFO:fmt-test:0:  hello"
    assert lines[1] == "INFO:fmt-test:0: [Epoch 3] again"


def test_str_to_bool_is_strict():
    assert str_to_bool("True") and str_to_bool("1") and str_to_bool("yes")
    assert not str_to_bool("False") and not str_to_bool("0") and not str_to_bool("")
    # the reference's substring test calls these True
    for bad in ("not", "testing"):
        with pytest.raises(ValueError):
            str_to_bool(bad)


def test_config_from_env_types(monkeypatch):
    monkeypatch.setenv("EPOCHS", "3")
    monkeypatch.setenv("DISTRIBUTED", "True")
    monkeypatch.setenv("FAKE", "False")
    monkeypatch.setenv("FAKE_DATA_LENGTH", "1000")
    monkeypatch.setenv("AZ_BATCHAI_OUTPUT_MODEL", "/tmp/out")
    cfg = TrainConfig.from_env()
    assert cfg.epochs == 3 and isinstance(cfg.epochs, int)
    assert cfg.distributed is True and cfg.fake is False
    assert cfg.fake_data_length == 1000 and cfg.output_dir == "/tmp/out"
    assert list(range(cfg.epochs)) == [0, 1, 2]  # reference raises here (§2.11 #1)
    assert cfg.batch_size == 64 and cfg.lr == 0.001 and cfg.momentum == 0.9


def test_config_args_override_env(monkeypatch):
    monkeypatch.setenv("EPOCHS", "3")
    cfg = TrainConfig.from_args(["--epochs", "5", "--fake", "true", "--model", "resnet18"])
    assert cfg.epochs == 5 and cfg.fake is True and cfg.model == "resnet18"


def test_engine_config_honours_horovod_names(monkeypatch):
    monkeypatch.setenv("HOROVOD_FUSION_THRESHOLD", str(16 << 20))
    monkeypatch.setenv("HOROVOD_TIMELINE", "/tmp/tl.json")
    cfg = EngineConfig.from_env()
    assert cfg.fusion_threshold == 16 << 20 and cfg.timeline == "/tmp/tl.json"
    monkeypatch.setenv("HVD_FUSION_THRESHOLD", str(4 << 20))
    assert EngineConfig.from_env().fusion_threshold == 4 << 20
    assert env_int("NOPE_NOT_SET", 7) == 7


def test_dotenv_roundtrip(tmp_path):
    path = envmod.dotenv_for(str(tmp_path))
    assert os.path.isfile(path)
    envmod.set_key(path, "GROUP_NAME", "rg1")
    envmod.set_key(path, "NUM_NODES", 2)
    assert envmod.get_key(path, "GROUP_NAME") == "rg1"
    assert envmod.get_key(path, "NUM_NODES") == "2"
    sub = tmp_path / "a" / "b"
    sub.mkdir(parents=True)
    assert envmod.dotenv_for(str(sub)) == path  # found by walking up
    envmod.set_key(path, "PASSWORD", "s3cret")
    assert envmod.get_password(path) == "s3cret"
    envmod.write_json_to_file({"a": 1}, str(tmp_path / "job.json"))
    assert (tmp_path / "job.json").read_text().strip().startswith("{")


def test_metrics_writer(tmp_path):
    import json

    p = tmp_path / "m" / "metrics.jsonl"
    with MetricsWriter(str(p), rank=0) as w:
        w.write(step=1, loss=2.5)
    rec = json.loads(p.read_text().splitlines()[0])
    assert rec["step"] == 1 and rec["loss"] == 2.5 and rec["rank"] == 0
    w2 = MetricsWriter(str(tmp_path / "n.jsonl"), rank=1)
    w2.write(step=1)
    assert not (tmp_path / "n.jsonl").exists()


def test_numa_affinity_helpers(tmp_path, monkeypatch):
    """utils/affinity.py: cpulist parsing, sysfs lookups against a fake tree, and the binding
    rules (never widen, never below min_cpus, HVD_NUMA_BIND=0 leaves the mask alone)."""
    import os
    from distributeddeeplearning_amd.utils import affinity as A

    assert A.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert A.parse_cpulist("") == set()
    bdf = "0000:75:00.0"
    dev = tmp_path / "bus" / "pci" / "devices" / bdf
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    allowed = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text(",".join(str(c) for c in allowed[: max(2, len(allowed) // 2)]))
    assert A.gpu_numa_node(bdf, str(tmp_path)) == 1
    assert A.gpu_numa_node("0000:00:00.0", str(tmp_path)) is None
    assert A.numa_cpus(1, str(tmp_path)) == set(allowed[: max(2, len(allowed) // 2)])
    assert A.numa_cpus(7, str(tmp_path)) == set()

    calls = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: calls.append(set(cpus)))
    monkeypatch.setenv("HVD_NUMA_BIND", "0")
    assert A.bind_to_gpu_numa(0, str(tmp_path), bdf=bdf) is None and not calls
    monkeypatch.setenv("HVD_NUMA_BIND", "1")
    got = A.bind_to_gpu_numa(0, str(tmp_path), bdf=bdf, min_cpus=2)
    if len(allowed) >= 4:
        assert got == set(allowed[: len(allowed) // 2]) and calls == [got]
    if len(allowed) // 2 < 8:  # the default refuses to squeeze a rank's threads onto < 8 CPUs
        assert A.bind_to_gpu_numa(0, str(tmp_path), bdf=bdf) is None
    # unknown node (-1) and a node with no allowed CPU leave the mask alone
    (dev / "numa_node").write_text("-1\n")
    assert A.bind_to_gpu_numa(0, str(tmp_path), bdf=bdf) is None
    (dev / "numa_node").write_text("1\n")
    (node / "cpulist").write_text("100000-100003")
    assert A.bind_to_gpu_numa(0, str(tmp_path), bdf=bdf) is None


def test_launch_stream_override_is_scoped_and_per_thread():
    """kernels.launch_on redirects this thread's launchers to a raw stream and restores the previous
    target on exit, nested use included; other threads keep their own."""
    import threading

    from distributeddeeplearning_amd.ops import kernels as K

    assert K._STREAM_OVERRIDE.raw is None
    seen = {}
    with K.launch_on(11):
        assert K._raw_stream() == 11
        with K.launch_on(22):
            assert K._raw_stream() == 22
        assert K._raw_stream() == 11
        t = threading.Thread(target=lambda: seen.setdefault("other", K._STREAM_OVERRIDE.raw))
        t.start()
        t.join()
    assert seen["other"] is None
    assert K._STREAM_OVERRIDE.raw is None
    try:
        with K.launch_on(33):
            raise ValueError("boom")
    except ValueError:
        pass
    assert K._STREAM_OVERRIDE.raw is None
