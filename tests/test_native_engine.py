"""In-process tests of the C++ engine pieces: bucket planner twin and size-1 engine plumbing."""
import ctypes
import socket

import pytest
import torch

from distributeddeeplearning_amd.ops import native
from distributeddeeplearning_amd.parallel.bucket import (BUCKET_ALIGN_BYTES, SEG_ALIGN_BYTES,
                                                         plan_arenas)

lib = native.engine_lib()
pytestmark = pytest.mark.skipif(lib is None, reason="engine library not built")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("tail", [0, 256 << 10])
def test_native_bucket_plan_matches_python_planner(tail):
    from distributeddeeplearning_amd.models import resnet50

    m = resnet50()
    entries = [(n, p.shape, torch.float32) for n, p in m.named_parameters()]
    plan = plan_arenas(entries, 8 << 20, 32 << 20, tail_bucket_bytes=tail)[torch.float32]
    numel = [s.numel for s in plan.segments]  # planning (reverse) order
    n = len(numel)
    arr = (ctypes.c_longlong * n)(*numel)
    offsets = (ctypes.c_longlong * n)()
    bucket = (ctypes.c_int * n)()
    bucket_end = (ctypes.c_longlong * n)()
    arena = ctypes.c_longlong(0)
    nb = lib.ddl_bucket_plan(arr, n, 4, ctypes.c_longlong(8 << 20), ctypes.c_longlong(32 << 20),
                             ctypes.c_longlong(tail), SEG_ALIGN_BYTES, BUCKET_ALIGN_BYTES, offsets,
                             bucket, bucket_end, ctypes.byref(arena))
    assert nb == len(plan.buckets) and arena.value == plan.numel
    assert list(offsets) == [s.offset for s in plan.segments]
    assert [bucket_end[i] for i in range(nb)] == [b.end for b in plan.buckets]
    seg_bucket = {}
    for b in plan.buckets:
        for s in b.segments:
            seg_bucket[s.name] = b.index
    assert list(bucket) == [seg_bucket[s.name] for s in plan.segments]


def test_engine_single_rank_store_and_handles(tmp_path):
    lib.ddl_eng_last_error.restype = ctypes.c_char_p
    lib.ddl_eng_store_data.restype = ctypes.c_void_p
    for fn in ("ddl_eng_allreduce_async", "ddl_eng_allgather_async", "ddl_eng_store_get",
               "ddl_eng_store_add"):
        getattr(lib, fn).restype = ctypes.c_longlong
    port = _free_port()
    assert lib.ddl_eng_init(0, 1, b"127.0.0.1", port, ctypes.c_double(10.0), ctypes.c_double(5.0)) == 0
    try:
        assert lib.ddl_eng_rank() == 0 and lib.ddl_eng_size() == 1
        assert lib.ddl_eng_timeline_start(str(tmp_path / "t.json").encode()) == 0
        t = torch.arange(8, dtype=torch.float32)
        h = lib.ddl_eng_allreduce_async(ctypes.c_void_p(t.data_ptr()), ctypes.c_longlong(8), 0, 1, b"x")
        assert lib.ddl_eng_wait(ctypes.c_longlong(h), ctypes.c_double(5.0)) == 0
        assert t.tolist() == list(range(8))
        assert lib.ddl_eng_wait(ctypes.c_longlong(h), ctypes.c_double(0.1)) == -1  # consumed
        assert lib.ddl_eng_store_set(b"k", b"hello", ctypes.c_longlong(5)) == 0
        n = lib.ddl_eng_store_get(b"k", 1000)
        assert ctypes.string_at(lib.ddl_eng_store_data(), n) == b"hello"
        assert lib.ddl_eng_store_get(b"missing", 100) == -1  # blocking get times out
        assert lib.ddl_eng_store_add(b"ctr", ctypes.c_longlong(3)) == 3
        assert lib.ddl_eng_store_add(b"ctr", ctypes.c_longlong(4)) == 7
        lib.ddl_eng_timeline_stop()
        assert "ENQUEUE" in (tmp_path / "t.json").read_text()
    finally:
        lib.ddl_eng_shutdown()


@pytest.mark.gpu
def test_engine_device_plane_single_gpu_event_plumbing():
    """The engine's device plane on one MI355X (communicator of size 1): the comm thread issues
    ncclAllReduce / ncclBroadcast on the plane's own stream; ordering with the torch stream is by
    events only (no host synchronisation), handles poll and wait like host handles."""
    lib.ddl_eng_device_collective.restype = ctypes.c_longlong
    lib.ddl_eng_device_collective.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_char_p]
    lib.ddl_eng_device_wait.argtypes = [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_double]
    lib.ddl_eng_device_wait_host.argtypes = [ctypes.c_longlong, ctypes.c_double]
    lib.ddl_rccl_last_error.restype = ctypes.c_char_p
    lib.ddl_eng_last_error.restype = ctypes.c_char_p
    lib.ddl_rccl_stream.restype = ctypes.c_void_p
    lib.ddl_rccl_launch_count.restype = ctypes.c_longlong
    port = _free_port()
    assert lib.ddl_eng_init(0, 1, b"127.0.0.1", port, ctypes.c_double(30.0), ctypes.c_double(10.0)) == 0
    try:
        assert lib.ddl_rccl_init(0, 1, torch.cuda.current_device(), 30000) == 0, lib.ddl_rccl_last_error()
        x = torch.randn(1 << 20, device="cuda").to(torch.bfloat16)
        ref = (x.float() * 2).to(torch.bfloat16)
        y = x * 2  # producer kernel on the torch stream; the collective must wait for it
        s = torch.cuda.current_stream().cuda_stream
        assert lib.ddl_rccl_stream() not in (None, s)
        h = lib.ddl_eng_device_collective(0, y.data_ptr(), None, y.numel(), 3, 1, 0, 0, s, b"g")
        assert h > 0, lib.ddl_eng_last_error()
        assert lib.ddl_eng_device_wait(h, s, 10.0) == 0, lib.ddl_eng_last_error()
        z = y + 0  # consumer ordered after the collective
        torch.cuda.synchronize()
        assert torch.equal(z, ref)
        h = lib.ddl_eng_device_collective(1, y.data_ptr(), None, y.numel(), 3, 0, 0, 0, s, b"b")
        assert lib.ddl_eng_device_wait_host(h, 10.0) == 0
        assert lib.ddl_eng_poll(ctypes.c_longlong(h)) == -1  # consumed
        assert lib.ddl_rccl_launch_count() == 2 and lib.ddl_eng_failed() == 0
    finally:
        lib.ddl_rccl_shutdown()
        lib.ddl_eng_shutdown()


def _stress_binary():
    """Builds csrc/engine/tests/engine_stress.cpp with -fsanitize=thread (host code only)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ddl_build", os.path.join(root, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        return mod.build_engine_stress(tsan=True, verbose=False)
    except RuntimeError as e:  # no clang++ with a TSan runtime on this machine
        pytest.skip(str(e))


@pytest.mark.parametrize("args,expect", [
    (["3", "60", "0.0005"], "PASS"),       # watchdog reporting on nearly every collective
    (["3", "30", "30", "7"], "peer loss surfaced"),  # last rank dies at round 7
])
def test_engine_stress_under_thread_sanitizer(args, expect):
    """the C++ engine built with -fsanitize=thread against its CPU data plane.
    Three ranks, mixed collectives with uneven counts, concurrent poll/store/timeline threads,
    every result word checked; the second case loses a rank mid-job and requires the loss to
    surface as failed waits on the survivors."""
    import subprocess
    exe = _stress_binary()
    proc = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "ThreadSanitizer" not in proc.stdout, proc.stdout[-4000:]
    assert proc.returncode == 0, proc.stdout[-4000:]
    assert expect in proc.stdout and "[stress] PASS" in proc.stdout


def test_unauthenticated_control_plane_objects_are_plain_data_only():
    """Without DDL_JOB_TOKEN (a torchrun launch) the store is unpickled with a restricted
    unpickler: containers / numbers / strings / dtypes pass, anything naming a callable is refused."""
    import io
    import pickle

    import torch

    from distributeddeeplearning_amd.parallel.comm import CommError, _PlainUnpickler

    obj = {"epoch": 7, "metrics": [0.5, 1.0], "state": ("__tensor__", 0, (3, 4), "torch.float32"),
           "dtype": torch.bfloat16, "shape": torch.Size([2, 3]), "b": b"xy", "none": None}
    assert _PlainUnpickler(io.BytesIO(pickle.dumps(obj))).load() == obj

    class Evil:
        def __reduce__(self):
            import os

            return (os.system, ("echo pwned",))

    with pytest.raises(CommError):
        _PlainUnpickler(io.BytesIO(pickle.dumps(Evil()))).load()
    with pytest.raises(CommError):
        _PlainUnpickler(io.BytesIO(pickle.dumps(torch.ones(2)))).load()   # tensors travel on the data plane
