"""Deployment / data-preparation tooling (SURVEY C19-C22 counterparts)."""
import io
import json
import os
import sys
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import create_resources  # noqa: E402
import local_smoke  # noqa: E402
import prepare_imagenet  # noqa: E402
import submit_job  # noqa: E402


def _tar_bytes(files):
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w") as t:
        for name, data in files.items():
            info = tarfile.TarInfo(name)
            info.size = len(data)
            t.addfile(info, io.BytesIO(data))
    return buf.getvalue()


def test_prepare_train_nested_tars(tmp_path):
    outer = _tar_bytes({"n01.tar": _tar_bytes({"n01_1.JPEG": b"a", "n01_2.JPEG": b"b"}),
                        "n02.tar": _tar_bytes({"n02_1.JPEG": b"c"})})
    (tmp_path / "train.tar").write_bytes(outer)
    assert prepare_imagenet.prepare_train(str(tmp_path / "train.tar"), str(tmp_path / "train")) == (2, 3)
    assert sorted(os.listdir(tmp_path / "train" / "n01")) == ["n01_1.JPEG", "n01_2.JPEG"]


def test_prepare_val_from_valprep_style_script(tmp_path):
    (tmp_path / "valprep.sh").write_text(
        "#!/usr/bin/env bash\nmkdir -p n01\nmkdir -p n02\nmv v1.JPEG n02/\nmv v2.JPEG n01/\n")
    flat = tmp_path / "flat"
    flat.mkdir()
    (flat / "v1.JPEG").write_bytes(b"x")
    (flat / "v2.JPEG").write_bytes(b"y")
    mapping, classes = prepare_imagenet.parse_mapping(str(tmp_path / "valprep.sh"))
    assert mapping == {"v1.JPEG": "n02", "v2.JPEG": "n01"} and classes == ["n01", "n02"]
    assert prepare_imagenet.prepare_val(str(flat), str(tmp_path / "val"), str(tmp_path / "valprep.sh")) == (2, 2, 0)
    assert os.path.isfile(tmp_path / "val" / "n02" / "v1.JPEG")
    arch = prepare_imagenet.pack(str(tmp_path / "val"), str(tmp_path / "val.tar.gz"))
    assert tarfile.is_tarfile(arch)


def test_reference_valprep_is_parseable_as_data():
    ref = "/root/reference/valprep.sh"
    if not os.path.exists(ref):
        pytest.skip("reference tree not mounted")
    mapping, classes = prepare_imagenet.parse_mapping(ref)
    assert len(classes) == 1000 and len(mapping) == 50000


def test_tar_path_traversal_refused(tmp_path):
    (tmp_path / "evil.tar").write_bytes(_tar_bytes({"../escape.txt": b"x"}))
    with tarfile.open(tmp_path / "evil.tar") as t, pytest.raises(RuntimeError):
        prepare_imagenet._safe_extract(t, str(tmp_path / "out"))


def test_job_json_has_launch_line_and_directories(tmp_path, monkeypatch):
    monkeypatch.setenv("DDL_JOBS_DIR", str(tmp_path))
    cl = json.load(open(os.path.join(ROOT, "deploy/cluster/cluster.json")))
    job = submit_job.build_job("j1", cl, 2, 8, "examples/imagenet_pytorch.py", True, 3, {"FOO": "bar"})
    line = job["properties"]["customToolkitSettings"]["commandLine"]
    assert "-np 16" in line and "--hostfile" in line and "-x FAKE=True" in line and "-x FOO=bar" in line
    assert "-x DISTRIBUTED=True" in line and "-x EPOCHS=3" in line
    assert job["hosts"] == ["node0 slots=8", "node1 slots=8"]
    ids = [d["id"] for d in job["properties"]["inputDirectories"]]
    assert ids == ["SCRIPTS", "TRAIN", "TEST"] and job["properties"]["outputDirectories"][0]["id"] == "MODEL"
    sb = submit_job.render_sbatch(job, str(tmp_path))
    assert "#SBATCH --nodes=2" in sb and "--gpus-per-node=8" in sb
    with pytest.raises(ValueError):
        submit_job.build_job("j2", cl, 3, 8, "x.py", True, 1, {})


def test_submit_cli_dry_run(tmp_path, monkeypatch, capsys):
    monkeypatch.setenv("DDL_JOBS_DIR", str(tmp_path))
    monkeypatch.chdir(tmp_path)
    assert submit_job.main(["create", "--name", "t", "--nodes", "1", "--ppn", "2", "--fake"]) == 0
    assert os.path.isfile(tmp_path / "t" / "job.json") and (tmp_path / "t" / "hostfile").read_text() == "node0 slots=2\n"
    assert submit_job.main(["submit", "t", "--dry-run"]) == 0
    assert "launch.run -np 2" in capsys.readouterr().out
    assert submit_job.main(["delete", "t"]) == 0 and not os.path.exists(tmp_path / "t")


def test_local_smoke_command_lines(capsys):
    assert local_smoke.main(["--dry-run", "--np", "2", "-x", "BATCHSIZE=8"]) == 0
    out = capsys.readouterr().out
    assert "-np 2 -H localhost:2" in out and "-x FAKE=True" in out and "-x BATCHSIZE=8" in out
    assert local_smoke.main(["--dry-run", "--build", "img", "--push"]) == 0
    out = capsys.readouterr().out
    assert "docker build -t img" in out and "--device=/dev/kfd" in out and "docker push img" in out


def test_create_resources_init_writes_cluster_and_env(tmp_path, monkeypatch):
    out = tmp_path / "cluster.json"
    monkeypatch.chdir(tmp_path)
    assert create_resources.main(["init", "--nodes", "a,b,c", "--gpus-per-node", "4", "--data", "/d",
                                  "--out", str(out)]) == 0
    cl = json.loads(out.read_text())
    assert cl["nodes"] == ["a", "b", "c"] and cl["gpus_per_node"] == 4 and cl["mounts"]["train"] == "/d/train"


def _write_csv(path, header, rows):
    import csv

    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_ALL)
        w.writerow(header)
        w.writerows(rows)


def test_pmc_summary_condenses_counter_csv(tmp_path):
    """scripts/pmc_summary.py: rocprofv3 --pmc counter_collection.csv -> markdown table."""
    import subprocess
    import sys

    hdr = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id",
           "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size",
           "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
           "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    k = "void (anonymous namespace)::conv_igemm_kernel<128, 128, 32, 1, false>((anonymous namespace)::IGemmParams)"
    base = [1, 1, "Agent 1", 1, 9, 9, 256 * 784, 3, k, 256, 38912, 0, 64, 0, 48]
    rows = []
    for disp in (1, 2):
        for name, val in (("SQ_WAVES", 3136), ("SQ_WAVE_CYCLES", 1e8), ("SQ_INSTS_MFMA", 3.6e6),
                          ("SQ_VALU_MFMA_BUSY_CYCLES", 5.76e7), ("SQ_WAIT_INST_ANY", 3e7),
                          ("SQ_LDS_BANK_CONFLICT", 3e5)):
            r = list(base)
            r[1] = disp
            rows.append(r + [name, val, 1000, 91000])
    rows.append([1, 3, "Agent 1", 1, 9, 9, 1024, 4, "some_torch_kernel", 256, 0, 0, 8, 0, 8,
                 "SQ_WAVES", 4, 0, 10])
    _write_csv(str(tmp_path / "run" / "sq" / "vm" / "1_counter_collection.csv"), hdr, rows)
    _write_csv(str(tmp_path / "run" / "fetch" / "vm" / "2_counter_collection.csv"), hdr,
               [base + ["FETCH_SIZE", 20480.0, 1000, 91000]])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"),
                          str(tmp_path / "run")], capture_output=True, text=True, check=True).stdout
    line = [l for l in out.splitlines() if l.startswith("| conv_igemm_kernel")]
    assert len(line) == 1 and "some_torch_kernel" not in out
    cells = [c.strip() for c in line[0].strip("|").split("|")]
    head = [c.strip() for c in [l for l in out.splitlines() if l.startswith("| kernel")][0].strip("|").split("|")]
    rec = dict(zip(head, cells))
    assert rec["blocks"] == "784" and rec["dur_us"] == "90"
    assert abs(float(rec["mfma_util%"]) - 100 * 5.76e7 / (90 * 2400 * 1024)) < 0.1
    assert abs(float(rec["TF/s"]) - 3.6e6 * 16384 / 90 / 1e6) < 1
    assert rec["hbm_MiB"] == "20.0"


def test_trace_gaps_reports_streams_and_gaps(tmp_path):
    """scripts/trace_gaps.py: per-stream busy time and main-stream gaps of the last step."""
    import subprocess
    import sys

    hdr = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id",
           "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size",
           "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X",
           "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"]

    def row(q, s, name, a, b):
        return ["KERNEL_DISPATCH", "Agent 1", q, s, 1, 1, 1, name, 1, a, b, 0, 0, 8, 0, 8, 256, 1, 1, 1024, 1, 1]

    rows = []
    t = 0
    for step in range(3):
        base = step * 10_000_000
        rows.append(row(1, 0, "conv_igemm_kernel<a>", base, base + 100_000))
        rows.append(row(1, 0, "bn_apply", base + 150_000, base + 250_000))       # 50 us gap
        rows.append(row(2, 1, "conv_wgrad_kernel<a>", base + 120_000, base + 400_000))
        rows.append(row(1, 0, "fused_sgd_kernel<true, true>", base + 5_000_000, base + 5_080_000))
    path = str(tmp_path / "1_kernel_trace.csv")
    _write_csv(path, hdr, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "trace_gaps.py"), path],
                         capture_output=True, text=True, check=True).stdout
    assert "queue 1 stream 0: 3 kernels" in out and "queue 2 stream 1: 1 kernels" in out
    assert "50.0 us" in out and "conv_igemm_kernel<a> -> bn_apply" in out


def test_flavour_images_point_at_real_scripts_and_known_settings():
    """The per-flavour Dockerfiles (reference: one image per framework flavour) only carry entry points and
    defaults: every script they start exists and every setting they export is one the scripts read."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg_src = open(os.path.join(root, "distributeddeeplearning_amd", "utils", "config.py")).read()
    for name in ("trainer.Dockerfile", "estimator.Dockerfile"):
        text = open(os.path.join(root, "deploy", "docker", name)).read()
        scripts = re.findall(r"examples/[a-z_]+\.py", text)
        assert scripts, name
        for s in scripts:
            assert os.path.exists(os.path.join(root, s)), (name, s)
        env_block = re.search(r"^ENV (.*?)(?=^\S)", text, re.S | re.M).group(1)
        keys = re.findall(r"([A-Z_]+)=", env_block)
        assert "DISTRIBUTED" in keys
        for k in keys:
            if k == "DDL_FLAVOUR":
                continue
            assert '"%s"' % k in cfg_src, (name, k)
        assert "ddlrun" in text and "BASE_IMAGE" in text


def test_pmc_step_traffic_sums_one_step_between_two_gathers(tmp_path):
    """scripts/pmc_step_traffic.py: FETCH_SIZE + WRITE_SIZE (KiB) of the dispatches between the last two loader gathers."""
    import subprocess
    import sys

    hdr = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id",
           "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size",
           "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
           "Counter_Value", "Start_Timestamp", "End_Timestamp"]

    def rows(counter, per_kernel):
        out, t = [], 0
        for step in range(3):
            for name, val in (("pool_gather_kernel", 1.0), ("conv_igemm_kernel<a>", per_kernel), ("bn_apply", per_kernel)):
                out.append([1, len(out) + 1, "Agent 1", 1, 9, 9, 1024, 1, name, 256, 0, 0, 8, 0, 8, counter,
                            val * (step + 1), t, t + 1000])
                t += 2000
        return out

    _write_csv(str(tmp_path / "run" / "fetch" / "vm" / "1_counter_collection.csv"), hdr, rows("FETCH_SIZE", 1e6))
    _write_csv(str(tmp_path / "run" / "write" / "vm" / "2_counter_collection.csv"), hdr, rows("WRITE_SIZE", 5e5))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_step_traffic.py"), str(tmp_path / "run")],
                         capture_output=True, text=True, check=True).stdout
    # the window is step 2 of 3 (values doubled): gather + two kernels
    fetch = (2.0 + 2 * 2e6) * 1024 / 1e9
    write = (2.0 + 2 * 1e6) * 1024 / 1e9
    assert "%-11s %7.2f GB over 3 dispatches" % ("FETCH_SIZE", fetch) in out
    assert "per step    %7.2f GB" % (fetch + write) in out


def test_trace_phases_splits_a_step_into_forward_backward_and_streams(tmp_path):
    """scripts/trace_phases.py: steps delimited by the loader gather, phases by the loss kernel, streams by queue."""
    import subprocess
    import sys

    hdr = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id",
           "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size",
           "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X",
           "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"]

    def row(q, s, name, a, b):
        return ["KERNEL_DISPATCH", "Agent 1", q, s, 1, 1, 1, name, 1, a, b, 0, 0, 8, 0, 8, 256, 1, 1, 1024, 1, 1]

    rows = []
    for step in range(4):
        base = step * 1_000_000
        rows.append(row(1, 0, "pool_gather_kernel", base, base + 10_000))
        rows.append(row(1, 0, "conv_igemm_kernel<1>", base + 10_000, base + 200_000))
        rows.append(row(1, 0, "softmax_xent_kernel", base + 200_000, base + 210_000))
        rows.append(row(1, 0, "conv_igemm_kernel<2>", base + 210_000, base + 350_000))
        rows.append(row(1, 0, "bn_bwd_apply_kernel", base + 380_000, base + 500_000))         # 30 us gap
        rows.append(row(2, 1, "conv_wgrad_kernel<a>", base + 250_000, base + 450_000))
        rows.append(row(2, 1, "fused_sgd_kernel<true, true>", base + 450_000, base + 460_000))  # early update: not a delimiter
        rows.append(row(1, 0, "fused_sgd_kernel<true, true>", base + 500_000, base + 520_000))
    path = str(tmp_path / "1_kernel_trace.csv")
    _write_csv(path, hdr, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "trace_phases.py"), path],
                         capture_output=True, text=True, check=True).stdout
    assert "8 kernels (6 main stream, 2 other streams)" in out
    assert "forward: 0.210 ms window" in out and "backward: 0.290 ms window" in out
    assert "other streams 2 kernels busy 0.210 ms" in out
    assert "gap   30.0 us" in out
