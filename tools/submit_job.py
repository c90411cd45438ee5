#!/usr/bin/env python
"""Job definition and submission — the counterpart of the reference's
``Horovod*/01_Train*Model.ipynb`` (build ``job.json`` with the multi-node launch line, submit,
stream stdout/stderr, delete) for an on-prem MI355X cluster.  

    python tools/submit_job.py create --name run1 --nodes 2 --ppn 8 --fake \
           --script examples/imagenet_pytorch.py --cluster deploy/cluster/cluster.json
    python tools/submit_job.py show run1        # job.json + the exact command line
    python tools/submit_job.py submit run1      # ssh (or sbatch) launch, logs under jobs/run1/
    python tools/submit_job.py logs run1 [-f]   # stream stdout.txt / stderr.txt
    python tools/submit_job.py delete run1

``job.json`` mirrors the reference's job dictionary (:107): nodeCount, the launch command
(``ddlrun -np N --hostfile ... -x VAR ... python -u script``), input directories SCRIPTS /
TRAIN / TEST, output directory MODEL, container image and stdout/err prefix.  State that the
notebooks keep in ``.env`` (names, counts; C9 ``dotenv_for``/``set_key``) is kept the same way.
"""
import argparse
import json
import os
import shlex
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributeddeeplearning_amd.utils import env as envmod  # noqa: E402


def jobs_dir():
    return os.environ.get("DDL_JOBS_DIR", os.path.join(ROOT, "jobs"))


def build_job(name, cluster, nodes, ppn, script, fake, epochs, exports, image=None):
    cl = json.load(open(cluster)) if isinstance(cluster, str) else cluster
    hosts = cl["nodes"][:nodes]
    if len(hosts) < nodes:
        raise ValueError("cluster lists %d nodes, %d requested" % (len(hosts), nodes))
    total = nodes * ppn
    mounts = cl.get("mounts", {})
    env = {"DISTRIBUTED": "True", "EPOCHS": str(epochs), "NCCL_DEBUG": "INFO",
           "HSA_ENABLE_IPC_MODE_LEGACY": "0",
           "AZ_BATCHAI_INPUT_TRAIN": mounts.get("train", ""), "AZ_BATCHAI_INPUT_TEST": mounts.get("test", ""),
           "AZ_BATCHAI_OUTPUT_MODEL": mounts.get("model", ""), "AZ_BATCHAI_JOB_TEMP_DIR": mounts.get("temp", "")}
    if fake:
        env["FAKE"] = "True"
    env.update(exports or {})
    xflags = " ".join("-x %s=%s" % (k, shlex.quote(v)) for k, v in sorted(env.items()) if v != "")
    hostfile = "$DDL_JOB_DIR/hostfile"
    cmd = ("python -m distributeddeeplearning_amd.launch.run -np {np} --hostfile {hf} -bind-to none "
           "-map-by slot {x} python -u $DDL_INPUT_SCRIPTS/{script}").format(
               np=total, hf=hostfile, x=xflags, script=script)
    return {
        "name": name,
        "properties": {
            "nodeCount": nodes,
            "processesPerNode": ppn,
            "customToolkitSettings": {"commandLine": cmd},
            "stdOutErrPathPrefix": os.path.join(jobs_dir(), name),
            "inputDirectories": [
                {"id": "SCRIPTS", "path": mounts.get("scripts", ROOT)},
                {"id": "TRAIN", "path": mounts.get("train", "")},
                {"id": "TEST", "path": mounts.get("test", "")}],
            "outputDirectories": [{"id": "MODEL", "pathPrefix": mounts.get("model", ""), "pathSuffix": "Models"}],
            "containerSettings": {"imageSourceRegistry": {"image": image or cl.get("container", {}).get("image", "")}},
        },
        "hosts": ["%s slots=%d" % (h, ppn) for h in hosts],
        "scheduler": cl.get("scheduler", "ssh"),
    }


def cmd_create(a):
    exports = dict(kv.split("=", 1) for kv in a.export)
    job = build_job(a.name, a.cluster, a.nodes, a.ppn, a.script, a.fake, a.epochs, exports, a.image)
    d = os.path.join(jobs_dir(), a.name)
    os.makedirs(d, exist_ok=True)
    envmod.write_json_to_file(job, os.path.join(d, "job.json"))
    with open(os.path.join(d, "hostfile"), "w") as fh:
        fh.write("\n".join(job["hosts"]) + "\n")
    dot = envmod.dotenv_for(ROOT)
    envmod.set_key(dot, "LAST_JOB", a.name)
    envmod.set_key(dot, "NUM_NODES", str(a.nodes))
    envmod.set_key(dot, "PROCESSES_PER_NODE", str(a.ppn))
    print(os.path.join(d, "job.json"))
    return 0


def _load(name):
    d = os.path.join(jobs_dir(), name)
    return d, json.load(open(os.path.join(d, "job.json")))


def cmd_show(a):
    d, job = _load(a.name)
    print(json.dumps(job, indent=2))
    return 0


def render_sbatch(job, d):
    p = job["properties"]
    return "\n".join([
        "#!/bin/bash", "#SBATCH --job-name=%s" % job["name"], "#SBATCH --nodes=%d" % p["nodeCount"],
        "#SBATCH --ntasks-per-node=1", "#SBATCH --gpus-per-node=%d" % p["processesPerNode"],
        "#SBATCH --output=%s/stdout.txt" % d, "#SBATCH --error=%s/stderr.txt" % d,
        "export DDL_JOB_DIR=%s" % d, "export DDL_INPUT_SCRIPTS=%s" % p["inputDirectories"][0]["path"],
        "scontrol show hostnames $SLURM_JOB_NODELIST | sed 's/$/ slots=%d/' > $DDL_JOB_DIR/hostfile" % p["processesPerNode"],
        "cd $DDL_INPUT_SCRIPTS", p["customToolkitSettings"]["commandLine"], ""])


def cmd_submit(a):
    d, job = _load(a.name)
    p = job["properties"]
    env = dict(os.environ, DDL_JOB_DIR=d, DDL_INPUT_SCRIPTS=p["inputDirectories"][0]["path"])
    if job.get("scheduler") == "slurm":
        script = os.path.join(d, "job.sbatch")
        open(script, "w").write(render_sbatch(job, d))
        if a.dry_run:
            print(script)
            return 0
        return subprocess.call(["sbatch", script])
    line = "cd %s && %s" % (shlex.quote(env["DDL_INPUT_SCRIPTS"]), p["customToolkitSettings"]["commandLine"])
    if a.dry_run:
        print(line)
        return 0
    out = open(os.path.join(d, "stdout.txt"), "ab")
    err = open(os.path.join(d, "stderr.txt"), "ab")
    proc = subprocess.Popen(["bash", "-c", line], env=env, stdout=out, stderr=err, start_new_session=True)
    open(os.path.join(d, "pid"), "w").write(str(proc.pid))
    print("submitted %s (pid %d); logs in %s" % (a.name, proc.pid, d))
    return 0


def cmd_logs(a):
    d, _ = _load(a.name)
    path = os.path.join(d, "stderr.txt" if a.stderr else "stdout.txt")
    pos = 0
    while True:
        if os.path.exists(path):
            with open(path) as fh:
                fh.seek(pos)
                sys.stdout.write(fh.read())
                pos = fh.tell()
        if not a.follow:
            return 0
        time.sleep(1.0)


def cmd_delete(a):
    d, _ = _load(a.name)
    pidf = os.path.join(d, "pid")
    if os.path.exists(pidf):
        try:
            os.killpg(int(open(pidf).read()), 15)
        except (ProcessLookupError, ValueError, PermissionError):
            pass
    import shutil

    shutil.rmtree(d)
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("create")
    c.add_argument("--name", required=True); c.add_argument("--cluster", default=os.path.join(ROOT, "deploy/cluster/cluster.json"))
    c.add_argument("--nodes", type=int, default=1); c.add_argument("--ppn", type=int, default=8)
    c.add_argument("--script", default="examples/imagenet_pytorch.py"); c.add_argument("--fake", action="store_true")
    c.add_argument("--epochs", type=int, default=1); c.add_argument("--image", default=None)
    c.add_argument("-x", "--export", action="append", default=[]); c.set_defaults(fn=cmd_create)
    for name, fn in (("show", cmd_show), ("delete", cmd_delete)):
        s = sub.add_parser(name); s.add_argument("name"); s.set_defaults(fn=fn)
    s = sub.add_parser("submit"); s.add_argument("name"); s.add_argument("--dry-run", action="store_true"); s.set_defaults(fn=cmd_submit)
    s = sub.add_parser("logs"); s.add_argument("name"); s.add_argument("-f", "--follow", action="store_true")
    s.add_argument("--stderr", action="store_true"); s.set_defaults(fn=cmd_logs)
    a = ap.parse_args(argv)
    return a.fn(a)


if __name__ == "__main__":
    sys.exit(main())
