#!/usr/bin/env python
"""Local 2-process smoke run — the counterpart of the reference's
``Horovod*/00_CreateImageAndTest.ipynb`` (build image, run ``mpirun -np 2 -H localhost:2`` on
fake data, stream the logs, stop, push).  

    python tools/local_smoke.py                      # bare-metal: 2 ranks via the in-repo launcher
    python tools/local_smoke.py --docker IMAGE       # inside the training image (GPU devices mapped)
    python tools/local_smoke.py --build IMAGE --push # docker build (+ push) first
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch_line(np_, script, fake_length, extra_env):
    env = {"DISTRIBUTED": "True", "FAKE": "True", "FAKE_DATA_LENGTH": str(fake_length), "EPOCHS": "1"}
    env.update(extra_env)
    x = " ".join("-x %s=%s" % kv for kv in sorted(env.items()))
    return "python -m distributeddeeplearning_amd.launch.run -np %d -H localhost:%d %s python -u %s" % (
        np_, np_, x, script)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--np", type=int, default=2)
    ap.add_argument("--script", default="examples/imagenet_pytorch.py")
    ap.add_argument("--fake-length", type=int, default=4096)
    ap.add_argument("--docker", default=None, help="run inside this image")
    ap.add_argument("--build", default=None, help="docker build -t IMAGE first")
    ap.add_argument("--push", action="store_true")
    ap.add_argument("-x", "--export", action="append", default=[])
    ap.add_argument("--dry-run", action="store_true")
    a = ap.parse_args(argv)
    extra = dict(kv.split("=", 1) for kv in a.export)
    line = launch_line(a.np, a.script, a.fake_length, extra)
    cmds = []
    if a.build:
        cmds.append(["docker", "build", "-t", a.build, "-f", "deploy/docker/Dockerfile", "."])
    image = a.docker or a.build
    if image:
        cmds.append(["docker", "run", "--rm", "--device=/dev/kfd", "--device=/dev/dri", "--group-add", "video",
                     "--group-add", "render", "--ipc=host", "--shm-size", "8G", "--security-opt",
                     "seccomp=unconfined", image, "bash", "-c", line])
    else:
        cmds.append(["bash", "-c", line])
    if a.push and a.build:
        cmds.append(["docker", "push", a.build])
    for c in cmds:
        print("+ " + " ".join(c), flush=True)
        if not a.dry_run:
            rc = subprocess.call(c, cwd=ROOT)
            if rc != 0:
                return rc
    return 0


if __name__ == "__main__":
    sys.exit(main())
