"""``ddlrun``: an ``mpirun``-style launcher without MPI.

Replaces the reference's launch lines:

    mpirun -np 2 -H localhost:2 python -u imagenet_pytorch_horovod.py      (00_CreateImageAndTest.ipynb:57)
    mpirun -np N --hostfile $AZ_BATCHAI_MPI_HOST_FILE -bind-to none -map-by slot \
           -x NCCL_DEBUG=INFO -x DISTRIBUTED=True ... python -u script.py  (01_TrainPyTorchModel.ipynb:107)

with

    python -m distributeddeeplearning_amd.launch.run -np 2 -H localhost:2 python -u script.py
    python -m distributeddeeplearning_amd.launch.run -np 16 --hostfile hosts -x DISTRIBUTED=True python -u script.py

Semantics kept from mpirun: ``-np`` total ranks, ``-H host:slots[,host:slots]`` /
``--hostfile`` (lines ``host slots=N``), ``-x VAR`` / ``-x VAR=value`` environment export,
ranks mapped by slot, every rank's output tagged, and **if any rank exits abnormally the whole
job is killed and the launcher returns non-zero** (SURVEY §5.3).  OpenMPI-only options that
have no meaning here (``-bind-to``, ``-map-by``, ``-mca k v``, ``--allow-run-as-root``) are
accepted and ignored so the reference's command lines keep working.

Each rank gets ``RANK``, ``LOCAL_RANK``, ``WORLD_SIZE``, ``LOCAL_WORLD_SIZE``,
``MASTER_ADDR``, ``MASTER_PORT`` (the rendezvous contract of ``parallel.backend.init``) and
the ``OMPI_COMM_WORLD_*`` aliases.  Remote hosts are reached with ``ssh`` (the reference's
images run sshd with ``StrictHostKeyChecking no``, Dockerfile:71-77).
"""
from __future__ import annotations

import argparse
import os
import shlex
import signal
import socket
import subprocess
import sys
import threading
import time

__all__ = ["parse_hosts", "parse_hostfile", "assign_slots", "build_env", "launch", "main"]

_LOCAL_NAMES = {"localhost", "127.0.0.1", "::1"}


def parse_hosts(spec: str):
    """``"a:2,b:4"`` → ``[("a", 2), ("b", 4)]``; a missing slot count means 1."""
    hosts = []
    for item in spec.split(","):
        item = item.strip()
        if not item:
            continue
        if ":" in item:
            name, slots = item.rsplit(":", 1)
            hosts.append((name, int(slots)))
        else:
            hosts.append((item, 1))
    if not hosts:
        raise ValueError("empty host list")
    return hosts


def parse_hostfile(path: str):
    """OpenMPI hostfile lines: ``hostname slots=N`` (comments with #)."""
    hosts = []
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            parts = line.split()
            slots = 1
            for p in parts[1:]:
                if p.startswith("slots="):
                    slots = int(p.split("=", 1)[1])
            hosts.append((parts[0], slots))
    if not hosts:
        raise ValueError("hostfile %s lists no hosts" % path)
    return hosts


def assign_slots(hosts, np_total):
    """Map ranks to hosts by slot (mpirun ``-map-by slot``): fill each host in order.
    Returns ``[(rank, host, local_rank, local_size)]``."""
    capacity = sum(s for _, s in hosts)
    if np_total > capacity:
        raise ValueError("-np %d exceeds the %d available slots" % (np_total, capacity))
    placement, rank = [], 0
    for host, slots in hosts:
        take = min(slots, np_total - rank)
        for lr in range(take):
            placement.append((rank, host, lr, take))
            rank += 1
        if rank >= np_total:
            break
    return placement


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def build_env(base_env, rank, local_rank, world, local_size, master_addr, master_port, exports):
    env = dict(base_env)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
        "LOCAL_WORLD_SIZE": str(local_size), "MASTER_ADDR": master_addr,
        "MASTER_PORT": str(master_port),
        "OMPI_COMM_WORLD_RANK": str(rank), "OMPI_COMM_WORLD_SIZE": str(world),
        "OMPI_COMM_WORLD_LOCAL_RANK": str(local_rank),
        "OMPI_COMM_WORLD_LOCAL_SIZE": str(local_size),
        "HSA_ENABLE_IPC_MODE_LEGACY": base_env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
    })
    env.update(exports)
    return env


def remote_command(cwd, assignments, command):
    """Shell line run by ssh on a remote host: the first line of stdin is the job token."""
    assigns = " ".join("%s=%s" % (k, shlex.quote(v)) for k, v in assignments.items())
    return "IFS= read -r DDL_JOB_TOKEN; export DDL_JOB_TOKEN; cd %s && env %s %s" % (
        shlex.quote(cwd), assigns, " ".join(shlex.quote(c) for c in command))


def _pump(stream, tag, sink, lock):
    for raw in iter(stream.readline, b""):
        with lock:
            sink.write(tag + raw.decode(errors="replace"))
            sink.flush()
    stream.close()


def launch(command, np_total, hosts=None, exports=None, tag_output=True, timeout=None,
           master_port=None, sink=None):
    """Run ``command`` (argv list) as ``np_total`` ranks; returns the job's exit code."""
    hosts = hosts or [("localhost", np_total)]
    exports = dict(exports or {})
    # per-job secret: the engine's control plane only unpickles objects signed with it
    # (a token already present in the launcher's environment is the job's token: ssh-started
    # ranks do not inherit the environment, so it has to travel with the exports all the same)
    if "DDL_JOB_TOKEN" not in exports:
        if os.environ.get("DDL_JOB_TOKEN"):
            exports["DDL_JOB_TOKEN"] = os.environ["DDL_JOB_TOKEN"]
        else:
            import secrets

            exports["DDL_JOB_TOKEN"] = secrets.token_hex(16)
    sink = sink or sys.stdout
    placement = assign_slots(hosts, np_total)
    first_host = placement[0][1]
    local_only = all(h in _LOCAL_NAMES or h == socket.gethostname() for _, h, _, _ in placement)
    master_addr = "127.0.0.1" if local_only else first_host
    master_port = master_port or _free_port()
    procs, threads, lock = [], [], threading.Lock()
    for rank, host, lrank, lsize in placement:
        env = build_env(os.environ, rank, lrank, np_total, lsize, master_addr, master_port, exports)
        secret = None
        if host in _LOCAL_NAMES or host == socket.gethostname():
            argv = list(command)
            penv = env
        else:
            # the job token never appears on a command line (`ps` shows those): the remote shell
            # reads it from ssh's stdin, everything else travels as env assignments
            keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR",
                    "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"] + [k for k in exports if k != "DDL_JOB_TOKEN"]
            argv = ["ssh", "-o", "StrictHostKeyChecking=no", host,
                    remote_command(os.getcwd(), {k: env[k] for k in keys}, command)]
            penv = dict(os.environ)
            secret = (env.get("DDL_JOB_TOKEN", "") + "\n").encode()
        p = subprocess.Popen(argv, env=penv, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             stdin=subprocess.PIPE if secret is not None else None,
                             start_new_session=True)
        if secret is not None:
            try:
                p.stdin.write(secret)
                p.stdin.close()
            except OSError:  # ssh died at once: the exit-code watch below reports it
                pass
        procs.append(p)
        tag = "[%d]<%s>: " % (rank, host) if tag_output else ""
        t = threading.Thread(target=_pump, args=(p.stdout, tag, sink, lock), daemon=True)
        t.start()
        threads.append(t)

    def kill_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    exit_code = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        live = set(range(len(procs)))
        while live:
            for i in list(live):
                rc = procs[i].poll()
                if rc is None:
                    continue
                live.discard(i)
                if rc != 0 and exit_code == 0:
                    exit_code = rc if rc > 0 else 128 - rc
                    with lock:
                        sink.write("ddlrun: rank %d exited with code %d; terminating the job\n"
                                   % (i, rc))
                        sink.flush()
                    kill_all(signal.SIGTERM)
                    grace = time.monotonic() + 5.0
                    while time.monotonic() < grace and any(p.poll() is None for p in procs):
                        time.sleep(0.05)
                    kill_all(signal.SIGKILL)
            if deadline is not None and time.monotonic() > deadline and live:
                with lock:
                    sink.write("ddlrun: timeout after %.0f s; terminating the job\n" % timeout)
                kill_all(signal.SIGKILL)
                exit_code = exit_code or 124
                break
            time.sleep(0.02)
    except KeyboardInterrupt:
        kill_all(signal.SIGKILL)
        exit_code = 130
    for p in procs:
        try:
            p.wait(timeout=5)
        except subprocess.TimeoutExpired:
            pass
    for t in threads:
        t.join(timeout=2)
    return exit_code


def _parse(argv):
    ap = argparse.ArgumentParser(prog="ddlrun", description=__doc__.split("\n\n")[0],
                                 allow_abbrev=False)
    ap.add_argument("-np", "-n", "--np", dest="np", type=int, required=True)
    ap.add_argument("-H", "--host", dest="hosts", default=None)
    ap.add_argument("--hostfile", "-hostfile", dest="hostfile", default=None)
    ap.add_argument("-x", dest="exports", action="append", default=[])
    ap.add_argument("--timeout", type=float, default=None)
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("--no-tag", action="store_true")
    # accepted-and-ignored OpenMPI options
    ap.add_argument("-bind-to", "--bind-to", dest="_bind", default=None)
    ap.add_argument("-map-by", "--map-by", dest="_map", default=None)
    ap.add_argument("-mca", "--mca", dest="_mca", nargs=2, action="append", default=[])
    ap.add_argument("--allow-run-as-root", action="store_true")
    ap.add_argument("command", nargs=argparse.REMAINDER)
    ns = ap.parse_args(argv)
    if ns.command and ns.command[0] == "--":
        ns.command = ns.command[1:]
    if not ns.command:
        ap.error("no program to run")
    return ns


def main(argv=None):
    ns = _parse(sys.argv[1:] if argv is None else argv)
    exports = {}
    for item in ns.exports:
        if "=" in item:
            k, v = item.split("=", 1)
            exports[k] = v
        elif item in os.environ:
            exports[item] = os.environ[item]
    if ns.hostfile:
        hosts = parse_hostfile(ns.hostfile)
    elif ns.hosts:
        hosts = parse_hosts(ns.hosts)
    else:
        hosts = [("localhost", ns.np)]
    return launch(ns.command, ns.np, hosts, exports, tag_output=not ns.no_tag,
                  timeout=ns.timeout, master_port=ns.master_port)


if __name__ == "__main__":
    sys.exit(main())
