#!/usr/bin/env python
"""Cluster bring-up bookkeeping — the counterpart of the reference's ``01_CreateResources.ipynb``
(resource group, storage, file share, NFS, fixed-size GPU cluster, names written to ``.env``)
for an on-prem MI355X cluster reachable over ssh.  

    python tools/create_resources.py init --nodes node0,node1 --gpus-per-node 8 --data /data
    python tools/create_resources.py nodeprep [--dry-run]    # run deploy/cluster/nodeprep.sh on every node
    python tools/create_resources.py upload SRC DST           # rsync scripts/data to every node
    python tools/create_resources.py status                   # rocm-smi / reachability per node

The cluster is fixed-size (min = max, like ``--min $NUM_NODES --max $NUM_NODES`` :299);
elastic resizing is out of scope exactly as in the reference.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributeddeeplearning_amd.utils import env as envmod  # noqa: E402

CLUSTER_JSON = os.path.join(ROOT, "deploy", "cluster", "cluster.json")


def cmd_init(a):
    cl = json.load(open(CLUSTER_JSON))
    cl["nodes"] = [n for n in a.nodes.split(",") if n]
    cl["gpus_per_node"] = a.gpus_per_node
    cl["scheduler"] = a.scheduler
    if a.data:
        cl["mounts"]["train"] = os.path.join(a.data, "train")
        cl["mounts"]["test"] = os.path.join(a.data, "validation")
    out = a.out or CLUSTER_JSON
    envmod.write_json_to_file(cl, out)
    dot = envmod.dotenv_for(ROOT)
    for k, v in (("CLUSTER_NAME", cl["cluster_name"]), ("NUM_NODES", len(cl["nodes"])),
                 ("PROCESSES_PER_NODE", a.gpus_per_node), ("GPU_TYPE", "MI355X"),
                 ("FILE_SHARE_NAME", a.data or "")):
        envmod.set_key(dot, k, str(v))
    print("wrote %s and %s" % (out, dot))
    return 0


def _each(cl, remote_cmd, dry):
    rc = 0
    for n in cl["nodes"]:
        cmd = ["ssh", "-o", "StrictHostKeyChecking=no", n, remote_cmd]
        print("+ " + " ".join(cmd))
        if not dry:
            rc |= subprocess.call(cmd)
    return rc


def cmd_nodeprep(a):
    cl = json.load(open(CLUSTER_JSON))
    script = os.path.join(cl["mounts"]["scripts"], cl["node_setup"]["command"])
    return _each(cl, ("sudo " if cl["node_setup"].get("elevated") else "") + script, a.dry_run)


def cmd_upload(a):
    cl = json.load(open(CLUSTER_JSON))
    rc = 0
    for n in cl["nodes"]:
        cmd = ["rsync", "-a", "--delete", a.src, "%s:%s" % (n, a.dst)]
        print("+ " + " ".join(cmd))
        if not a.dry_run:
            rc |= subprocess.call(cmd)
    return rc


def cmd_status(a):
    cl = json.load(open(CLUSTER_JSON))
    return _each(cl, "hostname && (rocm-smi --showuse --showmemuse || true)", a.dry_run)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    i = sub.add_parser("init"); i.add_argument("--nodes", required=True); i.add_argument("--gpus-per-node", type=int, default=8)
    i.add_argument("--data", default=None); i.add_argument("--scheduler", default="ssh", choices=["ssh", "slurm"])
    i.add_argument("--out", default=None); i.set_defaults(fn=cmd_init)
    for name, fn in (("nodeprep", cmd_nodeprep), ("status", cmd_status)):
        s = sub.add_parser(name); s.add_argument("--dry-run", action="store_true"); s.set_defaults(fn=fn)
    u = sub.add_parser("upload"); u.add_argument("src"); u.add_argument("dst"); u.add_argument("--dry-run", action="store_true")
    u.set_defaults(fn=cmd_upload)
    a = ap.parse_args(argv)
    return a.fn(a)


if __name__ == "__main__":
    sys.exit(main())
