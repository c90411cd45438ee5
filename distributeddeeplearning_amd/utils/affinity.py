"""CPU affinity for a rank: run its host threads on the NUMA node its GPU hangs off.

The reference leaves process placement to ``mpirun -bind-to none -map-by slot``
(``HorovodPytorch/01_TrainPyTorchModel.ipynb:107``).  Here one Python process per GPU issues
~700 kernel launches per step (``host_issue_ms_per_step`` in ``bench.py``); on a two-socket
MI355X node a rank scheduled on the far socket pays the inter-socket hop on every doorbell
write, so ``hvd.init()`` narrows the rank's affinity to the CPUs of the GPU's NUMA node.
Everything here is best effort: any missing sysfs entry, an empty intersection with the
CPUs the process is allowed to use, or ``HVD_NUMA_BIND=0`` leaves the affinity untouched.
"""
from __future__ import annotations

import os

__all__ = ["parse_cpulist", "gpu_numa_node", "numa_cpus", "bind_to_gpu_numa"]


def parse_cpulist(text: str) -> set[int]:
    """``"0-3,8,10-11"`` -> {0, 1, 2, 3, 8, 10, 11} (the kernel's cpulist format)."""
    cpus: set[int] = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_numa_node(pci_bdf: str, sysfs: str = "/sys") -> int | None:
    """NUMA node of a PCI device (``0000:75:00.0``); None if unknown (-1 or no such file)."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bdf, "numa_node")) as fh:
            node = int(fh.read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def numa_cpus(node: int, sysfs: str = "/sys") -> set[int]:
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as fh:
            return parse_cpulist(fh.read())
    except (OSError, ValueError):
        return set()


def _device_bdf(index: int) -> str | None:
    try:
        import torch

        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def bind_to_gpu_numa(device_index: int, sysfs: str = "/sys", bdf: str | None = None,
                     min_cpus: int = 8) -> set[int] | None:
    """Restrict this process to the allowed CPUs on the GPU's NUMA node.

    Returns the new CPU set, or None when nothing was changed.  Never widens the current mask
    (a container's cpuset stays in force) and never narrows it below ``min_cpus`` (a rank runs
    its Python thread, the autograd thread, RCCL's proxy/watchdog threads and the loader).
    """
    if os.environ.get("HVD_NUMA_BIND", "1").strip().lower() in ("0", "false", "no", "off"):
        return None
    if not hasattr(os, "sched_getaffinity"):
        return None
    bdf = bdf or _device_bdf(device_index)
    if not bdf:
        return None
    node = gpu_numa_node(bdf, sysfs)
    if node is None:
        return None
    try:
        allowed = os.sched_getaffinity(0)
        want = numa_cpus(node, sysfs) & allowed
        if len(want) < min_cpus or want == allowed:
            return None
        os.sched_setaffinity(0, want)
        return want
    except OSError:
        return None
