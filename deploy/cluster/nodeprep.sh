#!/usr/bin/env bash
# Node preparation for MI355X training nodes — counterpart of the reference's
# cluster_config/nodeprep.sh + docker.service (8 GB default /dev/shm for DataLoader workers),
# done as a systemd drop-in over the distribution's unit instead of a replacement unit.
# Run once per node, elevated.
set -euo pipefail
# 1. container runtime: large default shm (DataLoader workers collate through /dev/shm)
dropin="$(dirname "$0")/docker.service.d/10-ddl-training.conf"
if [ -d /etc/systemd/system ] && [ -f "$dropin" ] && systemctl cat docker.service >/dev/null 2>&1; then
    install -D -m 0644 "$dropin" /etc/systemd/system/docker.service.d/10-ddl-training.conf
    systemctl daemon-reload && systemctl restart docker
fi
# 2. GPU access for the job user, and the dmabuf IPC mode RCCL needs on this stack
getent group render >/dev/null && usermod -aG render,video "${SUDO_USER:-$USER}" || true
grep -q HSA_ENABLE_IPC_MODE_LEGACY /etc/environment || echo "HSA_ENABLE_IPC_MODE_LEGACY=0" >> /etc/environment
# 3. limits: pinned memory and open files for one process per GPU x 8
cat > /etc/security/limits.d/99-ddl.conf <<LIM
* soft memlock unlimited
* hard memlock unlimited
* soft nofile 1048576
* hard nofile 1048576
LIM
echo "node prepared"
