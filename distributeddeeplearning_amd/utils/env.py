"""``.env`` and JSON helpers for job hand-off files.

Parity target: reference ``common/utils.py:12-31`` (``dotenv_for`` locates or creates a
``.env`` file, ``get_password`` captures a secret interactively, ``write_json_to_file`` writes
``job.json``).  python-dotenv is not a dependency here: the tiny KEY=VALUE format is parsed
in-tree.
"""
from __future__ import annotations

import getpass
import json
import os

__all__ = ["dotenv_for", "load_dotenv", "set_key", "get_key", "get_password",
           "write_json_to_file"]


def dotenv_for(start_dir=None, filename=".env") -> str:
    """Path of the nearest ``.env`` walking upwards from ``start_dir``; created if absent."""
    here = os.path.abspath(start_dir or os.getcwd())
    probe = here
    while True:
        candidate = os.path.join(probe, filename)
        if os.path.isfile(candidate):
            return candidate
        parent = os.path.dirname(probe)
        if parent == probe:
            break
        probe = parent
    created = os.path.join(here, filename)
    open(created, "a").close()
    return created


def _parse(path):
    pairs = {}
    if not os.path.isfile(path):
        return pairs
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#") or "=" not in line:
                continue
            if line.startswith("export "):
                line = line[len("export "):]
            key, value = line.split("=", 1)
            value = value.strip()
            if len(value) >= 2 and value[0] == value[-1] and value[0] in "\"'":
                value = value[1:-1]
            pairs[key.strip()] = value
    return pairs


def load_dotenv(path=None, override=False) -> dict:
    """Export the file's pairs into ``os.environ`` and return them."""
    pairs = _parse(path or dotenv_for())
    for k, v in pairs.items():
        if override or k not in os.environ:
            os.environ[k] = v
    return pairs


def get_key(path, key, default=None):
    return _parse(path).get(key, default)


def set_key(path, key, value) -> None:
    pairs = _parse(path)
    pairs[key] = str(value)
    tmp = path + ".tmp"
    with open(tmp, "w") as fh:
        for k, v in pairs.items():
            fh.write('%s="%s"\n' % (k, v))
    os.replace(tmp, path)


def get_password(dotenv_path=None, key="PASSWORD", prompt="Enter password:") -> str:
    """Return the stored secret, asking once and persisting it if it is not stored yet."""
    path = dotenv_path or dotenv_for()
    stored = get_key(path, key)
    if stored:
        return stored
    secret = getpass.getpass(prompt)
    set_key(path, key, secret)
    return secret


def write_json_to_file(obj, filename, mode="w") -> None:
    with open(filename, mode) as fh:
        json.dump(obj, fh, indent=4)
