#!/usr/bin/env python
"""ImageNet (ILSVRC2012) data preparation — the counterpart of the reference's
``00_DataProcessing.ipynb`` + ``valprep.sh``.

    python tools/prepare_imagenet.py train  ILSVRC2012_img_train.tar  /data/train
    python tools/prepare_imagenet.py val    ILSVRC2012_img_val.tar    /data/validation --mapping map.txt
    python tools/prepare_imagenet.py pack   /data/train  train.tar.gz          # re-tar for upload

* ``train``: the outer tar holds one tar per class (``n01440764.tar`` ...); each is extracted
  into its own class directory (notebook cells :27-41).
* ``val``: the 50 000 validation JPEGs arrive flat; they are moved into class directories
  according to a mapping.  The reference ships the mapping as a 51 002-line generated shell
  script of ``mkdir -p <wnid>`` / ``mv <file> <wnid>/`` lines; here it is *data*: ``--mapping``
  accepts either a two-column text file (``ILSVRC2012_val_00000001.JPEG n01751748``) or such a
  ``valprep.sh`` file, which is parsed, never executed.
"""
import argparse
import io
import os
import re
import shutil
import sys
import tarfile

_MV = re.compile(r"^\s*mv\s+(\S+)\s+(\S+?)/?\s*$")
_MKDIR = re.compile(r"^\s*mkdir\s+-p\s+(\S+)\s*$")


def parse_mapping(path):
    """{filename: wnid} from a two-column file or a valprep.sh-style script."""
    mapping, classes = {}, set()
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            m = _MV.match(line)
            if m:
                mapping[os.path.basename(m.group(1))] = m.group(2).strip("/")
                continue
            m = _MKDIR.match(line)
            if m:
                classes.add(m.group(1))
                continue
            parts = line.replace(",", " ").split()
            if len(parts) == 2:
                mapping[parts[0]] = parts[1]
    classes.update(mapping.values())
    return mapping, sorted(classes)


def _safe_extract(tar, dest):
    dest = os.path.realpath(dest)
    for member in tar:
        target = os.path.realpath(os.path.join(dest, member.name))
        if not target.startswith(dest + os.sep) and target != dest:
            raise RuntimeError("refusing path outside the destination: %s" % member.name)
        if member.isfile() or member.isdir():
            tar.extract(member, dest)


def prepare_train(archive, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    n_classes = n_images = 0
    with tarfile.open(archive) as outer:
        for member in outer:
            if not member.isfile() or not member.name.endswith(".tar"):
                continue
            wnid = os.path.splitext(os.path.basename(member.name))[0]
            cdir = os.path.join(out_dir, wnid)
            os.makedirs(cdir, exist_ok=True)
            with tarfile.open(fileobj=io.BytesIO(outer.extractfile(member).read())) as inner:
                before = len(os.listdir(cdir))
                _safe_extract(inner, cdir)
                n_images += len(os.listdir(cdir)) - before
            n_classes += 1
    return n_classes, n_images


def prepare_val(archive_or_dir, out_dir, mapping_path):
    mapping, classes = parse_mapping(mapping_path)
    os.makedirs(out_dir, exist_ok=True)
    if os.path.isdir(archive_or_dir):
        src = archive_or_dir
    else:
        src = out_dir
        with tarfile.open(archive_or_dir) as tar:
            _safe_extract(tar, out_dir)
    for c in classes:
        os.makedirs(os.path.join(out_dir, c), exist_ok=True)
    moved = missing = 0
    for fname, wnid in mapping.items():
        p = os.path.join(src, fname)
        if os.path.isfile(p):
            shutil.move(p, os.path.join(out_dir, wnid, fname))
            moved += 1
        else:
            missing += 1
    return len(classes), moved, missing


def pack(directory, archive):
    mode = "w:gz" if archive.endswith("gz") else "w"
    with tarfile.open(archive, mode) as tar:
        tar.add(directory, arcname=os.path.basename(os.path.normpath(directory)))
    return archive


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    t = sub.add_parser("train"); t.add_argument("archive"); t.add_argument("out")
    v = sub.add_parser("val"); v.add_argument("archive"); v.add_argument("out")
    v.add_argument("--mapping", required=True)
    k = sub.add_parser("pack"); k.add_argument("directory"); k.add_argument("archive")
    a = ap.parse_args(argv)
    if a.cmd == "train":
        print("classes: %d images: %d" % prepare_train(a.archive, a.out))
    elif a.cmd == "val":
        print("classes: %d moved: %d missing: %d" % prepare_val(a.archive, a.out, a.mapping))
    else:
        print(pack(a.directory, a.archive))
    return 0


if __name__ == "__main__":
    sys.exit(main())
