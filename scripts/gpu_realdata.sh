#!/bin/bash
# Real-data path end to end on the GPU: a generated class-per-directory JPEG tree through
# ImageFolder workers -> uint8 transport -> DevicePrefetcher (copy stream + K20) -> training and
# validation of examples/imagenet_pytorch.py.
set -o pipefail
mkdir -p gpurun_out
root=/tmp/ddl_fake_imagenet
rm -rf $root
python - <<'PY' || exit 1
import os, numpy as np
from PIL import Image
rng = np.random.default_rng(0)
for split, n in (("train", 64), ("val", 16)):
    for c in range(8):
        d = "/tmp/ddl_fake_imagenet/%s/n%04d" % (split, c)
        os.makedirs(d)
        for i in range(n):
            h, w = int(rng.integers(200, 300)), int(rng.integers(200, 320))
            a = (rng.integers(0, 256, size=(h // 8, w // 8, 3), dtype=np.uint8)).repeat(8, 0).repeat(8, 1)
            a[..., c % 3] = np.minimum(255, a[..., c % 3].astype(np.int32) + 20 * c).astype(np.uint8)
            Image.fromarray(a).save(os.path.join(d, "img%03d.jpg" % i), quality=85)
print("generated")
PY
export FAKE=False INPUT_TRAIN_DIR=$root/train INPUT_TEST_DIR=$root/val VALIDATION=True EPOCHS=2 BATCHSIZE=64 NUM_WORKERS=8 LOG_INTERVAL=4 OUTPUT_MODEL_DIR=/tmp/ddl_realdata_model
rm -rf $OUTPUT_MODEL_DIR
for u in True False; do
  echo "== UINT8_TRANSPORT=$u"
  UINT8_TRANSPORT=$u timeout -k 10 300 python -u examples/imagenet_pytorch.py > gpurun_out/realdata_$u.log 2>&1 || { tail -30 gpurun_out/realdata_$u.log; exit 1; }
  grep -i "loss\|images/sec\|top" gpurun_out/realdata_$u.log | tail -6
  rm -rf $OUTPUT_MODEL_DIR
done
