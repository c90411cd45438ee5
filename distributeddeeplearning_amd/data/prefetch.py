"""Host loader -> HBM pipeline: upload and convert batch i+1 on a copy stream while step i runs.

The reference's hot loop does ``data.cuda(non_blocking=True)`` on the compute stream at the top
of every step (``HorovodPytorch/src/imagenet_pytorch_horovod.py:211``): 154 MB of normalised
fp32 NCHW per 256-image batch, then (here) a cast and a layout change, all in front of the first
convolution.  On MI355X the step is 19 ms, so that serial prologue is worth 15-20 % of it.

``DevicePrefetcher`` wraps any host loader that yields ``(images, labels)``:

* ``depth`` batches are in flight: pinned host tensor -> ``hipMemcpyAsync`` on a copy stream with
  a hardware queue of its own -> kernel K20 (``ops.kernels.nchw_to_nhwc_normalize``: NCHW
  fp32/uint8 -> NHWC bf16, mean/sd fused) on the same stream -> event;
* the consumer's stream waits for that event only (no host sync) and gets compute-ready
  channels_last bf16 tensors, which ``training.loop.to_device_batch`` passes through untouched;
* with ``uint8`` batches (``train_transform(..., raw_uint8=True)``) PCIe carries a quarter of the
  bytes and the workers skip the float conversion; normalisation happens inside K20.

On a CPU device it is a transparent pass-through, so training scripts need no second code path.
"""
from __future__ import annotations

from collections import deque

import torch

from ..utils.config import RGB_MEAN, RGB_SD

__all__ = ["DevicePrefetcher"]


class DevicePrefetcher:
    def __init__(self, loader, device, dtype=torch.bfloat16, depth=2, mean=RGB_MEAN, sd=RGB_SD):
        self.loader = loader
        self.device = torch.device(device)
        self.dtype = dtype
        self.depth = max(1, int(depth))
        self.mean, self.sd = tuple(mean), tuple(sd)
        self._stream = None

    def __len__(self):
        return len(self.loader)

    @property
    def sampler(self):
        return getattr(self.loader, "sampler", None)

    # ---- one batch onto the device, on the copy stream ------------------------------------------
    def _convert(self, data):
        """Device tensor in loader layout -> what the model consumes."""
        if data.dim() != 4:
            return data.to(self.dtype) if data.is_floating_point() else data
        use_kernel = (self.dtype == torch.bfloat16 and data.shape[1] <= 4 and data.is_contiguous()
                      and data.dtype in (torch.uint8, torch.float32))
        if use_kernel:
            from ..ops import kernels as K

            if data.dtype == torch.uint8:  # raw pixels: scale to [0, 1], then (x - mean) / sd
                return K.nchw_to_nhwc_normalize(data, self.mean[: data.shape[1]],
                                                self.sd[: data.shape[1]], in_scale=1.0 / 255.0)
            return K.nchw_to_nhwc_normalize(data)  # already normalised by the transforms
        if data.dtype == torch.uint8:
            c = data.shape[1]
            mean = torch.tensor(self.mean[:c], device=data.device).view(1, c, 1, 1)
            sd = torch.tensor(self.sd[:c], device=data.device).view(1, c, 1, 1)
            data = (data.float() / 255.0 - mean) / sd
        return data.to(self.dtype).contiguous(memory_format=torch.channels_last)

    def _stage(self, batch):
        data, target = batch
        if not data.is_pinned():
            data = data.pin_memory()
        if not target.is_pinned():
            target = target.pin_memory()
        with torch.cuda.stream(self._stream):
            d = data.to(self.device, non_blocking=True)
            t = target.to(self.device, non_blocking=True)
            out = self._convert(d)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        # the pinned host tensors must outlive the copy: they travel with the queue entry
        return out, t, ev, (data, target)

    def __iter__(self):
        if self.device.type != "cuda":
            for data, target in self.loader:
                if data.dtype == torch.uint8 and data.dim() == 4:
                    data = self._convert(data)
                yield data, target
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device)
        it = iter(self.loader)
        queue = deque()
        exhausted = False
        while True:
            while not exhausted and len(queue) < self.depth:
                try:
                    queue.append(self._stage(next(it)))
                except StopIteration:
                    exhausted = True
            if not queue:
                return
            out, t, ev, _host = queue.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # the allocator must not hand these blocks to the copy stream again while the
            # consumer's stream still reads them
            out.record_stream(cur)
            t.record_stream(cur)
            yield out, t
