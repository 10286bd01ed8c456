"""Frame ingest on the device: the caller-side ``cv2.resize`` + BGR->RGB of reference ``run.py:179-183`` and
``:415-421`` as one HIP kernel, so raw uint8 frames at source resolution are all that crosses PCIe.

    ing = FrameIngest(device)
    frame_rgb = ing.frame(decoded_bgr_u8, H, 2 * H, bgr=True)      # torch.uint8 [H, 2H, 3] on the device
    label = ing.label(teacher_png_u8, H, 2 * H)                    # torch.uint8 [H, 2H]

The arithmetic is that of ``ams_amd.utils.resize_linear`` / ``resize_nearest`` (bit-identical; those restate OpenCV's
INTER_LINEAR / INTER_NEAREST, the former the former in its fixed-point uint8 arithmetic, bit for bit).  No CPU fallback: without the HIP
library or a GPU this raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Union

import numpy as np
import torch

from . import hip

ArrayLike = Union[np.ndarray, torch.Tensor]


class FrameIngest:
    def __init__(self, device: Union[str, torch.device, None] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("FrameIngest needs an MI355X (there is no CPU fallback; the host path is ams_amd.utils.resize_*)")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.lib = hip.lib()

    def _src(self, a: ArrayLike, channels: int) -> torch.Tensor:
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        assert t.dtype == torch.uint8, "ingest takes uint8 images, got %s" % t.dtype
        assert (t.dim() == 3 and t.shape[2] == channels) if channels > 1 else t.dim() == 2, "bad image shape %s" % (tuple(t.shape),)
        return t.to(self.device, non_blocking=True).contiguous()

    def _run(self, src: torch.Tensor, channels: int, mode: int, swap: bool, H: int, W: int) -> torch.Tensor:
        out = torch.empty((H, W, channels) if channels > 1 else (H, W), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            hip.check(self.lib.ams_ingest_resize_u8(C.c_void_p(src.data_ptr()), int(src.shape[0]), int(src.shape[1]), channels, mode,
                                                    int(swap), C.c_void_p(out.data_ptr()), H, W, st), "ams_ingest_resize_u8")
        return out

    def frame(self, image: ArrayLike, H: int, W: int, bgr: bool = False) -> torch.Tensor:
        """cv2.resize(image, (W, H)) [INTER_LINEAR] (+ cv2.COLOR_BGR2RGB when ``bgr``) -> uint8 [H, W, 3] on the device."""
        return self._run(self._src(image, 3), 3, hip.RESIZE_LINEAR, bgr, H, W)

    def label(self, label: ArrayLike, H: int, W: int) -> torch.Tensor:
        """cv2.resize(label, (W, H), interpolation=INTER_NEAREST) -> uint8 [H, W] on the device."""
        return self._run(self._src(label, 1), 1, hip.RESIZE_NEAREST, False, H, W)
