"""Device-side image pipeline: the reference's `transforms.Compose([RandomHorizontalFlip(), ToTensor(), preprocess])`
(data/src/cifar10.py:14-25, data/src/utils.py:7-25) for batches that are already on the GPU as uint8 HWC, and
`postprocess` for sampled images.  At > 10^4 images/s per GPU the CPU ToTensor / preprocess path cannot feed the step."""
import torch

from . import capi
from .capi import check, ptr


def preprocess_images(u8_bhwc, flip=None):
    """uint8 [B,H,W,C] (CUDA) -> fp32 [B,C,H,W] = ((u/255)*255)/256 - 0.5, bit-identical to ToTensor + preprocess.
    flip: optional bool/uint8 [B]; True mirrors that sample horizontally (RandomHorizontalFlip's draw is the caller's)."""
    capi.require_cuda(u8_bhwc, "images")
    if u8_bhwc.dtype != torch.uint8 or u8_bhwc.dim() != 4:
        raise TypeError("expected a uint8 [B,H,W,C] tensor")
    u = u8_bhwc.contiguous()
    B, H, W, C = u.shape
    f = None
    if flip is not None:
        f = flip.to(device=u.device, dtype=torch.uint8).contiguous()
        if f.numel() != B:
            raise ValueError("flip must have one entry per sample")
    out = torch.empty((B, C, H, W), device=u.device, dtype=torch.float32)
    check(capi.lib().nfk_preprocess_u8(ptr(u), ptr(f), ptr(out), B, H, W, C, capi.stream()), "nfk_preprocess_u8")
    return out


def postprocess_images(x_bchw):
    """fp32 [B,C,H,W] (CUDA) -> uint8 [B,H,W,C]: clamp(x,-0.5,0.5)+0.5, *255, clamp, truncate (data/src/utils.py:21-25)."""
    capi.require_cuda(x_bchw, "images")
    x = x_bchw.to(torch.float32).contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), device=x.device, dtype=torch.uint8)
    check(capi.lib().nfk_postprocess_u8(ptr(x), ptr(out), B, H, W, C, capi.stream()), "nfk_postprocess_u8")
    return out
