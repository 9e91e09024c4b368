"""maskunet_amd -- MI355X-native (gfx950) MaskAttn-UNet forward/backward path.

Drop-in for the model classes of Belis0811/MaskUnet (code/ade20k/ade_semantic.py:152-314,
code/cityscapes/city_instance.py:216-276): same names, signatures and state_dict keys, running on
hand-written HIP kernels through the C ABI in include/maskunet_hip.h.
"""
from .modules import (ConvBlock, DoubleConv, Down, DownSample, InstanceUNet, Mask2FormerAttention, MaskAttention, OutConv,
                      UNet, Up, UpSample, set_default_compute_dtype)
from .dp import DataParallel, shard_batch
from .losses import CrossEntropyLoss, InstanceContrastiveLoss, cross_entropy, mean_iou, pixel_cross_entropy_nhwc
from .ops import resize_labels_u8, resize_u8_to_nhwc
from .optim import FusedAdamW
from .graph import GraphedStep
from ._lib import get_float32_matmul_precision, set_float32_matmul_precision

__all__ = ["ConvBlock", "DownSample", "UpSample", "Mask2FormerAttention", "UNet", "InstanceUNet", "DoubleConv", "Down", "Up",
           "MaskAttention", "OutConv", "set_default_compute_dtype", "DataParallel", "shard_batch", "pixel_cross_entropy_nhwc",
           "mean_iou", "InstanceContrastiveLoss", "FusedAdamW", "CrossEntropyLoss", "cross_entropy", "GraphedStep",
           "resize_u8_to_nhwc", "resize_labels_u8", "set_float32_matmul_precision", "get_float32_matmul_precision"]
__version__ = "0.1.0"
