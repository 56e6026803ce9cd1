"""Spectrogram -> node features (mirror of peak_extractor.py:11-82)."""
import torch
from torch import nn

from . import ops


class GPUPeakExtractorv2(nn.Module):
    """Min-max normalised log-mel + time/frequency ramps -> Conv2d(3 -> n_filters, blur_kernel, stride
    (peak_stride, 1)) -> ReLU -> (B, n_filters, N).  One fused HIP kernel (ops.peak_extract); the conv
    module is kept as the parameter holder (`convs.0.weight`, `convs.0.bias`)."""

    def __init__(self, cfg):
        super().__init__()
        self.blur_kernel = cfg["blur_kernel"]
        self.n_filters = cfg["n_filters"]
        self.stride = cfg["peak_stride"]
        kh, kw = self.blur_kernel
        self.convs = nn.Sequential(
            nn.Conv2d(3, self.n_filters, kernel_size=(kh, kw), stride=(self.stride, 1), padding=(kh // 2, kw // 2)),
            nn.ReLU())
        self.n_gpus = max(torch.cuda.device_count(), 1) if torch.cuda.is_available() else 1
        self.init_weights()

    def init_weights(self):
        conv = self.convs[0]
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(conv.bias, 0)

    def forward(self, spec_tensor):
        conv = self.convs[0]
        return ops.peak_extract(spec_tensor, conv.weight, conv.bias, self.stride)
