"""DeepSense convolution blocks -- parameter containers with the reference's names (models/ConvModules.py:54-217).
Execution lives in focal_amd.deepsense_engine (channel-last MFMA kernels); these hold weights, BatchNorm buffers
and the geometry derived from the constructor arguments."""
import numpy as np
import torch.nn as nn


class ConvLayer2D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, padding_mode, bias, dropout_ratio, activation="GELU"):
        super().__init__()
        self.inc, self.out = in_channels, out_channels
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              padding_mode=padding_mode, bias=bias)
        self.batch_norm = nn.BatchNorm2d(out_channels, eps=1e-5, momentum=0.1, track_running_stats=True)
        self.dropout_ratio = dropout_ratio


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, in_spectrum_len, interval_num=9, conv_lens=[[1, 3], [1, 3], [1, 3]],
                 dropout_ratio=0, num_inter_layers=2, in_stride=1):
        super().__init__()
        self.conv_lens, self.num_inter_layers, self.in_stride = conv_lens, num_inter_layers, in_stride
        if conv_lens[1][0] > 1 or conv_lens[0][0] > 1:
            raise NotImplementedError("time-fusing convolutions (kernel height > 1) are not used by any shipped config")
        half = int(out_channels / 2)
        strided = not (in_stride == 1 or int(np.max(in_stride)) == 1)
        self.conv_layer_in = ConvLayer2D(in_channels, half, kernel_size=conv_lens[0], stride=in_stride,
                                         padding="valid" if strided else "same", padding_mode="zeros", bias=True,
                                         dropout_ratio=dropout_ratio)
        self.conv_layers_inter = nn.ModuleList([
            ConvLayer2D(half, half, kernel_size=conv_lens[1], stride=1, padding="same", padding_mode="zeros", bias=True,
                        dropout_ratio=dropout_ratio) for _ in range(num_inter_layers)])
        self.out_spectrum = in_spectrum_len if not strided else int(in_spectrum_len / in_stride[1])
        self.conv_layer_out = nn.Conv1d(half * self.out_spectrum, out_channels, kernel_size=1, stride=1, padding="same",
                                        padding_mode="zeros", bias=True)
        self.geometry = dict(C=half, C_out=out_channels, S=self.out_spectrum, k_in=conv_lens[0][1], k=conv_lens[1][1],
                             stride=in_stride[1] if strided else 1, pad_in=0 if strided else (conv_lens[0][1] - 1) // 2,
                             n_inter=num_inter_layers)
