"""RecurrentBlock parameter container (reference: models/RecurrentModule.py:5-31); executed by deepsense_engine."""
import torch.nn as nn


class RecurrentBlock(nn.Module):
    def __init__(self, in_channel, out_channel, num_layers=2, dropout_ratio=0) -> None:
        super().__init__()
        self.gru = nn.GRU(in_channel, out_channel, num_layers, bias=True, batch_first=True, dropout=dropout_ratio, bidirectional=True)
