"""Fusion blocks of the reference (models/FusionModules.py).  Neither runs in FOCAL pretraining on a 1-location
dataset ("dead on MOD pretrain"), but their parameters are part of `state_dict()`, so the parameter
containers are kept for checkpoint interchange; calling them is an error rather than a silent CPU path."""
import torch.nn as nn


class MeanFusionBlock(nn.Module):
    def forward(self, *a, **k):
        raise NotImplementedError("multi-location fusion is outside the MI355X FOCAL pretraining hot path")


class TransformerFusionBlock(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout_rate, attention_dropout_rate):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dim)
        self.mha = nn.MultiheadAttention(embed_dim, num_heads, dropout=attention_dropout_rate, batch_first=True)

    def forward(self, *a, **k):
        raise NotImplementedError("attention fusion (class_head=True) is outside the MI355X FOCAL pretraining hot path")
