"""`from lib.nn import SynchronizedBatchNorm2d` keeps working (reference: lib/pranet.py:7,13).

The reference vendors Synchronized-BatchNorm-PyTorch but never uses it on any executed path (every BN on
the hot path is plain nn.BatchNorm2d, pranet.py:37); per-replica statistics are also what one-process-per-GPU
data parallelism gives, so the name is an alias here.
"""
import torch.nn as nn

SynchronizedBatchNorm1d = nn.BatchNorm1d
SynchronizedBatchNorm2d = nn.BatchNorm2d
SynchronizedBatchNorm3d = nn.BatchNorm3d
