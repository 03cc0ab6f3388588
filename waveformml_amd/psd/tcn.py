"""Causal dilated 1-D conv front end (hybrid config C5): mirror of the reference's TemporalConvNet
(src/models/ConvBlocks.py:105-173, itself the locuslab TCN): per level two weight-normed Conv1d with
left padding (k-1)*d chomped on the right, ReLU, dropout, residual; dilation doubles per level.
Dense torch ops (MIOpen) -- not part of the hand-written path yet."""
from torch import nn
from torch.nn.utils import weight_norm


class _Chomp(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, x):
        return x[:, :, :-self.n].contiguous()


class TemporalBlock(nn.Module):
    def __init__(self, n_in, n_out, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        layers = []
        for cin in (n_in, n_out):
            conv = weight_norm(nn.Conv1d(cin, n_out, kernel_size, stride=stride, padding=padding, dilation=dilation))
            conv.weight.data.normal_(0, 0.01)
            layers += [conv, _Chomp(padding), nn.ReLU()]
            if dropout != 0:
                layers.append(nn.Dropout(dropout))
        self.net = nn.Sequential(*layers)
        self.downsample = nn.Conv1d(n_in, n_out, 1) if n_in != n_out else None
        if self.downsample is not None:
            self.downsample.weight.data.normal_(0, 0.01)
        self.relu = nn.ReLU()

    def forward(self, x):
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(self.net(x) + res)


class TemporalConvNet(nn.Module):
    def __init__(self, num_inputs, num_channels, kernel_size=3, dropout=0.2):
        super().__init__()
        blocks = []
        for i, n_out in enumerate(num_channels):
            d = 2 ** i
            n_in = num_inputs if i == 0 else num_channels[i - 1]
            blocks.append(TemporalBlock(n_in, n_out, kernel_size, 1, d, (kernel_size - 1) * d, dropout))
        self.network = nn.Sequential(*blocks)

    def forward(self, x):
        return self.network(x)
