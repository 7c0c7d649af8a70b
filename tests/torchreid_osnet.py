"""TEST INFRASTRUCTURE: an independent PyTorch OSNet with torchreid's module structure and parameter names
(torchreid/models/osnet.py: ConvLayer, Conv1x1, Conv1x1Linear, LightConv3x3, ChannelGate, OSBlock, OSNet),
written from the published architecture.  Its state_dict is what a torchreid checkpoint contains; it is the
reference the checkpoint loader (fastmot_amd/models/torchreid_weights.py) is tested against."""
import torch
from torch import nn
from torch.nn import functional as F


class ConvLayer(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class Conv1x1(ConvLayer):
    def __init__(self, cin, cout):
        super().__init__(cin, cout, 1)


class Conv1x1Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return self.bn(self.conv(x))


class LightConv3x3(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False, groups=cout)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv2(self.conv1(x))))


class ChannelGate(nn.Module):
    def __init__(self, c, reduction=16):
        super().__init__()
        self.fc1 = nn.Conv2d(c, max(c // reduction, 1), 1, bias=True)
        self.fc2 = nn.Conv2d(max(c // reduction, 1), c, 1, bias=True)

    def forward(self, x):
        g = F.adaptive_avg_pool2d(x, 1)
        g = torch.sigmoid(self.fc2(F.relu(self.fc1(g))))
        return x * g


class OSBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        mid = cout // 4
        self.conv1 = Conv1x1(cin, mid)
        self.conv2a = LightConv3x3(mid, mid)
        self.conv2b = nn.Sequential(LightConv3x3(mid, mid), LightConv3x3(mid, mid))
        self.conv2c = nn.Sequential(*[LightConv3x3(mid, mid) for _ in range(3)])
        self.conv2d = nn.Sequential(*[LightConv3x3(mid, mid) for _ in range(4)])
        self.gate = ChannelGate(mid)
        self.conv3 = Conv1x1Linear(mid, cout)
        self.downsample = Conv1x1Linear(cin, cout) if cin != cout else None

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = self.gate(self.conv2a(x1)) + self.gate(self.conv2b(x1)) + self.gate(self.conv2c(x1)) + \
            self.gate(self.conv2d(x1))
        x3 = self.conv3(x2)
        ident = self.downsample(x) if self.downsample is not None else x
        return F.relu(x3 + ident)


class OSNet(nn.Module):
    def __init__(self, channels, feature_dim=512, num_classes=10):
        super().__init__()
        c0, c1, c2, c3 = channels
        self.conv1 = ConvLayer(3, c0, 7, stride=2, padding=3)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.conv2 = nn.Sequential(OSBlock(c0, c1), OSBlock(c1, c1), nn.Sequential(Conv1x1(c1, c1), nn.AvgPool2d(2, stride=2)))
        self.conv3 = nn.Sequential(OSBlock(c1, c2), OSBlock(c2, c2), nn.Sequential(Conv1x1(c2, c2), nn.AvgPool2d(2, stride=2)))
        self.conv4 = nn.Sequential(OSBlock(c2, c3), OSBlock(c3, c3))
        self.conv5 = Conv1x1(c3, c3)
        self.fc = nn.Sequential(nn.Linear(c3, feature_dim), nn.BatchNorm1d(feature_dim), nn.ReLU(inplace=True))
        self.classifier = nn.Linear(feature_dim, num_classes)

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        x = self.conv5(self.conv4(self.conv3(self.conv2(x))))
        return self.fc(F.adaptive_avg_pool2d(x, 1).flatten(1))      # eval mode: the feature vector


def random_osnet(channels, seed=0):
    torch.manual_seed(seed)
    m = OSNet(channels)
    for mod in m.modules():
        if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
            mod.weight.data.uniform_(0.8, 1.2)
            mod.bias.data.normal_(0, 0.1)
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.8, 1.2)
    return m.eval()
