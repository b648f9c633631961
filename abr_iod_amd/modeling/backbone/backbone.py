"""build_backbone (mirror of maskrcnn_benchmark/modeling/backbone/backbone.py): R-50-C4 = nn.Sequential(body=ResNet)."""
from collections import OrderedDict

from torch import nn

from . import resnet


class _Body(nn.Sequential):
    def forward(self, x, prefix=None):
        return self.body(x, prefix)

    def frozen_prefix(self, x):
        return self.body.frozen_prefix(x)


def build_backbone(cfg):
    assert cfg.MODEL.BACKBONE.CONV_BODY == "R-50-C4", "every configs/voc YAML uses R-50-C4; FPN / FBNet are out of scope"
    body = resnet.ResNet(cfg)
    model = _Body(OrderedDict([("body", body)]))
    model.out_channels = cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS
    return model
