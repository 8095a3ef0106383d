"""EMCADNet (reference: multiclass_seg/EMCAD/lib/networks.py:10-128), dual-supervision configuration of BASELINE config 5:
PVTv2 encoder (lib/pvtv2.py) + EMCAD_dual decoder (lib/decoders.py) + x32/x16/x8/x4 bilinear up-sampling of the 8 K-class maps."""
import os

import torch
import torch.nn as nn

from pn2.graph import run_module
from lib import pvtv2
from lib.decoders import EMCAD_dual


class EMCADNet(nn.Module):
    def __init__(self, num_classes=1, kernel_sizes=[1, 3, 5], expansion_factor=2, dw_parallel=True, add=True, lgag_ks=3, activation='relu', encoder='pvt_v2_b2', pretrain=True, **kwargs):
        super().__init__()
        self.dual = kwargs.get('dual', False)
        if not self.dual:
            raise NotImplementedError("only the dual-supervision EMCADNet (dual=True, the PraNet-V2 configuration) is built")
        self.conv = nn.Sequential(nn.Conv2d(1, 3, kernel_size=1), nn.BatchNorm2d(3), nn.ReLU(inplace=True))
        if encoder not in ('pvt_v2_b1', 'pvt_v2_b2', 'pvt_v2_b3', 'pvt_v2_b4', 'pvt_v2_b5'):
            raise NotImplementedError(f"encoder {encoder}: only the head_dim-64 PVTv2 encoders are built")
        self.backbone = getattr(pvtv2, encoder)()
        path = f'./pretrained_pth/pvt/{encoder}.pth'
        channels = [512, 320, 128, 64]
        if pretrain is True and (os.path.exists(path) or os.environ.get('PN2_NO_PRETRAINED', '0') != '1'):
            save_model = torch.load(path)
            model_dict = self.backbone.state_dict()
            model_dict.update({k: v for k, v in save_model.items() if k in model_dict.keys()})
            self.backbone.load_state_dict(model_dict)
        self.decoder = EMCAD_dual(channels=channels, kernel_sizes=kernel_sizes, expansion_factor=expansion_factor, dw_parallel=dw_parallel, add=add, lgag_ks=lgag_ks,
                                  activation=activation, num_class=num_classes)
        self.out_head4 = nn.Conv2d(channels[0], num_classes, 1)
        self.out_head3 = nn.Conv2d(channels[1], num_classes, 1)
        self.out_head2 = nn.Conv2d(channels[2], num_classes, 1)
        self.out_head1 = nn.Conv2d(channels[3], num_classes, 1)
        self.interpolation = 'bilinear'

    def hot_parameters(self, one_channel=True):
        """Parameters forward() touches: the out_head* convs only serve the single-supervision branch."""
        return [p for n, p in self.named_parameters() if not n.startswith('out_head') and (one_channel or not n.startswith('conv.'))]

    def _build(self, eng, x):
        """forward :100-118 (dual branch)"""
        if x.C == 1:
            x = eng.conv_bn_act(x, self.conv[0], self.conv[1], relu=True, bias=self.conv[0].bias)
        x1, x2, x3, x4 = self.backbone._build_features(eng, x)
        outs = self.decoder._build(eng, x4, [x3, x2, x1])
        return [eng.bilinear(o, s) for o, s in zip(outs, [32, 16, 8, 4] * 2)]

    def forward(self, x, mode='test'):
        return list(run_module(self._build, [x], self.hot_parameters(x.shape[1] == 1), self.training))
