"""VGG19 feature extractor of the perceptual loss (codes/models/networks/vgg_nets.py:6-38)
on the HIP path: every conv is the MFMA conv3x3 kernel with fused ReLU, pooling is
maxpool2, the input normalisation its own kernel.

The reference builds `torchvision.models.vgg19(pretrained=True).features`; there is no
torchvision (and no network) here, so the architecture is restated from the published
configuration "E" and the weights come from a file the user supplies: a torchvision
`vgg19` state dict (keys `features.N.weight/bias`, classifier entries ignored) or a
state dict of this module.  Parameters are frozen, as in the reference (:12-13)."""
import torch
import torch.nn as nn

from ... import ops
from .. import train_graph as TG
from .tecogan_nets import _Conv, _block

# torchvision.models.vgg19().features: index -> layer
_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
          512, 512, 512, 512, 'M']


def vgg19_layers():
    """[(index, kind, cin, cout)] with kind in {'conv', 'relu', 'pool'}."""
    layers, cin, i = [], 3, 0
    for v in _CFG_E:
        if v == 'M':
            layers.append((i, 'pool', cin, cin)); i += 1
        else:
            layers.append((i, 'conv', cin, v)); i += 1
            layers.append((i, 'relu', v, v)); i += 1
            cin = v
    return layers


class VGGFeatureExtractor(nn.Module):
    def __init__(self, feature_indexs=(8, 17, 26, 35)):
        super().__init__()
        self.layers = vgg19_layers()
        self.feature_indexs = sorted(feature_indexs)
        kinds = {i: k for i, k, _, _ in self.layers}
        for i in self.feature_indexs:
            if kinds.get(i) not in ('relu', 'pool'):
                # the reference's features run ReLU in place, so a conv-indexed feature would
                # be read after the ReLU anyway; only post-activation taps are meaningful
                raise ValueError(f'feature index {i} is not a ReLU / pooling layer of vgg19.features')
        self.features = _block([(i, _Conv(ci, co)) for i, k, ci, co in self.layers if k == 'conv'])
        for p in self.features.parameters():
            p.requires_grad = False
        self.register_buffer('mean', torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def load_vgg19_state_dict(self, sd):
        """Accepts torchvision's vgg19 state dict or this module's own."""
        own = self.state_dict()
        picked = {k: v for k, v in sd.items() if k in own}
        missing = [k for k in own if k.startswith('features.') and k not in picked]
        if missing:
            raise KeyError(f'VGG19 weights: missing {missing[:4]}...')
        for k, v in picked.items():
            if own[k].shape != v.shape:
                raise ValueError(f'VGG19 weights: {k} has shape {tuple(v.shape)}, '
                                 f'expected {tuple(own[k].shape)}')
        self.load_state_dict(picked, strict=False)
        for p in self.features.parameters():
            ops.bump_version(p)

    def forward(self, x, tape=None):
        """x in [0,1], (n,3,h,w) -> list of feature maps (vgg_nets.py:27-40).  With a tape
        the graph is recorded so that the loss gradient reaches x."""
        out = TG.channel_norm(tape, x, self.mean.view(-1), self.std.view(-1))
        feats, last = [], self.feature_indexs[-1]
        it = iter(self.layers)
        for i, kind, _, _ in it:
            if i > last:
                break
            if kind == 'conv':
                # conv i and ReLU i+1 are one launch; the map is complete at index i+1
                out = TG.conv3x3(tape, self.features[str(i)], out, act=TG.RELU)
                next(it)
                if i + 1 in self.feature_indexs:
                    feats.append(out)
            elif kind == 'pool':
                out = TG.maxpool2(tape, out)
                if i in self.feature_indexs:
                    feats.append(out)
        return feats
