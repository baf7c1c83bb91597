from .tecogan_nets import FRNet, FNet, SRNet


def define_generator(opt):
    """Factory with the reference's contract (codes/models/networks/__init__.py:4-19)."""
    net_G_opt = opt['model']['generator']
    if net_G_opt['name'].lower() == 'frnet':
        return FRNet(in_nc=net_G_opt['in_nc'], out_nc=net_G_opt['out_nc'],
                     nf=net_G_opt['nf'], nb=net_G_opt['nb'],
                     degradation=opt['dataset']['degradation']['type'],
                     scale=opt['scale'])
    raise ValueError(f'Unrecognized generator: {net_G_opt["name"]}')
