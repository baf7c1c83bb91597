from .tecogan_nets import (FRNet, FNet, SRNet, SpatioTemporalDiscriminator,
                           SpatialDiscriminator)


def define_generator(opt):
    """Factory with the reference's contract (codes/models/networks/__init__.py:4-19)."""
    net_G_opt = opt['model']['generator']
    if net_G_opt['name'].lower() == 'frnet':
        return FRNet(in_nc=net_G_opt['in_nc'], out_nc=net_G_opt['out_nc'],
                     nf=net_G_opt['nf'], nb=net_G_opt['nb'],
                     degradation=opt['dataset']['degradation']['type'],
                     scale=opt['scale'])
    raise ValueError(f'Unrecognized generator: {net_G_opt["name"]}')


def define_discriminator(opt):
    """codes/models/networks/__init__.py:22-47 ."""
    net_D_opt = opt['model']['discriminator']
    if opt['dataset']['degradation']['type'] == 'BD':
        spatial_size = opt['dataset']['train']['crop_size']
    else:
        spatial_size = opt['dataset']['train']['gt_crop_size']
    if net_D_opt['name'].lower() == 'stnet':
        return SpatioTemporalDiscriminator(
            in_nc=net_D_opt['in_nc'], spatial_size=spatial_size,
            tempo_range=net_D_opt['tempo_range'],
            degradation=opt['dataset']['degradation']['type'], scale=opt['scale'])
    if net_D_opt['name'].lower() == 'snet':
        return SpatialDiscriminator(in_nc=net_D_opt['in_nc'], spatial_size=spatial_size,
                                    use_cond=net_D_opt['use_cond'])
    raise ValueError(f'Unrecognized discriminator: {net_D_opt["name"]}')
