from .vsr_model import VSRModel
from .vsrgan_model import VSRGANModel

vsr_model_lst = ['frvsr']
vsrgan_model_lst = ['tecogan']


def define_model(opt):
    """codes/models/__init__.py:16-26."""
    name = opt['model']['name'].lower()
    if name in vsr_model_lst:
        return VSRModel(opt)
    if name in vsrgan_model_lst:
        return VSRGANModel(opt)
    raise ValueError(f'Unrecognized model: {opt["model"]["name"]}')
