"""Import harness for the upstream reference (authoring container ONLY).

Used solely by tests/golden/make_golden.py to *generate* fixtures.  Nothing
here (and nothing under /root/reference) travels to the GPU box; tests read
only the committed .npz vectors.

The reference pulls cv2 / lmdb / torchvision / skimage / IPython at module
top; none are installed, none are needed for the hot path,
so inert stub modules are planted in sys.modules first.
"""
import os
import sys
import types

REF_ROOT = '/root/reference'
REF_CODES = os.path.join(REF_ROOT, 'codes')


def available():
    return os.path.isdir(REF_CODES)


def _stub(name, attrs=()):
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package so `import a.b` resolves
    for a in attrs:
        setattr(m, a, None)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns a namespace with the reference modules needed for fixtures."""
    if not available():
        raise RuntimeError('reference tree not present (expected only in the '
                           'authoring container)')
    sys.dont_write_bytecode = True
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'

    for name in ['cv2', 'lmdb', 'torchvision', 'torchvision.models',
                 'skimage', 'skimage.measure', 'skimage.color',
                 'skimage.transform', 'IPython']:
        if name not in sys.modules:
            _stub(name)
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['skimage.measure'].compare_ssim = None
    sys.modules['skimage'].color = sys.modules['skimage.color']
    sys.modules['skimage'].transform = sys.modules['skimage.transform']
    sys.modules['IPython'].embed = None

    # scipy >= 1.13 removed signal.gaussian (used by the reference's
    # create_kernel); the windows.gaussian function is the same code.
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, 'gaussian'):
        scipy.signal.gaussian = scipy.signal.windows.gaussian

    if REF_CODES not in sys.path:
        sys.path.insert(0, REF_CODES)

    ns = types.SimpleNamespace()
    import utils.net_utils as net_utils
    import utils.data_utils as data_utils
    import models.networks.tecogan_nets as nets
    import models.networks as networks
    import metrics.model_summary as model_summary
    ns.net_utils, ns.data_utils, ns.nets = net_utils, data_utils, nets
    ns.networks, ns.model_summary = networks, model_summary
    return ns
