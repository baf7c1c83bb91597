"""Import shim: the package directory is named `tecogan-pytorch_amd` (not a
valid Python identifier); this module makes it importable as
`tecogan_pytorch_amd` by acting as its package object."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'tecogan-pytorch_amd')]
__file__ = _os.path.join(__path__[0], '__init__.py')
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, 'exec'))
