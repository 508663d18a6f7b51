"""Importable alias of the ``eeg-gan_amd`` source directory (a hyphen cannot appear in a Python
package name): ``import eeggan_amd.network`` loads ``eeg-gan_amd/network.py``."""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "eeg-gan_amd")
__path__ = [_src]
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
del _os, _f, _src
