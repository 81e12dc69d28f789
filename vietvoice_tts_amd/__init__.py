"""Import alias for the hyphenated package directory ``vietvoice-tts_amd/``.

The repo layout names the package ``vietvoice-tts_amd`` (not a valid Python
identifier), so this stub makes ``import vietvoice_tts_amd`` resolve to it by
extending ``__path__`` and executing the real ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vietvoice-tts_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
