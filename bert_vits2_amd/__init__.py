"""Importable alias for the ``bert-vits2_amd/`` package directory.

The product package lives in ``bert-vits2_amd/`` (a name Python cannot import
directly because of the hyphen).  This stub makes ``import bert_vits2_amd``
resolve every submodule from that directory; it holds no code of its own.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "bert-vits2_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), "r") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
