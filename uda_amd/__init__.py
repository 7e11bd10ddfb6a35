"""Import alias for the product package.

The product lives in ``uncertainty-detection-autolabeling_amd/`` (the directory
name the build contract asks for).  A hyphenated directory cannot be imported
by name, so this shim extends its own ``__path__`` to that directory:
``import uda_amd.infer_lib`` resolves to
``uncertainty-detection-autolabeling_amd/infer_lib.py``.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_pkg = _os.path.join(_os.path.dirname(_here), "uncertainty-detection-autolabeling_amd")
if not _os.path.isdir(_pkg):  # pragma: no cover
    raise ImportError("product package directory missing: %s" % _pkg)
__path__.append(_pkg)
PACKAGE_DIR = _pkg
