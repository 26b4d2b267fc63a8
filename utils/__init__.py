"""Import-path shim: `from utils.box_utils import ...` / `from utils.anchor_utils import ...`
resolve to the MI355X implementations, as the reference's train.py / inference.py expect.

The reference keeps its other helpers in the same package (`utils.datasets`,
`utils.transforms`, `utils.losses`, ...; a directory without `__init__.py`).  This regular
package takes the name `utils` wherever it sits on `sys.path`, so it adds every other `utils`
directory found there to its own search path, after its own: the two modules above come from
here, every other `utils.*` module from the reference.
"""
import os as _os
import sys as _sys

_own = [_os.path.abspath(_p) for _p in __path__]
for _entry in list(_sys.path):
    _d = _os.path.abspath(_os.path.join(_entry or _os.getcwd(), "utils"))
    if _os.path.isdir(_d) and _d not in _own and _d not in __path__:
        __path__.append(_d)
