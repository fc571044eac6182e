"""Import shim: the package directory is ``wenet-celoss_amd/`` (the name the
project layout prescribes), which is not a valid Python identifier.  Importing
``wenet_celoss_amd`` executes this file, which loads that directory as the
package ``wenet_celoss_amd`` and replaces itself in ``sys.modules``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "wenet-celoss_amd")
_spec = _ilu.spec_from_file_location("wenet_celoss_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["wenet_celoss_amd"] = _mod
_spec.loader.exec_module(_mod)
