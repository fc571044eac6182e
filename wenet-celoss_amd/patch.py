"""Drop-in seam for the unchanged reference tree.

The reference has no plugin registry: `wenet/utils/init_model.py:16-22` imports its
model classes by name,

    from wenet.transducer.joint import TransducerJoint
    from wenet.transducer.predictor import (ConvPredictor, EmbeddingPredictor, RNNPredictor)
    from wenet.transducer.transducer import Transducer
    from wenet.transformer.ctc import CTC

and builds the model from them at `:92-102`.  `install()` registers replacement modules under exactly
those four names in `sys.modules` *before* `wenet.utils.init_model` is imported, so that
`wenet/bin/train.py` and `wenet/bin/recognize.py` construct the MI355X-backed classes without a single
edit to the reference:

    import wenet_celoss_amd.patch as patch; patch.install()      # e.g. from sitecustomize / a 2-line launcher
    runpy.run_module("wenet.bin.train", run_name="__main__")

Everything else of the reference (encoder, attention decoder, ContextBias, dataset, executor) is left
alone: it is stock PyTorch and runs on PyTorch-ROCm as it is.  The replaced `wenet.transducer.transducer`
also spares the reference's module-level `import torchaudio` (transducer.py:4), which is the only
reason the reference needs torchaudio on the training path.

`uninstall()` restores whatever was registered before (for tests).
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Dict, Optional

# reference module name -> (our submodule, names the reference's importers expect there)
_TARGETS = {
    "wenet.transducer.transducer": ("transducer", ("Transducer",)),
    "wenet.transducer.joint": ("joint", ("TransducerJoint",)),
    "wenet.transducer.predictor": ("predictor", ("RNNPredictor", "EmbeddingPredictor", "ConvPredictor",
                                                 "PredictorBase")),
    "wenet.transformer.ctc": ("ctc", ("CTC",)),
    "wenet.transducer.search.greedy_search": ("search.greedy_search", ("basic_greedy_search",
                                                                       "basic_greedy_search_both", "edit_distance")),
    "wenet.transducer.search.prefix_beam_search": ("search.prefix_beam_search", ("PrefixBeamSearch", "Sequence")),
}
_saved: Dict[str, Optional[types.ModuleType]] = {}


def _ensure_parent_packages(name: str) -> None:
    """Make `wenet`, `wenet.transducer`, ... importable names.  With the reference on sys.path its real packages
    are imported (they are plain directories / empty __init__ files); without it, empty namespace stand-ins are
    registered so that `from wenet.transducer.joint import X` resolves from sys.modules alone."""
    parts = name.split(".")[:-1]
    for i in range(1, len(parts) + 1):
        pkg = ".".join(parts[:i])
        if pkg in sys.modules:
            continue
        try:
            importlib.import_module(pkg)
        except Exception:
            m = types.ModuleType(pkg)
            m.__path__ = []                      # marks it as a package
            _saved.setdefault(pkg, None)
            sys.modules[pkg] = m


def install() -> None:
    """Register the replacements.  Must run before `wenet.utils.init_model` (or anything that imports it) is
    imported; raises if that already happened with the reference's own classes bound."""
    pkg = importlib.import_module("wenet_celoss_amd")
    im = sys.modules.get("wenet.utils.init_model")
    if im is not None and getattr(getattr(im, "Transducer", None), "__module__", "").startswith("wenet."):
        raise RuntimeError("wenet_celoss_amd.patch.install(): wenet.utils.init_model was imported before install(); "
                           "call install() first (it binds its classes at import time, init_model.py:16-22)")
    for ref_name, (sub, names) in _TARGETS.items():
        ours = importlib.import_module(f"{pkg.__name__}.{sub}")
        _ensure_parent_packages(ref_name)
        shim = types.ModuleType(ref_name)
        shim.__doc__ = f"wenet_celoss_amd replacement for {ref_name} (installed by wenet_celoss_amd.patch)"
        shim.__wr_replacement__ = True
        for k, v in vars(ours).items():
            if not k.startswith("__"):
                setattr(shim, k, v)
        for n in names:
            if not hasattr(shim, n):
                setattr(shim, n, _unavailable(ref_name, n))
        _saved.setdefault(ref_name, sys.modules.get(ref_name))
        sys.modules[ref_name] = shim
        parent = sys.modules.get(ref_name.rsplit(".", 1)[0])
        if parent is not None:
            setattr(parent, ref_name.rsplit(".", 1)[1], shim)


def _unavailable(module: str, name: str):
    """Placeholder for a name the reference's importers expect in a replaced module but this package does not define
    (none at present: all three predictors, the joiner, CTC, Transducer and the search entry points exist); it refuses
    at construction instead of failing the import."""
    class _Unavailable:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{module}.{name} is not implemented by wenet_celoss_amd")
    _Unavailable.__name__ = name
    return _Unavailable


def installed() -> bool:
    m = sys.modules.get("wenet.transducer.transducer")
    return bool(getattr(m, "__wr_replacement__", False))


def uninstall() -> None:
    for name, old in list(_saved.items()):
        if old is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = old
    _saved.clear()
