"""Plug-in registry with the reference's interface (main/util.py:10,33-62).

``register_module(category, name)`` / ``get_module(category, name)`` behave like the
reference's: classes are stored in a module-global dict keyed by category then name, a missing
entry raises ``ValueError``.  The MI355X classes register under the SAME keys the reference
uses (``score_fn/ncsnpp``, ``sde/psld``, ``losses/psld_score_loss``, ``samplers/em_sde``,
``pl_modules/sde_wrapper``) so that ``train_sde.py:36-60`` / ``eval/sample.py:38-69`` resolve
them unchanged; ``install_into(util)`` additionally overwrites the reference's own registry
entries when both code bases live in one process.
"""
from __future__ import annotations

_MODULES = {}


def register_module(category=None, name=None):
    def _register(cls):
        cat = category if category is not None else (cls.__name__ if name is None else name)
        local_name = cls.__name__ if name is None else name
        bucket = _MODULES.setdefault(cat, {})
        if local_name in bucket and bucket[local_name] is not cls:
            raise ValueError(f"Already registered module with name: {local_name} in category: {cat}")
        bucket[local_name] = cls
        return cls

    return _register


def get_module(category, name):
    module = _MODULES.get(category, dict()).get(name, None)
    if module is None:
        raise ValueError(f"No module named `{name}` found in category: `{category}`")
    return module


def install_into(ref_util) -> None:
    """Overwrite the reference registry (``util._MODULES``) with the MI355X implementations."""
    for cat, bucket in _MODULES.items():
        ref_util._MODULES.setdefault(cat, {}).update(bucket)
