"""psld_amd — MI355X-native hot path of PSLD (mandt-lab/PSLD): NCSN++ score-network training and
reverse-SDE sampling behind the reference's plug-in surface.  See DESIGN.md / INTEGRATION.md."""
from . import config  # noqa: F401
from .registry import get_module, register_module, install_into  # noqa: F401


def import_modules_into_registry():
    """Counterpart of main/util.py:116-121: importing the plug-ins registers them."""
    from . import sde, vpsde, score_fn, losses, samplers, wrapper, datasets  # noqa: F401
