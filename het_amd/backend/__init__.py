"""Autograd boundary of the hot path: ``torch.autograd.Function`` classes and
output-allocating wrappers with the names and signatures of the reference's
``hrt/python/backend`` package (rgnn/rgat/rgcn/hgt ``*_layers_and_funcs.py``)."""
from .rgnn_layers_and_funcs import *  # noqa: F401,F403
from .rgat_layers_and_funcs import *  # noqa: F401,F403
from .rgcn_layers_and_funcs import *  # noqa: F401,F403
from .hgt_layers_and_funcs import *  # noqa: F401,F403


def plan_enabled() -> bool:
    """Whether the cached device-side groupings (het_amd/plan.py) are in use (the fast paths of the ops)."""
    from .. import plan as _plan
    return bool(_plan.is_enabled())
