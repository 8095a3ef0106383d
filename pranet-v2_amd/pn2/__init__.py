"""pn2 — MI355X-native engine behind the PraNet-V2 nn.Module surface (see ../lib)."""
from .capi import F32, BF16, LIB_PATH, load as load_library      # noqa: F401
from .graph import set_compute_dtype, get_compute_dtype, run_module  # noqa: F401
