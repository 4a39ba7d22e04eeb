"""Import shim: the package directory is `mini-stark_amd/` (not a valid Python
identifier), so `import mini_stark_amd` resolves here and this module points its
package path at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mini-stark_amd")]
from mini_stark_amd._native import (  # noqa: E402,F401
    Context, MsError, build_library, library_path, load_library,
    GOLDILOCKS, BABYBEAR, FLAG_ZERO_DISPLAY_EMPTY, FLAG_TRACE_MONT64, FLAG_LATENCY,
    OK, ERR_SHAPE, ERR_LEAF_NOT_FOUND, ERR_OUT_OF_RANGE, ERR_STATE, ERR_ARG, ERR_HIP, ERR_NOMEM,
)
