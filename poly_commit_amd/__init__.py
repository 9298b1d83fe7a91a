"""MI355X (gfx950) backend for the commit/open hot path of arkworks-rs/poly-commit.

Python is only the harness language here (tests, bench, torch.distributed plumbing); the
product is the C-ABI library ``libpc_hip.so`` (include/pc_hip.h) built from ``csrc/``.
"""
from ._ffi import (  # noqa: F401
    CURVES, Context, Group, GroupSrs, PcHipError, Srs, library_path, load_library, point_mul, points_sum, universal_params_layout,
)
