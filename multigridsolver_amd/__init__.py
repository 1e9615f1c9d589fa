"""multigridsolver_amd — MI355X-native aggregation-AMG V-cycle hot path behind the
`solve()` / .mtx surface of mishraiiit/MultiGridSolver.  See DESIGN.md / INTEGRATION.md."""
from ._lib import MgsError, lib, SO_PATH  # noqa: F401
from .core import (Context, Csr, Hierarchy, Vec, Xfer, bicgstab, fgcr, read_mtx, write_mtx,  # noqa: F401
                   OP_JACOBI, OP_RESIDUAL, OP_SPMV)
