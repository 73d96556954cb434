"""MI355X-native batched-solve backend behind cvxpygen's generate_code / cpg_solve surface."""
from . import cpg  # noqa: F401
