"""Re-export of the build-owned synthetic-weight generator (siu3r_amd/synthetic_weights.py).

The generator is data synthesis (seeded integer hashing -> tensors with the reference's state-dict names and shapes), not part
of the algorithm under test; it lives with the package so that bench.py and the CLI's plumbing mode do not import anything from
oracle/.  The oracle and the tests keep using it through this name."""
from siu3r_amd.synthetic_weights import *  # noqa: F401,F403
from siu3r_amd.synthetic_weights import _hash_uniform, make_tensor, make_weights, param_spec  # noqa: F401
