"""sosrt -- MI355X-native Successive-Orders-of-Scattering hot path.

Host-side mirror of the reference's function surface for the path
I1 -> [Jn -> In] x orders:

  sosrt.I1_In      I1_NumInt, Jn_NumInt, In_NumInt            (SOS_Aer_I1_In.py)
  sosrt.In_limit   mu -> 0 helpers                              (SOS_Aer_In_limit.py)
  sosrt.global_va  thresholds                                   (SOS_Aer_global_va.py)
  sosrt.main       SOS_Aer(**overrides), SOS_Aer_batch(...)     (SOS_Aer_main_*.py)
  sosrt.inputs     tau_profile, direction grid, phase functions (inputs of the path)
  sosrt.solver     handle-level API over the C ABI (include/sosrt.h)
  sosrt.dist       column sharding over the GPUs of a node + RCCL gather

All computation runs in hand-written HIP kernels behind libsosrt.so; importing
this package does not load the library, the first call that computes does.
"""
from .global_va import MU_THRESHOLD, MU_EXTREME_THRESHOLD, MU_VERY_SMALL_THRESHOLD  # noqa: F401

__all__ = ["I1_In", "In_limit", "global_va", "main", "inputs", "solver", "dist"]
