"""pailliercryptolib_amd -- MI355X-native batched modular exponentiation for Paillier
(drop-in for the hot path of intel/pailliercryptolib).  The product is the HIP library
``libpgpu.so`` behind the C-ABI ``include/pgpu.h`` and the ``ipcl::`` C++ mirror in
``include/ipcl``; this Python package is the thin host-side binding used by tests and bench.
"""
from . import _capi  # noqa: F401
from .engine import (initialize, terminate, mod_exp, mod_exp_limbs, mod_mul, mod_mul_limbs,  # noqa: F401
                     PublicKey, PrivateKey)
from .limbs import ints_to_limbs, limbs_to_ints  # noqa: F401
