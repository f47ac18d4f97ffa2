"""cggp -- MI355X-native conjugate-gradient sparse-GP hot path.

Drop-in for the CG + kernel-matvec path of `awav/conjugate-gradient-sparse-gp`
(`cggp.conjugate_gradient`, `cggp.models`, `cggp.distance`, `cggp.utils.add_diagonal`): same
names and call signatures on torch tensors that live on the GPU; the arithmetic runs in
`libmgp.so` (hand-written HIP for gfx950, C ABI in include/mgp.h).  There is no CPU fallback.
"""

__version__ = "0.1.0"
