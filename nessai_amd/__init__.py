"""nessai_amd — MI355X-native normalising-flow proposal path for nested sampling.

A from-scratch gfx950 implementation of the hot path of ``mj-will/nessai``:
the RealNVP / neural-spline coupling flows, ``FlowModel`` and
``FlowProposal.populate``, behind the reference's own plug-in surface
(SURVEY.md §8b).  The arithmetic runs in hand-written HIP kernels
(``nessai_amd/csrc``) reached through a C ABI (``include/nessai_amd.h``).
"""

__version__ = "0.1.0"
