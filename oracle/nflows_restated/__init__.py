"""CPU restatement of the ``glasflow.nflows`` pieces the reference imports.

Test infrastructure only — see ``oracle/__init__.py`` for who may import this
and for the "parity unpinned" statement.
"""

from . import distributions, nets, splines, transforms, utils  # noqa: F401
