"""CPU restatement of the ``glasflow.nflows`` pieces the reference imports.

Test infrastructure only — see ``oracle/__init__.py`` for who may import this
and for the "parity unpinned" statement.

Attribution: the algorithms restated in this package are those of **nflows** (Conor Durkan, Artur Bekasov, Iain
Murray, George Papamakarios: "nflows: normalizing flows in PyTorch", https://github.com/bayesiains/nflows, MIT
License, Copyright (c) 2020 Conor Durkan, Artur Bekasov, Iain Murray, George Papamakarios), which ``glasflow`` vendors
as ``glasflow.nflows`` and the reference imports.  Structure, argument names and operation order follow the public
nflows sources closely on purpose - a checker has to compute what the reference computes.  The MIT permission notice
is reproduced in ``oracle/nflows_restated/LICENSE.nflows``.
"""

from . import distributions, nets, splines, transforms, utils  # noqa: F401
