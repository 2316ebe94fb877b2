"""Entry-point target for the reference's ``nessai.proposals`` plug-in group.

The reference resolves ``flow_proposal_class="<name>"`` through that entry-point group and accepts the
class only if it subclasses ITS ``FlowProposal`` (``src/nessai/proposal/utils.py:97-155``).  Importing this
module therefore needs the reference installed; it builds that subclass once
(``nessai_amd.adapter.as_reference_proposal``) and exports it::

    [project.entry-points."nessai.proposals"]
    mi355x = "nessai_amd.plugin:MI355XFlowProposal"
"""

try:
    from nessai.proposal import FlowProposal as _ReferenceFlowProposal
except ImportError as exc:  # pragma: no cover - the reference is not installed on the GPU box
    raise ImportError(
        "nessai_amd.plugin exports the proposal class for nessai's `nessai.proposals` entry-point group and "
        "needs `nessai` to be importable; without the reference use nessai_amd.proposal.FlowProposal directly"
    ) from exc

from .adapter import as_reference_proposal

MI355XFlowProposal = as_reference_proposal(_ReferenceFlowProposal)

__all__ = ["MI355XFlowProposal"]
