"""Training loop of ``FlowModel.train`` (placeholder until the backward kernels land)."""


def train_flow(flow_model, samples, max_epochs=None, patience=None, val_size=None):
    raise NotImplementedError(
        "nessai_amd training kernels are not built yet (FlowModel.train, "
        "src/nessai/flowmodel/base.py:530-696); load trained weights with "
        "FlowModel.load_weights / model.load_state_dict"
    )
