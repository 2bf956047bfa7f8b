"""Training-mode executor (train-mode BatchNorm + autograd over the HIP operators)."""


def run_network_train(model, x):
    raise NotImplementedError(
        "training-mode forward/backward on the HIP engine is not wired yet; call model.eval() for inference"
    )
