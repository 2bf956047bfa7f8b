def run_network(model, x):
    raise NotImplementedError
