"""mxnet.context of the stub."""


class Context:
    def __init__(self, device_type, device_id=0):
        self.device_type = device_type
        self.device_id = int(device_id)

    def __eq__(self, other):
        return isinstance(other, Context) and (self.device_type, self.device_id) == (other.device_type, other.device_id)

    def __hash__(self):
        return hash((self.device_type, self.device_id))

    def __repr__(self):
        return "%s(%d)" % (self.device_type, self.device_id)

    def torch_device(self):
        import torch
        return torch.device("cpu") if self.device_type == "cpu" else torch.device("cuda", self.device_id)


def cpu(device_id=0):
    return Context("cpu", device_id)


def gpu(device_id=0):
    return Context("gpu", device_id)


def current_context():
    return cpu()
