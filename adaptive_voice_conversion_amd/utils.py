"""Host helpers mirroring utils.py:8-35 of the reference (device placement, infinite iterator, optional logger)."""
import os

import torch


def local_device():
    """One process per GPU: LOCAL_RANK picks the device (the reference's ``cc`` picks 'cuda', utils.py:8-10)."""
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    return torch.device("cpu")


def cc(net):
    return net.to(local_device())


def infinite_iter(iterable):
    """utils.py:28-35."""
    it = iter(iterable)
    while True:
        try:
            yield next(it)
        except StopIteration:
            it = iter(iterable)


class Logger:
    """utils.py:12-20; tensorboardX is optional (not installed in this image)."""

    def __init__(self, logdir="./log"):
        try:
            from tensorboardX import SummaryWriter  # type: ignore
            self.writer = SummaryWriter(logdir)
        except Exception:
            self.writer = None

    def scalars_summary(self, tag, dictionary, step):
        if self.writer is not None:
            self.writer.add_scalars(tag, dictionary, step)

    def scalar_summary(self, tag, value, step):
        if self.writer is not None:
            self.writer.add_scalar(tag, value, step)
