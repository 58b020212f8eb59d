"""Small torch helpers with the reference's names (uhc/khrylib/utils/torch.py)."""
import torch

__all__ = ["to_device", "to_test", "to_train", "set_optimizer_lr", "lambda_rule", "batch_to", "get_eta_str"]


def to_device(device, *modules):
    for m in modules:
        m.to(device)


class to_test:
    def __init__(self, *models):
        self.models = [m for m in models if m is not None]
        self.prev = [m.training for m in self.models]
        for m in self.models:
            m.train(False)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        for m, p in zip(self.models, self.prev):
            m.train(p)
        return False


def to_train(*models):
    for m in models:
        if m is not None:
            m.train(True)


def batch_to(dst, *args):
    return [x.to(dst) if x is not None else None for x in args]


def set_optimizer_lr(optimizer, lr):
    for g in optimizer.param_groups:
        g["lr"] = lr


def lambda_rule(epoch, nepoch, nepoch_fix):
    """khrylib/utils/torch.py:165-167: 1 until nepoch_fix, then linear decay to 0 at nepoch."""
    return 1.0 - max(0, epoch - nepoch_fix) / float(nepoch - nepoch_fix + 1)


def get_eta_str(cur_iter, total_iter, time_per_iter):
    eta = time_per_iter * (total_iter - cur_iter - 1)
    return "%02d:%02d:%02d" % (eta // 3600, (eta % 3600) // 60, eta % 60)
