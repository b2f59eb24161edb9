class Logger:
    def __init__(self, log_path):
        self.log_path = log_path

    def log(self, str_to_log):
        print(str_to_log)
        if self.log_path is not None:
            with open(self.log_path, "a") as f:
                f.write(str_to_log + "\n")
                f.flush()


def L0_norm(x):
    return (x != 0.0).view(x.shape[0], -1).sum(-1)


def _keep(v, x, keepdim):
    return v.view(-1, *[1] * (x.ndim - 1)) if keepdim else v


def L1_norm(x, keepdim=False):
    return _keep(x.abs().view(x.shape[0], -1).sum(-1), x, keepdim)


def L2_norm(x, keepdim=False):
    return _keep((x ** 2).view(x.shape[0], -1).sum(-1).sqrt(), x, keepdim)
