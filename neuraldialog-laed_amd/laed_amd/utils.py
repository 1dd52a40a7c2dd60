"""Host plumbing shared with the reference's `laed.utils` (Pack, cast_type, dtype tags, str2bool).

Reference: laed/utils.py:17-55,119-142.  Tokenisers / session-dir helpers are out of scope.
"""
import torch

INT = 0
LONG = 1
FLOAT = 2


class Pack(dict):
    """Attribute-style dict (laed/utils.py:22-37)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def add(self, **kwargs):
        for k, v in kwargs.items():
            self[k] = v

    def copy(self):
        pack = Pack()
        for k, v in self.items():
            pack[k] = list(v) if type(v) is list else v
        return pack


def str2bool(v):
    return v.lower() in ('true', '1')


_DTYPES = {INT: torch.int32, LONG: torch.int64, FLOAT: torch.float32}


def cast_type(var, dtype, use_gpu):
    """laed/utils.py:123-142: cast to Int/Long/Float on the GPU (use_gpu) or the CPU."""
    if dtype not in _DTYPES:
        raise ValueError("Unknown dtype")
    var = var.to(_DTYPES[dtype])
    return var.cuda(non_blocking=True) if use_gpu else var.cpu()
