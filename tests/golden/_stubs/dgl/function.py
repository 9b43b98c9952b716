"""Builtin message/reduce functions of the DGL stand-in (see package docstring)."""


class _Builtin:
    def __init__(self, op, msg, out):
        self.op, self.msg, self.out = op, msg, out


def mean(msg, out):
    return _Builtin('mean', msg, out)


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return _Builtin('sum', msg, out)


class _CopyU:
    def __init__(self, u, out):
        self.u, self.out = u, out

    def __call__(self, edges):
        return {self.out: edges.src[self.u]}


def copy_u(u, out):
    return _CopyU(u, out)


copy_src = copy_u
