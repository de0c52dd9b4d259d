"""Replay recorded CPU-generator draws (tests/golden/make_golden.py RngTape) into code that calls
torch.rand / randint / randperm / Tensor.uniform_ in the reference's order."""
import torch


class RngReplay:
    def __init__(self, draws):
        """draws: list of (name, tensor) in call order; names in {rand, randint, randperm, uniform_}."""
        self.draws = list(draws)
        self.pos = 0
        self._orig = {}

    def _next(self, name, shape=None):
        assert self.pos < len(self.draws), f"unexpected extra draw {name}"
        n, t = self.draws[self.pos]
        assert n == name, f"draw #{self.pos}: expected {n}, code asked for {name}"
        self.pos += 1
        if shape is not None and t is not None:
            assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), tuple(shape))
        return t

    def __enter__(self):
        self._orig = {"rand": torch.rand, "randint": torch.randint, "randperm": torch.randperm,
                      "uniform_": torch.Tensor.uniform_}
        rep = self

        def rand(*size, **kw):
            size = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
            t = rep._next("rand", size)
            return t.clone() if t is not None else rep._orig["rand"](*size)

        def randint(high, size, **kw):
            t = rep._next("randint", size)
            return t.clone() if t is not None else rep._orig["randint"](high, size)

        def randperm(n, **kw):
            t = rep._next("randperm")
            return t.clone() if t is not None else rep._orig["randperm"](n)

        def uniform_(self_t, a=0.0, b=1.0):
            t = rep._next("uniform_", self_t.shape)
            return self_t.copy_(t) if t is not None else rep._orig["uniform_"](self_t, a, b)

        torch.rand, torch.randint, torch.randperm, torch.Tensor.uniform_ = rand, randint, randperm, uniform_
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randint, torch.randperm = self._orig["rand"], self._orig["randint"], self._orig["randperm"]
        torch.Tensor.uniform_ = self._orig["uniform_"]
        if exc[0] is None:
            assert self.pos == len(self.draws), f"only {self.pos} of {len(self.draws)} recorded draws were consumed"
