"""ops.BranchesFn (round 6: independent sub-networks as ONE autograd node whose forward / backward run the branches' own graphs
while the library records their launches, csrc/batch.h) -- the autograd plumbing alone, on CPU tensors with the recorder stubbed:
same outputs and gradients as running the branches one after the other, parameter gradients adopted as .grad, the records that
travel with a tensor (planes, bounds) handed through the node in both directions, the scope closed on every path."""
import contextlib

import pytest
import torch
import torch.nn as nn

from mit_semseg import ops, tuner


class _Lib:
    def __init__(self):
        self.log = []
        self.active = 0

    def semseg_batch_begin(self, n, st):
        assert not self.active
        self.active = 1
        self.log.append(('begin', n))
        return 0

    def semseg_batch_branch(self, i):
        self.log.append(('branch', i))
        return 0

    def semseg_batch_end(self):
        assert self.active
        self.active = 0
        self.log.append(('end',))
        return 0

    def semseg_batch_abort(self):
        self.active = 0
        self.log.append(('abort',))
        return 0

    def semseg_batch_active(self):
        return self.active

    def semseg_batch_next_op(self):
        return 0


class _Stream:
    cuda_stream = 7


@pytest.fixture
def stub(monkeypatch):
    lib = _Lib()
    monkeypatch.setattr(ops._native, 'lib', lambda: lib)
    monkeypatch.setattr(ops._native, '_lib', lib)
    monkeypatch.setattr(ops, '_fork_streams', lambda main, streams: None)
    monkeypatch.setattr(ops, '_join_streams', lambda main, streams: None)
    monkeypatch.setattr(ops, '_branch_streams', lambda dev, n: [_Stream() for _ in range(n)])
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda dev=None: _Stream())
    monkeypatch.setattr(torch.cuda, 'stream', lambda st: contextlib.nullcontext())
    monkeypatch.setattr(ops, '_batchable', lambda fns, args: ops._BATCH[0])
    monkeypatch.setattr(ops, 'CONV_MODE', 'h2')
    return lib


class Branch(nn.Module):
    def __init__(self, c, seen):
        super().__init__()
        self.a = nn.Conv2d(c, c, 3, padding=1)
        self.b = nn.Conv2d(c, c, 1, bias=False)
        self.seen = seen

    def forward(self, x):
        self.seen.append((ops._native.RECORDING[0], tuner.NO_TIMING[0], getattr(x, '_semseg_absmax', None) is not None))
        y = torch.relu(self.a(x)) + x
        y = self.b(y)
        ops.attach_absmax(y, torch.ones(1))
        return y


def _run(batched, seed=0):
    torch.manual_seed(seed)
    seen = []
    fns = [Branch(c, seen) for c in (2, 3, 4)]
    xs = [torch.randn(1, c, 5, 5, requires_grad=True) for c in (2, 3, 4)]
    ins = []
    for x in xs:
        t = x * 1.0
        ops.attach_absmax(t, torch.ones(1))
        ins.append(t)
    with ops.batch_branches(batched):
        ys = ops.run_branches(fns, ins)
    bounds = [ops.bounds_of(y) is not None for y in ys]
    loss = sum((y * y).sum() for y in ys[:2])            # the third branch's output is not used: its gradient is None
    loss.backward()
    return ys, xs, fns, seen, bounds


def test_batched_branches_give_the_gradients_of_the_sequential_run(stub, monkeypatch):
    monkeypatch.setattr(ops, 'BRANCH_STREAMS', False)
    ys0, xs0, f0, seen0, b0 = _run(False)
    assert stub.log == []
    ys1, xs1, f1, seen1, b1 = _run(True)
    assert stub.log == [('begin', 3), ('branch', 0), ('branch', 1), ('branch', 2), ('end',),
                        ('begin', 3), ('branch', 0), ('branch', 1), ('end',)]       # forward; backward (branch 2 got no gradient)
    assert all(s == (True, True, True) for s in seen1) and all(s == (False, False, True) for s in seen0)
    assert b0 == b1 == [True, True, True]                  # the records the last op of a branch left on its output survive the node
    for a, b in zip(ys0, ys1):
        assert torch.equal(a, b)
    for a, b in zip(xs0[:2], xs1[:2]):
        assert torch.equal(a.grad, b.grad)
    assert xs1[2].grad is None and xs0[2].grad is None
    for m0, m1 in zip(f0[:2], f1[:2]):
        for p0, p1 in zip(m0.parameters(), m1.parameters()):
            assert torch.equal(p0.grad, p1.grad)
    assert all(p.grad is None for p in f1[2].parameters())
    assert not ops._native.RECORDING[0] and not tuner.NO_TIMING[0] and not stub.active


def test_a_failing_branch_closes_the_scope(stub):
    class Boom(nn.Module):
        def forward(self, x):
            raise ZeroDivisionError
    xs = [torch.randn(1, 2, 3, 3, requires_grad=True) for _ in range(2)]
    with ops.batch_branches(True), pytest.raises(ZeroDivisionError):
        ops.run_branches([nn.Identity(), Boom()], xs)
    assert stub.log[-1] == ('abort',) and not stub.active and not ops._native.RECORDING[0] and not tuner.NO_TIMING[0]


def test_batch_check_refuses_torch_kernels_inside_a_scope(stub, monkeypatch):
    """SEMSEG_BATCH_CHECK: a torch operator that launches a kernel inside a branch (here: the whole branch is torch arithmetic) would
    run ahead of the recorded launches; the dispatch-mode check raises and the scope is aborted.  Launch-free operators (views,
    allocation) pass."""
    monkeypatch.setattr(ops, 'BATCH_CHECK', True)
    xs = [torch.randn(1, 2, 3, 3, requires_grad=True) for _ in range(2)]
    with ops.batch_branches(True), pytest.raises(RuntimeError, match='side-by-side scope'):
        ops.run_branches([Branch(2, []), Branch(2, [])], xs)
    assert stub.log[-1] == ('abort',) and not stub.active

    class Views(nn.Module):
        def forward(self, x):
            return x.permute(0, 2, 3, 1).detach().permute(0, 3, 1, 2)[:, :1].view_as(x[:, :1])
    with ops.batch_branches(True):
        ys = ops.run_branches([Views(), Views()], xs)
    assert stub.log[-1] == ('end',) and tuple(ys[0].shape) == (1, 1, 3, 3)
