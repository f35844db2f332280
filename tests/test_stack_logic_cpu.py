"""Bookkeeping of the per-layer spectrum stacks (integral_operators._stack_take and friends: the weight gradient of a layer that a
roll-out uses several times per graph is batched over its uses, reference ns_train_2d.py:46-68) - host logic only, CPU tensors,
no kernels: how many slots a layer gets, when a stack is closed, the 2 GiB bound of the per-mode GEMM's operand offsets."""
import pytest
import torch

import uno_amd.integral_operators as io


def _leaf(ci=4, co=6, m=3):
    return torch.nn.Parameter(torch.zeros(ci, co, m, m, dtype=torch.cfloat))


def test_no_stack_until_a_pass_used_the_layer_twice():
    w = _leaf()
    shape = (2, 4, 6, 3)
    assert io._stack_take(w, shape, w.device, True) is None            # no hint yet
    w._uno_uses = 1
    assert io._stack_take(w, shape, w.device, True) is None            # one use per pass: nothing to batch
    w._uno_uses = 3
    assert io._stack_take(w, shape, w.device, False) is None           # gradient not wanted
    st, slot = io._stack_take(w, shape, w.device, True)
    assert slot == 0
    assert tuple(st.X.shape) == (3, *shape) and st.X.dtype == torch.complex64
    assert [io._stack_take(w, shape, w.device, True)[1] for _ in range(2)] == [1, 2]
    st2, slot2 = io._stack_take(w, shape, w.device, True)              # a fourth use: the stack is full, a second one starts
    assert st2 is not st and slot2 == 0 and st.n == 3


def test_a_stack_closes_when_a_backward_touched_it_or_the_weights_changed():
    w = _leaf()
    w._uno_uses = 4
    shape = (2, 4, 6, 3)
    st, _ = io._stack_take(w, shape, w.device, True)
    st.sealed = True                                                   # what the first backward call of a pass does
    st2, slot = io._stack_take(w, shape, w.device, True)
    assert st2 is not st and slot == 0
    with torch.no_grad():
        w.add_(1.0)                                                    # the optimiser step bumps the version counter
    st3, slot = io._stack_take(w, shape, w.device, True)
    assert st3 is not st2 and slot == 0
    st4, slot = io._stack_take(w, (3, 4, 6, 3), w.device, True)        # another batch size: another stack
    assert st4 is not st3 and slot == 0
    w._uno_nostack = True
    assert io._stack_take(w, shape, w.device, True) is None


def test_stacks_stay_below_the_32_bit_operand_offsets_of_the_mode_gemm():
    w = torch.nn.Parameter(torch.zeros(8, 256, 1, 1, dtype=torch.cfloat))         # Co = 256 output channels decide the bound
    w._uno_uses = 1000
    shape = (32, 8, 44, 22)                                            # per slot: 8 * 32 * max(8, 256) * 44 * 22 bytes = 63 MB
    per_slot = 8 * 32 * 256 * 44 * 22
    cap = (2 ** 31 - 4096) // per_slot
    assert 2 <= cap < 1000
    # (allocating cap slots of the INPUT spectra is 8 * 32 * 8 * 44 * 22 * cap = 65 MB: fine on the host)
    st, _ = io._stack_take(w, shape, w.device, True)
    assert st.X.shape[0] == cap


def test_switches():
    w = _leaf()
    w._uno_uses = 3
    shape = (2, 4, 6, 3)
    for name in ("TIME_BATCHED_WGRAD", "INPLACE_PARAM_GRADS"):
        setattr(io, name, False)
        try:
            assert io._stack_take(w, shape, w.device, True) is None
        finally:
            setattr(io, name, True)
    assert io._stack_take(torch.zeros(4, 6, 3, 3, dtype=torch.cfloat), shape, w.device, True) is None     # not a leaf parameter with a hint


def test_stale_backward_pass_entries_are_swept(monkeypatch):
    """a backward pass that raises never runs its final callback; its _PASSES entry must not live for ever."""
    import time
    from uno_amd import integral_operators as io
    io._PASSES.clear()
    io._PASSES[123456] = {"id": 123456, "acc": {}, "stacks": {}, "uses": {}, "born": time.monotonic() - 2 * io._STALE_PASS_SECONDS}
    io._PASSES[123457] = {"id": 123457, "acc": {}, "stacks": {}, "uses": {}, "born": time.monotonic()}
    assert io._pass_state() is None            # outside a pass: sweeps
    assert 123456 not in io._PASSES and 123457 in io._PASSES
    io._PASSES.clear()


def test_a_swept_pass_that_shows_up_again_raises(monkeypatch):
    """wall-clock age only guesses that a pass is dead; a live pass whose state was swept must fail loudly at its next
    contribution instead of overwriting the gradient it had been summing (beta = 0 write into the registered buffer)."""
    import time
    from uno_amd import integral_operators as io
    io._PASSES.clear()
    io._SWEPT.clear()
    io._PASSES[777] = {"id": 777, "acc": {}, "stacks": {}, "uses": {}, "born": time.monotonic() - 2 * io._STALE_PASS_SECONDS}
    assert io._pass_state() is None and 777 in io._SWEPT
    monkeypatch.setattr(io, "_current_graph_task_id", lambda: 777)
    with pytest.raises(RuntimeError, match="was idle for more than"):
        io._pass_state()
    io._SWEPT.clear()


def test_complex_adam_loads_a_capturable_checkpoint_into_a_plain_optimiser():
    """the step entries of a capturable state are one shared device tensor; a non-capturable ComplexAdam that loads them must
    count in host integers (one increment per step, not one per parameter)."""
    from uno_amd.harness import ComplexAdam
    ps = [torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(3, dtype=torch.cfloat))]
    src = ComplexAdam(ps, lr=1e-3)
    for p in ps:
        p.grad = torch.randn_like(p)
    src.step()
    sd = src.state_dict()
    shared = torch.tensor([4], dtype=torch.int32)
    for st in sd["state"].values():
        st["step"] = shared                          # what a capturable optimiser's state_dict carries
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    dst = ComplexAdam(qs, lr=1e-3)
    dst.load_state_dict(sd)
    assert all(isinstance(st["step"], int) and st["step"] == 4 for st in dst.state.values())
    for q in qs:
        q.grad = torch.randn_like(q)
    dst.step()
    assert all(st["step"] == 5 for st in dst.state.values()) and int(shared) == 4


def test_private_autograd_entry_points_are_optional(monkeypatch):
    """without torch._C._current_graph_task_id / the engine's callback queue the library must take the ordinary
    gradient path (no pass state), not fail."""
    from uno_amd import integral_operators as io
    assert io._PASS_STATE_AVAILABLE                  # this torch has both
    monkeypatch.setattr(io, "_PASS_STATE_AVAILABLE", False)
    assert io._graph_task_id() == -1 and io._pass_state() is None
    p = torch.nn.Parameter(torch.zeros(3))
    assert io._grad_plan(p, None) is None and io._grad_targets([p]) is None
