"""optim::lr_scheduler (host-only scalar logic, no GPU): the five policies of neuronika-optim/src/lr_scheduler/
stepped against their definitions in f32 — prepare_step (last <- current, epoch += 1; mod.rs:51-59), then the policy."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def O():
    import neuronika_amd
    return neuronika_amd.tape.optim


f32 = np.float32


def test_step_and_multistep(O):
    opt = O.SGD(0.5)
    s = O.lr_scheduler.StepLR(opt, 3, 0.1)
    want = f32(0.5)
    for epoch in range(1, 11):
        s.step()
        last = want
        if epoch % 3 == 0:                                    # step_lr/mod.rs:61-64
            want = f32(want * f32(0.1))
        assert s.get_current_epoch() == epoch and f32(s.get_last_lr()) == last
        assert f32(s.get_current_lr()) == want and f32(opt.get_lr()) == want
    opt = O.SGD(1.0)
    m = O.lr_scheduler.MultiStepLR(opt, [2, 5], 0.5)
    seen = []
    for _ in range(6):
        m.step(); seen.append(opt.get_lr())
    assert seen == [1.0, 0.5, 0.5, 0.5, 0.25, 0.25]           # multi_step_lr/mod.rs:58-66
    m.set_current_epoch(1); m.step()                          # epoch 2 again -> another decay
    assert opt.get_lr() == 0.125


def test_exponential_lambda_multiplicative(O):
    opt = O.Adam(0.01)
    e = O.lr_scheduler.ExponentialLR(opt, 0.9)
    want = f32(0.01)
    for _ in range(5):
        e.step(); want = f32(want * f32(0.9))
        assert f32(opt.get_lr()) == want
    opt = O.SGD(2.0)
    lam = O.lr_scheduler.LambdaLR(opt, lambda epoch: 1.0 / (1 + epoch))      # lr = initial * f(epoch)
    for epoch in range(1, 5):
        lam.step()
        assert f32(opt.get_lr()) == f32(f32(2.0) * f32(1.0 / (1 + epoch)))
    opt = O.SGD(2.0)
    mul = O.lr_scheduler.MultiplicativeLR(opt, lambda epoch: 0.5)            # lr = last * f(epoch)
    for epoch in range(1, 5):
        mul.step()
        assert opt.get_lr() == 2.0 * 0.5 ** epoch and mul.get_last_lr() == 2.0 * 0.5 ** (epoch - 1)


def test_reference_scenarios(O):
    """The scenarios of neuronika-optim/src/lr_scheduler/*/test.rs (5 epochs from lr = 1.0): StepLR(1, 2) -> 2^e,
    ExponentialLR(5) -> 5^e, MultiStepLR([1,2,3,4], 2) -> 16, LambdaLR(e) -> e, MultiplicativeLR(e) -> 5! = 120."""
    ls = O.lr_scheduler

    def run(make):
        opt = O.SGD(1.0, l2=0.1)
        s = make(opt)
        s.set_current_epoch(5); assert s.get_current_epoch() == 5
        s.set_current_epoch(0); assert s.get_current_epoch() == 0
        cur = []
        for epoch in range(5):
            cur.append(s.get_current_lr())
            assert s.get_current_epoch() == epoch
            s.step()
        return s, cur

    s, cur = run(lambda o: ls.StepLR(o, 1, 2.0))
    assert cur == [2.0 ** e for e in range(5)] and s.get_last_lr() == 16.0
    s, _ = run(lambda o: ls.ExponentialLR(o, 5.0))
    assert s.get_last_lr() == 5.0 ** 4 and s.get_current_lr() == 5.0 ** 5
    s, _ = run(lambda o: ls.MultiStepLR(o, [1, 2, 3, 4], 2.0))
    assert s.get_last_lr() == 16.0 and s.get_current_lr() == 16.0
    s, cur = run(lambda o: ls.LambdaLR(o, lambda e: float(e)))
    assert cur[1:] == [1.0, 2.0, 3.0, 4.0] and s.get_last_lr() == 4.0
    s, _ = run(lambda o: ls.MultiplicativeLR(o, lambda e: float(e)))
    assert s.get_last_lr() == 24.0 and s.get_current_lr() == 120.0
