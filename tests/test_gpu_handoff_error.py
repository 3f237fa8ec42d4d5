"""A hand-off time-out inside an update that was queued without a host wait (the default, device-resident learner): the
sticky error words of k_disc_step4 / k_ppo_pair travel in the results ring, sg_results_fetch returns the error, clears the
words and the objects run their multi-launch forms from then on (the words used to be read on the
synchronous path only, so every later step skipped its wait and trained on garbage without a word)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W = dict(kind="split", T=32, N=64, O=14, A=7, F=25, H=100, feet=1, Hd=100, E_p=2, M=4, E_d=2, B=128, Ne=1024, clip=0.2)
SLOTS = ["disc_chain", "disc_wgrad", "ppo_fwd", "ppo_bwd", "ppo_reduce", "relabel_fwd", "ppo_adam", "disc_step"]


def _launches(ctx, learner):
    ctx.profile_reset()
    ctx.profile(True)
    learner.update().resolve()
    ctx.profile(False)
    return {name: ctx.profile_read(i)[1] for i, name in enumerate(SLOTS)}


def test_queued_update_reports_a_hand_off_time_out_and_falls_back(monkeypatch):
    # the one-launch forms whatever else this pytest process still holds on the device (idle contexts of earlier tests would
    # otherwise make the library choose the multi-launch forms, and there would be nothing to test)
    monkeypatch.setenv("SG_DISC_FUSED", "1")
    monkeypatch.setenv("SG_PPO_PAIR", "1")
    import bench
    import simgan_amd as sg
    from simgan_amd import _lib
    ctx = _lib.Context.default()
    pol, disc, agent, ro, loader, expert, learner = bench.build_problem(sg, W, seed=3)
    _lib.check(ctx.lib.sg_rollout_fill_synthetic(ro.h, pol.h, 99, 0.02))
    first = learner.update().resolve()
    assert all(np.isfinite(v) for v in first.values())
    n = _launches(ctx, learner)
    if not n["disc_step"]:
        pytest.skip("this context does not run the one-launch discriminator step (device shared with other learner contexts)")
    pair = n["ppo_fwd"] == 0           # SplitPolicy without a forward launch: k_ppo_pair
    dp, (dm, dv, dt) = disc.get_flat_params(), disc.get_adam()
    pp, (pm, pv, pt) = pol.get_flat_params(), agent.get_adam()
    tl = _lib.load_test()
    _lib.check_test(tl.sg_test_raise_handoff_error(disc.h, agent.h if pair else None))
    pending = learner.update()         # queued: nothing has been read yet
    with pytest.raises(_lib.SimganHipError, match="k_disc_step4"):
        pending.resolve()
    # the words are cleared and both objects have left the one-launch forms
    disc.set_flat_params(dp); disc.set_adam(dm, dv, dt)
    pol.set_flat_params(pp); agent.set_adam(pm, pv, pt)
    after = learner.update().resolve()
    assert all(np.isfinite(v) for v in after.values()), after
    n2 = _launches(ctx, learner)
    assert n2["disc_step"] == 0 and n2["disc_chain"] > 0, n2
    if pair:
        assert n2["ppo_fwd"] > 0, n2
