"""GPU: the env-side controllers of the two Bullet robots as batched device kernels (csrc/robot_ctrl.h) against the
float64 restatement in oracle/ctrl_oracle.py (reference: robots/turtlebot3.py:214-238, robots/drone.py:58-159,175-193)."""
import numpy as np
import pytest

from oracle import ctrl_oracle as CO

pytestmark = pytest.mark.gpu


def _engine():
    from mobrob_amd.engine import PPOEngine
    return PPOEngine(obs_dim=12, act_dim=18, n_envs=4, n_steps=4, batch_size=8, n_epochs=1)


def test_turtlebot3_proportional_controller_matches_the_restatement():
    e = _engine()
    rng = np.random.default_rng(0)
    n = 5000
    pos, goal = rng.uniform(-0.8, 0.8, (n, 2)), rng.uniform(-0.8, 0.8, (n, 2))
    theta = rng.uniform(-np.pi, np.pi, n)
    act = rng.uniform(-1, 1, (n, 2))
    goal[:50] = pos[:50] + [[0.3, 0.0]]            # on the axis: np.sign(0) = 0 -> bearing 0
    goal[50:60] = pos[50:60]                        # on the goal: distance 0
    theta[60:80] = np.pi - 1e-3; goal[60:80] = pos[60:80] + [[-0.5, -1e-3]]   # heading error wraps around +-pi
    got = e.ctrl_turtlebot3(pos, theta, goal, act)
    ref = CO.turtlebot3_prop_ctrl(pos.astype(np.float32), theta.astype(np.float32), goal.astype(np.float32), act.astype(np.float32))
    assert got.shape == (n, 2) and np.max(np.abs(got - ref)) < 2e-5
    assert np.all(np.abs(got[:, 0]) <= 0.26 + 1e-7) and np.all(np.abs(got[:, 1]) <= 1.82 + 1e-6)
    assert (np.abs(got[:, 0]) == np.float32(0.26)).any() and (np.abs(got[:, 1]) < 1.82).any()   # both regimes occur
    assert (got[:, 0] < 0).any()                    # negative distance gain: the robot may reverse (gain radius 1.5 > mean 1)
    with pytest.raises(ValueError):
        e.ctrl_turtlebot3(pos[:, :1], theta, goal, act)
    e.close()


def test_drone_cascaded_pid_matches_the_restatement_over_a_trajectory():
    """80 controller periods with carried integrator / derivative state, gains re-tuned by a fresh action every step
    (what DroneEnv.step does), on a toy point-mass response so that errors evolve."""
    e = _engine()
    rng = np.random.default_rng(1)
    n, mass = 512, 0.5
    prm = dict(mass=mass, max_thrust=4 * mass * 9.8 * 0.9, max_xy_torque=0.05, max_z_torque=0.01)
    pos = rng.uniform([-3, -3, 1], [3, 3, 3], (n, 3)).astype(np.float32)
    rpy = rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32)
    rpy[:8, 2] = 3.1                                # yaw error beyond pi: wraps
    goal = rng.uniform([-5, -5, 0], [5, 5, 5], (n, 3)).astype(np.float32)
    st_dev = np.zeros((n, 12), np.float32)
    st_ref = np.zeros((n, 12), np.float64)
    vel = np.zeros((n, 3))
    sat_thrust = sat_torque = 0
    for t in range(80):
        act = rng.uniform(-1, 1, (n, 18)).astype(np.float32)
        got = e.ctrl_drone_pid(pos, rpy, goal, act, st_dev, **prm)
        ref = CO.drone_pid(pos, rpy, goal, act, st_ref, **prm)
        scale = np.maximum(1.0, np.abs(ref))
        assert np.max(np.abs(got - ref) / scale) < 2e-4, (t, float(np.max(np.abs(got - ref) / scale)))
        assert np.max(np.abs(st_dev - st_ref)) < 1e-3 * max(1.0, float(np.abs(st_ref).max()))
        st_ref[:] = st_dev                          # follow the device's state: compare one period at a time
        sat_thrust += int((got[:, 0] >= prm["max_thrust"] - 1e-6).sum() + (got[:, 0] <= 0).sum())
        sat_torque += int((np.abs(got[:, 1]) >= prm["max_xy_torque"] - 1e-9).sum())
        # toy response: thrust lifts along z, attitude relaxes towards the commanded torque direction
        vel[:, 2] += (got[:, 0] / mass - 9.8) / 50
        vel[:, :2] += np.stack([np.sin(rpy[:, 1]), -np.sin(rpy[:, 0])], 1) * 9.8 / 50
        pos = (pos + vel / 50).astype(np.float32)
        rpy = (rpy + 2.0 * got[:, 1:] / 50).astype(np.float32)
    assert sat_thrust > 0 and sat_torque > 0        # the limiters were exercised
    e.close()
