// Env-side controllers of the two Bullet robots, batched over N robots on the device.
//
// In the reference the RL action of these robots is not a motor command but a correction of controller gains, and the
// controller runs inside `env.step` on the host, one robot at a time:
//   turtlebot3  /root/reference/src/mobrob/envs/pybullet_robots/robots/turtlebot3.py:214-238 `prop_ctrl`
//               (called from Turtlebot3Env.step, src/mobrob/envs/wrapper.py:540-546): gains = means + radius * action,
//               twist = clip((distance, heading error) * gains, +-(0.26 m/s, 1.82 rad/s))
//   drone       robots/drone.py:58-159 `DronePIDController.control` with the gains set by
//               `finetune_force_pid_coef` / `finetune_torque_pid_coef` :175-193 (DroneEnv.step, wrapper.py:481-489:
//               the 18 actions are 6 x 3 gain corrections): position PID -> target force -> thrust along the body
//               axis and a target attitude (roll / pitch limited to 30 degrees), attitude PID -> torques.
// Here they are kernels over [N] arrays (one thread per robot, state of the PID integrators in HBM) behind
// mobrob_ctrl_turtlebot3 / mobrob_ctrl_drone_pid, for a device-resident simulator to call between its physics
// steps.  The motor mixing of the reference (`_compute_rpm`: rpm^2 = A^-1 (B x), NNLS when a rotor saturates) belongs
// to the Bullet actuator model and stays out: the kernels return thrust and torques.  The kinematic stand-in robot of
// this build has no heading or attitude, so the controllers are not in its loop; tests/test_robot_ctrl_gpu.py checks
// them against a float64 NumPy restatement (oracle/ctrl_oracle.py) over many steps with carried integrator state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mobrob {

struct DroneCtrlParams {   // mobrob_drone_params_t of the C ABI
  float mass, g, dt, max_thrust, max_xy_torque, max_z_torque, max_roll_pitch, tune_fac;
};

// twist[n] = (v, w) of robot n
__global__ void k_ctrl_turtlebot3(int n, const float* __restrict__ pos, const float* __restrict__ theta,
                                  const float* __restrict__ goal, const float* __restrict__ gain_changes,
                                  float* __restrict__ twist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float kPi = 3.14159265358979323846f;
  const float k_dist = 1.0f + 1.5f * gain_changes[2 * i];        // prop_gain_means + prop_gain_radius * change
  const float k_ang = 0.2f + 0.5f * gain_changes[2 * i + 1];
  const float gx = goal[2 * i] - pos[2 * i], gy = goal[2 * i + 1] - pos[2 * i + 1];
  const float dist = sqrtf(gx * gx + gy * gy);
  // bearing of the goal: arccos of the x component of the unit goal vector, signed by y (np.sign: 0 on the axis)
  const float c = fminf(fmaxf(gx / (dist + 1e-5f), -1.0f), 1.0f);
  const float sgn = gy > 0.f ? 1.f : (gy < 0.f ? -1.f : 0.f);
  const float bearing = acosf(c) * sgn;
  float err = -(bearing - theta[i]);
  if (err > kPi) err -= 2.0f * kPi;
  else if (err < -kPi) err += 2.0f * kPi;
  twist[2 * i] = fminf(fmaxf(dist * k_dist, -0.26f), 0.26f);
  twist[2 * i + 1] = fminf(fmaxf(err * k_ang, -1.82f), 1.82f);
}

// state[n][12] = last position error | integral of it | last attitude error | integral of it (carried between calls)
// action[n][18] = corrections of (force P, I, D | torque P, I, D), three axes each; out[n][4] = thrust, torque x y z
__global__ void k_ctrl_drone_pid(int n, DroneCtrlParams p, const float* __restrict__ pos, const float* __restrict__ rpy,
                                 const float* __restrict__ goal, const float* __restrict__ action,
                                 float* __restrict__ state, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float kPi = 3.14159265358979323846f;
  const float mean[6][3] = {{0.1f, 0.1f, 0.2f}, {0.0001f, 0.0001f, 0.0001f}, {0.3f, 0.3f, 0.4f},
                            {0.3f, 0.3f, 0.05f}, {0.0001f, 0.0001f, 0.0001f}, {0.3f, 0.3f, 0.5f}};
  float k[6][3];  // gain = mean + correction * (mean * tune_fac): the tuning radius is a fraction of the default gain
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) k[a][j] = mean[a][j] + action[18 * i + 3 * a + j] * (mean[a][j] * p.tune_fac);
  float* st = state + 12 * (size_t)i;
  // ---- position loop: PID on the position error around gravity compensation ----
  float F[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float e = goal[3 * i + j] - pos[3 * i + j];
    const float de = (e - st[j]) / p.dt;
    const float ie = st[3 + j] + e * p.dt;
    st[j] = e;
    st[3 + j] = ie;
    F[j] = (j == 2 ? p.mass * p.g : 0.f) + k[0][j] * e + k[1][j] * ie + k[2][j] * de;
  }
  // ---- thrust: z component of the force turned by the attitude (roll-pitch-yaw, Bullet's fixed-axis convention) ----
  const float r = rpy[3 * i], pt = rpy[3 * i + 1], y = rpy[3 * i + 2];
  const float cr = cosf(r), sr = sinf(r), cp = cosf(pt), sp = sinf(pt), cy = cosf(y), sy = sinf(y);
  (void)cy; (void)sy;  // the third row of Rz Ry Rx does not depend on yaw
  const float tz = -sp * F[0] + cp * sr * F[1] + cp * cr * F[2];
  const float thrust = fminf(fmaxf(tz, 0.f), p.max_thrust);
  // ---- target attitude from the direction of the force; yaw is held at zero ----
  const float sz = F[2] < 0.f ? -1.f : 1.f;  // sign(0) counts as +
  const float nF = sqrtf(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]);
  float tr = asinf(fminf(fmaxf(-sz * F[1] / nF, -1.f), 1.f));
  tr = fminf(fmaxf(tr, -p.max_roll_pitch), p.max_roll_pitch);
  float tp = atan2f(sz * F[0], sz * F[2]);
  tp = fminf(fmaxf(tp, -p.max_roll_pitch), p.max_roll_pitch);
  const float target[3] = {tr, tp, 0.f};
  const float cur[3] = {r, pt, y};
  const float lim[3] = {p.max_xy_torque, p.max_xy_torque, p.max_z_torque};
  // ---- attitude loop ----
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float e = target[j] - cur[j];
    if (j == 2) {
      if (e > kPi) e -= 2.0f * kPi;
      if (e < -kPi) e += 2.0f * kPi;
    }
    const float de = (e - st[6 + j]) / p.dt;
    const float ie = st[9 + j] + e * p.dt;
    st[6 + j] = e;
    st[9 + j] = ie;
    const float tq = k[3][j] * e + k[4][j] * ie + k[5][j] * de;
    out[4 * i + 1 + j] = fminf(fmaxf(tq, -lim[j]), lim[j]);
  }
  out[4 * i] = thrust;
}

}  // namespace mobrob
