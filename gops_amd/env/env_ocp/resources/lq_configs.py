"""Linear-quadratic problem definitions s2a1 ... s6a3 (same numbers as the reference's
gops/env/env_ocp/resources/lq_configs.py:15-116; they define the benchmark problems)."""


def _cfg(A, B, Q, R, dt, init_mean, init_std, state_lim, action_lim, max_step):
    n, m = len(Q), len(R)
    return dict(A=A, B=B, Q=Q, R=R, dt=dt, init_mean=init_mean, init_std=init_std,
                state_high=[float(state_lim)] * n, state_low=[-float(state_lim)] * n,
                action_high=[float(action_lim)] * m, action_low=[-float(action_lim)] * m,
                max_step=max_step, reward_scale=1.0, reward_shift=0.0)


config_s2a1 = _cfg([[0.0, 1.0], [0.0, 0.0]], [[0.0], [1.0]], [2, 1], [1.0], 0.05,
                   [0.0, 0.0], [1.0, 1.0], 20, 5, 200)
config_s3a1 = _cfg([[-1.01887, 0.90506, -0.00215], [0.82225, -1.07741, -0.17555], [0.0, 0.0, -1.0]],
                   [[0.0], [0.0], [5.0]], [50.0, 1, 1], [1.0], 0.1, [0, 0, 0], [2, 2, 2], 20, 5, 200)
config_s4a2 = _cfg([[0, 1, 0, 0], [0, 1, 0, 0], [0.1, -0.2, 0, 0.5], [-0.2, 0.1, 0.1, 0]],
                   [[0, 0], [-2, -1], [0.0, 0], [1, 1.5]], [1, 2, 2, 1], [1.0, 1.0], 0.1,
                   [0, 0, 0, 0], [0.7, 0.3, 0.7, 0.3], 15, 8, 200)
config_s5a1 = _cfg([[1, 1, 0, 0, 0], [0, 0.2, 1, 0, 0], [0, 0, 0.3, 1, 0], [0, 0, 0, 0.4, 1], [0, 0, 0, 0, 0.5]],
                   [[1], [1], [1], [1], [1]], [50, 10, 20, 10, 10], [100], 0.05, [0] * 5, [0.1] * 5, 50, 10, 500)
config_s6a3 = _cfg([[0, 1, 0, 0, 0, 0], [3, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0], [2.5, 0, 0, 0, 0, 0],
                    [0, 0, 0, 0, 1, 0], [-2, 0, 0, 0, 0, 0]],
                   [[0, 0, 0], [1.5, 1.5, 0], [0.0, 0, 0], [0.5, 0.5, 0.5], [0, 0, 1], [2, 2, 2]],
                   [0, 2, 10, 10, 5, 5], [1.0, 1.0, 1.0], 0.05, [0] * 6, [0.1] * 6, 10, 10, 500)
