"""TensorBoard tag names (reference gops/utils/tensorboard_setup.py:154-168) and launch helpers.

The tag strings are part of the interface (trainers and result parsers key on them).  The
helpers degrade to no-ops when the `tensorboard` package is not installed.
"""
tb_tags = {
    "TAR of RL iteration": "Evaluation/1. TAR-RL iter",
    "TAR of total time": "Evaluation/2. TAR-Total time [s]",
    "TAR of collected samples": "Evaluation/3. TAR-Collected samples",
    "TAR of replay samples": "Evaluation/4. TAR-Replay samples",
    "Buffer RAM of RL iteration": "RAM/RAM [MB]-RL iter",
    "loss_actor": "Loss/Actor loss-RL iter",
    "loss_actor_reward": "Loss/Actor reward loss-RL iter",
    "loss_actor_constraint": "Loss/Actor constraint loss-RL iter",
    "loss_critic": "Loss/Critic loss-RL iter",
    "alg_time": "Time/Algorithm time [ms]-RL iter",
    "sampler_time": "Time/Sampler time [ms]-RL iter",
    "critic_avg_value": "Train/Critic avg value-RL iter",
    "lips_value": "Lipschitz/Lipschitz value - RL iter",
}


def add_scalars(tb_info: dict, writer, step: int):
    """Values may be 0-dim device tensors (the data-parallel path leaves its losses on the GPU so that no
    host sync sits between the backward pass and the gradient all-reduce): they are read here, at log time."""
    if writer is None:
        return
    for key, value in tb_info.items():
        writer.add_scalar(key, float(value), step)


def make_writer(log_dir: str):
    """SummaryWriter if tensorboard is importable, else None (logging is then skipped)."""
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir, flush_secs=20)
    except Exception:
        return None


def start_tensorboard(logdir, port=6006):
    try:
        import tensorboard  # noqa: F401
    except Exception:
        print("tensorboard is not installed: skipping start_tensorboard")
        return
    import os
    os.system("tensorboard --logdir {} --port {} &".format(logdir, port))


def save_tb_to_csv(path):
    try:
        from tensorboard.backend.event_processing import event_accumulator  # noqa: F401
    except Exception:
        print("tensorboard is not installed: skipping save_tb_to_csv")
        return
