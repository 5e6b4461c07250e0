"""Helpers shared by the factories and algorithms (interface of the reference's
gops/utils/common_utils.py: get_activation_func :26-55, get_apprfunc_dict :58-135,
seed_everything/set_seed :186-237, ModuleOnDevice :276-289)."""
import random
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from gops_amd.utils.act_distribution import DiracDistribution

_ACTIVATIONS = {"relu": nn.ReLU, "elu": nn.ELU, "gelu": nn.GELU, "selu": nn.SELU,
                "sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "linear": nn.Identity}


def get_activation_func(key: str):
    assert isinstance(key, str)
    if key not in _ACTIVATIONS:
        print("input activation name:" + key)
        raise RuntimeError
    return _ACTIVATIONS[key]


def get_apprfunc_dict(key: str, **kwargs):
    """Collect the per-network constructor arguments `<key>_func_type`, `<key>_hidden_sizes`, ...
    out of the flat args dict (only the MLP family exists on this path)."""
    var = dict()
    var["apprfunc"] = kwargs[key + "_func_type"]
    var["name"] = kwargs[key + "_func_name"]
    var["obs_dim"] = kwargs["obsv_dim"]
    var["pre_horizon"] = kwargs.get("pre_horizon", None)
    apprfunc_type = kwargs[key + "_func_type"]
    if apprfunc_type != "MLP":
        raise NotImplementedError(f"apprfunc type {apprfunc_type} is outside the MI355X ADP path (MLP only)")
    var["hidden_sizes"] = kwargs[key + "_hidden_sizes"]
    var["hidden_activation"] = kwargs[key + "_hidden_activation"]
    var["output_activation"] = kwargs.get(key + "_output_activation", "linear")
    if kwargs["action_type"] == "continu":
        var["act_high_lim"] = np.array(kwargs["action_high_limit"])
        var["act_low_lim"] = np.array(kwargs["action_low_limit"])
        var["act_dim"] = kwargs["action_dim"]
    else:
        raise NotImplementedError("discrete actions are outside the MI355X ADP path")
    if kwargs.get("policy_act_distribution", "default") != "default":
        raise NotImplementedError("only the default (Dirac) action distribution exists on the ADP path")
    var["action_distribution_cls"] = DiracDistribution
    return var


def make_adam(params, lr: float):
    """Adam with the reference's defaults (gops/algorithm/fhadp.py:45-47): one HIP launch per step
    (`gops_adam_step`) instead of torch's multi-kernel foreach implementation."""
    from gops_amd.hip_backend import HipAdam
    return HipAdam(params, lr=lr)


def seed_everything(seed: Optional[int] = None) -> int:
    if seed is None:
        seed = random.randint(0, 2 ** 32 - 1)
    seed = int(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def set_seed(trainer_name, seed, offset, env=None):
    """Sub-process seeding rule of the reference: only `*_sync_*` / `*_async_*` trainers reseed
    (with seed + offset); serial trainers leave the global RNG alone."""
    if trainer_name.split("_")[1] in ["async", "sync"]:
        print("Setting seed of a subprocess to {}".format(seed + offset))
        seed_everything(seed + offset)
        if env is not None:
            env.seed(seed + offset)
        return seed + offset, env
    if env is not None:
        env.seed(seed)
    return None, env


class ModuleOnDevice:
    """Context manager moving a module to `device` and back to CPU on exit."""

    def __init__(self, module, device):
        self.module, self.device = module, device
        self.different = next(module.parameters()).device.type != device

    def __enter__(self):
        if self.different:
            self.module.to(self.device)

    def __exit__(self, exc_type, exc_val, exc_tb):
        if self.different:
            self.module.to("cpu")
