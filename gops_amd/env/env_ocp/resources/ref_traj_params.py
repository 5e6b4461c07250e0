"""Parameters of the reference trajectories of the veh3dofconti / veh2dofconti families.

The reference builds `MultiRefTrajModel(path_para, u_para)` (gops/env/env_ocp/resources/ref_traj_model.py:26-52): the
caller's dicts update the default parameter set (ref_traj_data.py:18-37) per path / speed profile.  The arithmetic of the
trajectories lives in csrc/aux_kernels.hip (`ref_point`); what crosses the C ABI is `GopsEnv.ref_c`, the table of
constants this module folds - in double, at exactly the places where the reference multiplies Python scalars before
they meet an fp32 tensor (e.g. `-self.A / self.omega * torch.cos(...)`, ref_traj_model.py:119-124).
"""
import math
from copy import deepcopy
from typing import Dict, List, Optional

DEFAULT_PATH_PARAM = {
    "sine": {"A": 1.5, "omega": 2 * math.pi / 10, "phi": 0.0},
    "double_lane": {"t1": 5.0, "t2": 9.0, "t3": 14.0, "t4": 18.0, "y1": 0.0, "y2": 3.5},
    "triangle": {"A": 3.0, "T": 10.0},
    "circle": {"r": 100.0},
}
DEFAULT_SPEED_PARAM = {
    "sine": {"A": 1.0, "omega": 2 * math.pi / 10, "phi": 0.0, "b": 5.0},
    "constant": {"u": 5.0},
}


def merged(path_para: Optional[Dict[str, Dict]], u_para: Optional[Dict[str, Dict]]):
    """(path parameters, speed parameters) after the caller's updates; unknown profile names raise like the
    reference's `self.path_param[k].update(v)` does (KeyError)."""
    path, speed = deepcopy(DEFAULT_PATH_PARAM), deepcopy(DEFAULT_SPEED_PARAM)
    for k, v in (path_para or {}).items():
        path[k].update(v)
    for k, v in (u_para or {}).items():
        speed[k].update(v)
    return path, speed


def ref_constants(path_para: Optional[Dict[str, Dict]] = None, u_para: Optional[Dict[str, Dict]] = None) -> List[float]:
    """The 24 entries of `GopsEnv.ref_c` (include/gops_hip.h) as Python floats (rounded to fp32 by the binding)."""
    path, speed = merged(path_para, u_para)
    us, uc = speed["sine"], speed["constant"]
    ps, dl, tr, ci = path["sine"], path["double_lane"], path["triangle"], path["circle"]
    return [-us["A"] / us["omega"], us["omega"], us["phi"], us["b"], us["A"] / us["omega"] * math.cos(us["phi"]), us["A"],
            uc["u"],
            ps["A"], ps["omega"], ps["phi"],
            dl["t1"], dl["t2"], dl["t3"], dl["t4"], dl["y1"], dl["y2"],
            (dl["y2"] - dl["y1"]) / (dl["t2"] - dl["t1"]), (dl["y1"] - dl["y2"]) / (dl["t4"] - dl["t3"]),
            tr["T"], 2 * tr["A"] / tr["T"], -2 * tr["A"] / tr["T"], tr["T"] / 2,
            ci["r"], 0.0]
