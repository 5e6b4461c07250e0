"""CPU: the strict reference-point provider (gops_amd/env/env_ocp/resources/ref_traj_host.py) against the oracle's step-by-step
restatement of MultiRefTrajModel and against the appended points the UNMODIFIED reference recorded in the step fixtures."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_meta, load_golden
from helpers import data_from_golden
from oracle import adp_oracle as orc

from gops_amd.env.env_ocp.resources.ref_traj_host import HostRefTraj, ReferencePointPipeline
from gops_amd.env.env_ocp.resources.ref_traj_params import ref_constants

CUSTOM = dict(path_para={"sine": {"A": 2.0, "omega": 0.9, "phi": 0.3}, "double_lane": {"t1": 3.0, "t2": 6.5, "y2": 2.5},
                         "triangle": {"A": 2.0, "T": 7.0}, "circle": {"r": 60.0}},
              u_para={"sine": {"A": 0.7, "omega": 0.5, "phi": 0.2, "b": 4.0}, "constant": {"u": 6.5}})


def _same_libm_as_the_fixtures():
    from golden.make_libm_canary import canary
    with open(os.path.join(GOLDEN, "host_libm_canary.json")) as f:
        want = json.load(f)
    got = canary()
    return all(got[k] == want[k] for k in got)


def _oracle_points(ref_time, path_num, u_num, H, P, params=None):
    """The reference's sequence, one step at a time on [B] tensors, all 8 profiles per sample (oracle/adp_oracle.py ref_point)."""
    t, pts = ref_time.clone(), []
    for _ in range(H):
        t = t + 0.1
        if params is None:
            pts.append(orc.ref_point(t + P * 0.1, path_num, u_num))
        else:
            with orc._active_ref({"ref_params": params}):
                pts.append(orc.ref_point(t + P * 0.1, path_num, u_num))
    return torch.stack(pts, 1)


def _bits(t):
    return t.contiguous().view(torch.int32)


def _assert_same_points(got, want, B):
    """Bit equality, tail samples of `torch.atan2`'s loop included (ref_traj_host.py `_scalar_tail`)."""
    assert torch.equal(_bits(got), _bits(want)), float((got - want).abs().max())


@pytest.mark.parametrize("B,H,P", [(4096, 30, 30), (70, 12, 10), (1, 5, 10), (33, 1, 50), (1000, 10, 10)])
def test_provider_is_bit_equal_to_the_step_by_step_restatement(B, H, P):
    g = torch.Generator().manual_seed(B + H)
    rt = 20 * torch.rand(B, generator=g)
    pn = torch.randint(0, 4, (B,), generator=g).float()
    un = torch.randint(0, 2, (B,), generator=g).float()
    got = HostRefTraj().appended_points(rt, pn, un, H, P)
    assert got.shape == (B, H, 4) and got.is_contiguous()
    want = _oracle_points(rt, pn, un, H, P)
    _assert_same_points(got, want, B)


def test_ids_outside_the_registered_sets_select_nothing():
    rt = torch.tensor([1.0, 2.0, 3.0, 4.0])
    pn, un = torch.tensor([0.0, 7.0, 2.0, 1.5]), torch.tensor([0.0, 1.0, 3.0, 1.0])
    got = HostRefTraj().appended_points(rt, pn, un, 3, 10)
    want = _oracle_points(rt, pn, un, 3, 10)
    _assert_same_points(got, want, 4)
    assert (got[1] == 0).all() and (got[3] == 0).all() and (got[0] != 0).any()
    assert (got[2, :, 0] == 0).all() and (got[2, :, 3] == 0).all() and (got[2, :, 1] != 0).all()   # known path, unknown speed profile


def test_custom_trajectory_parameters():
    params = orc.ref_params(CUSTOM["path_para"], CUSTOM["u_para"])
    g = torch.Generator().manual_seed(5)
    B, H, P = 512, 8, 10
    rt = 20 * torch.rand(B, generator=g)
    pn, un = torch.randint(0, 4, (B,), generator=g).float(), torch.randint(0, 2, (B,), generator=g).float()
    got = HostRefTraj(ref_constants(**CUSTOM)).appended_points(rt, pn, un, H, P)
    want = _oracle_points(rt, pn, un, H, P, params)
    _assert_same_points(got, want, B)


@pytest.mark.parametrize("name", ["step_veh_p10", "step_veh_p30", "step_veh_p10_refpara", "step_veh_p10_nomask", "step_veh2dof_p10",
                                  "step_veh2dof_p10_refpara"])
def test_provider_reproduces_the_points_the_reference_appended(name):
    """`s<k>/ref_last` of the step fixtures is the point the UNMODIFIED reference appended at step k.  Bit for bit on a host whose
    vector math library is the one the fixtures were recorded with (tests/golden/host_libm_canary.json); elsewhere x, y, u at
    the reference's own tolerance and the heading inside the finite difference's last-bit band."""
    g = load_golden(name)
    meta = golden_meta(g)
    data = data_from_golden(g)
    P, n = meta["cfg"].get("pre_horizon", 10), int(g["meta/nsteps"])
    extra = meta["extra"]
    traj = HostRefTraj(ref_constants(extra.get("path_para"), extra.get("u_para")))
    got = traj.appended_points(data["ref_time"], data["path_num"], data["u_num"], n, P).numpy()
    same = _same_libm_as_the_fixtures()
    for s in range(n):
        want = g[f"s{s}/ref_last"]
        mine = got[:, s] if want.shape[1] == 4 else got[:, s, 1:3]   # veh2dofconti keeps (y, phi)
        if same:
            assert np.array_equal(mine, want), (name, s, float(np.abs(mine - want).max()))
        else:
            phi = 2 if want.shape[1] == 4 else 1
            rest = [c for c in range(want.shape[1]) if c != phi]
            np.testing.assert_allclose(mine[:, rest], want[:, rest], rtol=1e-5, atol=2e-5)
            assert np.abs(mine[:, phi] - want[:, phi]).max() < 2e-3


def test_the_hosts_sin_and_cos_do_not_depend_on_an_elements_position():
    """The provider evaluates each sample on its own profile, i.e. at another position of another tensor than the reference
    does - exact only if the library's result for a value does not depend on where it stands (true of MKL VML and Sleef)."""
    g = torch.Generator().manual_seed(1)
    x = torch.rand(100003, generator=g) * 40 - 5
    perm = torch.randperm(x.numel(), generator=g)
    for f in (torch.sin, torch.cos):
        a = f(x)
        b = torch.empty_like(a)
        b[perm] = f(x[perm].contiguous())
        assert torch.equal(a, b)
        for n in (1, 3, 17, 33, 4097):
            assert torch.equal(f(x[5:5 + n].clone()), a[5:5 + n])


def test_pipeline_request_and_collect_on_the_host():
    g = torch.Generator().manual_seed(2)
    B, H, P = 300, 7, 10
    mk = lambda: dict(ref_time=20 * torch.rand(B, generator=g), path_num=torch.randint(0, 4, (B,), generator=g).float(),   # noqa: E731
                      u_num=torch.randint(0, 2, (B,), generator=g).float())
    pipe = ReferencePointPipeline(HostRefTraj(), P)
    a, b = mk(), mk()
    pipe.request(a, H, None)
    pipe.request(b, H, None)
    got_b = pipe.collect(b, H, None)     # any order
    got_a = pipe.collect(a, H, None)
    got_a2 = pipe.collect(a, H, None)    # nothing pending any more: evaluated on the spot
    assert pipe.evaluated == 3
    for d, got in ((a, got_a), (b, got_b), (a, got_a2)):
        assert torch.equal(got, HostRefTraj().appended_points(d["ref_time"], d["path_num"], d["u_num"], H, P))
    pipe.close()


def test_single_rank_phases_reducer_takes_the_multi_rank_code_path_without_collectives():
    """bench.py --dp-path: one rank, the algorithms still run the two-phase backward; nothing is exchanged, `_pending` is consumed."""
    from gops_amd.trainer.grad_sync import GradAllReducer
    plain, phased = GradAllReducer(), GradAllReducer(single_rank_phases=True)
    assert not plain.overlap_enabled() and phased.overlap_enabled()
    g = [torch.ones(4), torch.ones(2)]
    phased.start_(g)                      # world size 1: no collective, no handle
    assert phased._works == []
    info = {"grad": g, "_pending": True}
    out = phased.average_(info, defer_scale=True)
    assert "_pending" not in out and "_grad_scale" not in out and torch.equal(out["grad"][0], torch.ones(4))
    assert not GradAllReducer(overlap=False, single_rank_phases=True).overlap_enabled()
