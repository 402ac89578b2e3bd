"""The sub-step arithmetic of the TIMED kernels, pinned to the oracle.

The ground flag of a step (manytor.py:191-192) is `z < 0` of the observation / pickup frames at ANY of the S
interpolated poses (manytor.py:182).  The production kernels do not evaluate a sincos per pose: the S - 2 interior
poses come from an angle-addition recurrence run from both ends (kernels.h: route_kinematics /
route_kinematics_split).  With MT_FLAG_DEBUG_ZMIN the very kernels that are timed -- step_kernel (streaming /
prefetch), step_split_kernel<2|4>, rollout_kernel, rollout_split_kernel -- also store the z-minimum their ground
test used (MT_F_ZMIN: one extra store behind a NULL-pointer test, same template instantiation).  Here that value
is compared with the fp64 signed minimum of the reference's own sub-step poses:

  * fixture F3 (the reference's joints_coordinates at all 25 sub-steps of 32 routes),
  * every env of 1 048 576 random integer-degree routes against the C oracle, for the reference arm (interleaved
    schedule), the 7-joint table (sequential schedule + ZJoints trim) and a runtime 5-joint table, at
    S in {2, 3, 25, 64}, from the zero pose and from a random whole-degree pose,
  * float (non-integer) actions through mt_step,
  * every forced schedule (MT_SPLIT / MT_PREFETCH), and the fused rollouts after a PoseCache hand-over.

Tolerance: POS_TOL = 1e-4, an order of magnitude inside the 1e-3 guard band of the ground flag.
"""
import json
import os

import numpy as np
import pytest

from parity_util import POS_TOL

pytestmark = pytest.mark.gpu

RECORD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "zmin_errors.jsonl")


def record(**kw):
    """Measured maxima go to gpurun_out/ (scratch) so a run can be quoted in BASELINE.md / profiles/."""
    try:
        os.makedirs(os.path.dirname(RECORD), exist_ok=True)
        with open(RECORD, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module")
def m():
    import manytor_amd
    if manytor_amd.device_count() < 1:
        pytest.fail("gpu tests need a visible MI355X and the in-tree libmanytor_hip.so")
    return manytor_amd


def tables(m):
    rng = np.random.RandomState(55)
    rt5 = np.column_stack([rng.uniform(0, 9, 5), rng.choice([-np.pi / 2, 0.3, np.pi / 2], 5), rng.uniform(2, 12, 5),
                           np.zeros(5)])
    return {"ref": (m.REF_DH_TABLE, 51.3), "dh7": (m.DH7_TABLE, 92.6), "rt5": (rt5, 40.0)}


SCHEDULES = {"streaming": ("0", "0"), "prefetch": ("0", "1"), "split2": ("2", "0"), "split4": ("4", "0")}


def force(monkeypatch, schedule):
    if schedule is None:
        monkeypatch.delenv("MT_SPLIT", raising=False)
        monkeypatch.delenv("MT_PREFETCH", raising=False)
    else:
        split, pf = SCHEDULES[schedule]
        monkeypatch.setenv("MT_SPLIT", split)
        monkeypatch.setenv("MT_PREFETCH", pf)


@pytest.mark.parametrize("schedule", [None, "streaming", "prefetch", "split2", "split4"])
def test_zmin_of_the_timed_kernels_on_the_reference_substep_fixture(m, golden, monkeypatch, schedule):
    """F3 = joints_coordinates the reference itself produced at each of the 25 sub-steps of 32 routes: the kernel's
    z-minimum must be the fixture's min over sub-steps of min(z[2], z[3]) (manytor.py:191), and its sign the
    reference's ground flag wherever that minimum is outside the guard band."""
    from oracle import manytor_oracle as mo
    g = golden("f3_substep_trace")
    prev, action, jc = g["prev"], g["action"], g["jc"]
    n = len(prev)
    ref = np.minimum(jc[:, :, 2, 2], jc[:, :, 3, 2]).min(axis=1)
    # the fp64 restatement agrees with the reference's numbers to rounding
    route = np.linspace(prev, action, 25)                                  # (25, n, 4), manytor.py:182
    jo = np.stack([mo.batch_joints_coordinates(route[k]) for k in range(25)], axis=1)
    assert np.abs(np.minimum(jo[:, :, 2, 2], jo[:, :, 3, 2]).min(axis=1) - ref).max() < 1e-9
    force(monkeypatch, schedule)
    eng = m.StepEngine(n, 3, debug_zmin=True)
    eng.reset(np.full((n, 3, 3), 40.0, dtype=np.float32))
    eng.set(m.lib.F_GOALS, prev)
    eng.step(action)
    err = np.abs(eng.zmin() - ref)
    record(test="f3", schedule=schedule or "default", kernel=eng.step_kernel_name(), max_err=float(err.max()))
    assert err.max() <= POS_TOL, err.max()
    clear = np.abs(ref) > 1e-3
    np.testing.assert_array_equal((eng.zmin() < 0)[clear], g["ground_sub"].any(axis=1)[clear])
    np.testing.assert_array_equal((eng.reward() == -1)[clear], (g["reward"] == -1)[clear])


@pytest.mark.parametrize("substeps", [2, 3, 25, 64])
@pytest.mark.parametrize("table_name", ["ref", "dh7", "rt5"])
def test_zmin_every_env_of_a_million_random_routes(m, table_name, substeps):
    """1 048 576 envs x 2 steps with the kernel mt_create picks at this size (and 3 steps of ONE fused launch, whose
    steps 2 and 3 start from the PoseCache): step 1 leaves the zero pose, step 2 runs from one random whole-degree
    pose to another -- |delta| up to 359 degrees, i.e. up to 15 degrees per recurrence rotation at S = 25 and the
    wide-increment path (sincos_deg instead of sincos_deg_small) at S <= 8."""
    from oracle import c_oracle
    from oracle import philox_ref as px
    table, radius = tables(m)[table_name]
    n, k, seed = 1048576, 2, 0xBEEF + substeps
    dof = len(table)
    ids = np.arange(n, dtype=np.uint64)
    eng = m.StepEngine(n, k, dh_table=table, radius=radius, substeps=substeps, debug_zmin=True)
    fused = m.StepEngine(n, k, dh_table=table, radius=radius, substeps=substeps, debug_zmin=True)
    ora = c_oracle.COracle(n, k, table=np.asarray(table), radius=radius, substeps=substeps, threads=16)
    eng.reset_random(seed, 0)
    fused.reset_random(seed, 0)
    ora.reset(eng.points().astype(np.float64))
    worst = 0.0
    for t in range(3):
        act = px.sample_actions(seed, ids, t, dof).astype(np.float64)
        ora.step(act)
        if t < 2:
            eng.step_random(seed, t)
            err = np.abs(eng.zmin() - ora.zmin)
            worst = max(worst, float(err.max()))
            assert err.max() <= POS_TOL, (t, err.max(), int(err.argmax()))
            clear = np.abs(ora.zmin) > 1e-3
            np.testing.assert_array_equal((eng.reward() == -1)[clear], ora.ground_hit[clear])
    fused.rollout_fused(3, seed, 0)                      # last step's z-minimum: computed from a handed-over pose
    err = np.abs(fused.zmin() - ora.zmin)
    assert err.max() <= POS_TOL, ("fused", err.max(), int(err.argmax()))
    record(test="million_routes", table=table_name, S=substeps, kernel=eng.step_kernel_name(), max_err_step=worst,
           max_err_fused_step3=float(err.max()))


@pytest.mark.parametrize("table_name", ["ref", "dh7", "rt5"])
@pytest.mark.parametrize("schedule", ["streaming", "prefetch", "split2", "split4"])
def test_zmin_under_every_forced_schedule(m, monkeypatch, table_name, schedule):
    """The four step schedules and the three rollout schedules are separate code paths around the same device functions
    (route_kinematics / route_kinematics_split): each is pinned on its own, 262 144 envs, per-step and fused."""
    from oracle import c_oracle
    from oracle import philox_ref as px
    table, radius = tables(m)[table_name]
    n, k, seed = 262144, 3, 0xABCD
    dof = len(table)
    ids = np.arange(n, dtype=np.uint64)
    force(monkeypatch, schedule)
    eng = m.StepEngine(n, k, dh_table=table, radius=radius, debug_zmin=True)
    fused = m.StepEngine(n, k, dh_table=table, radius=radius, debug_zmin=True)
    want = {"streaming": "pf=0", "prefetch": "pf=8", "split2": "L=2", "split4": "L=4"}[schedule]
    assert want in eng.step_kernel_name(), eng.step_kernel_name()
    ora = c_oracle.COracle(n, k, table=np.asarray(table), radius=radius, threads=16)
    eng.reset_random(seed, 0)
    fused.reset_random(seed, 0)
    ora.reset(eng.points().astype(np.float64))
    worst = 0.0
    for t in range(4):
        ora.step(px.sample_actions(seed, ids, t, dof).astype(np.float64))
        eng.step_random(seed, t)
        err = np.abs(eng.zmin() - ora.zmin)
        worst = max(worst, float(err.max()))
        assert err.max() <= POS_TOL, (t, err.max())
    fused.rollout_fused(4, seed, 0)
    errf = np.abs(fused.zmin() - ora.zmin)
    assert errf.max() <= POS_TOL, errf.max()
    np.testing.assert_array_equal(fused.zmin(), eng.zmin())              # same device functions: same bits
    record(test="forced_schedule", table=table_name, schedule=schedule, kernel=eng.step_kernel_name(), max_err=worst,
           max_err_fused=float(errf.max()))


@pytest.mark.parametrize("table_name", ["ref", "dh7"])
def test_zmin_with_fractional_degree_actions(m, table_name):
    """A policy's actions are not whole degrees: mt_step (step_kernel<SAMPLE = false>) on uniformly random float
    actions, where the quadrant reduction of the end poses and of the increment really rounds."""
    from oracle import c_oracle
    table, radius = tables(m)[table_name]
    n, k = 1048576, 2
    dof = len(table)
    rng = np.random.RandomState(17)
    eng = m.StepEngine(n, k, dh_table=table, radius=radius, debug_zmin=True)
    ora = c_oracle.COracle(n, k, table=np.asarray(table), radius=radius, threads=16)
    eng.reset_random(3, 0)
    ora.reset(eng.points().astype(np.float64))
    worst = 0.0
    for t in range(2):
        act = rng.uniform(-180.0, 180.0, size=(n, dof)).astype(np.float32)
        eng.step(act)
        ora.step(act.astype(np.float64))
        err = np.abs(eng.zmin() - ora.zmin)
        worst = max(worst, float(err.max()))
        assert err.max() <= POS_TOL, (t, err.max())
    record(test="float_actions", table=table_name, kernel=eng.step_kernel_name(), max_err=worst)


@pytest.mark.parametrize("substeps", [25, 40])
def test_zmin_with_selectable_frames(m, substeps):
    """mt_config.obs_frame / ee_frame other than the last two rows run step_kernel<RtTableF<D>>: the ground test then
    looks at the z of THOSE rows (manytor.py:191 with [2], [3] replaced); routes longer than 12 rotations per half take
    the per-pose sincos instantiation of the same kernel."""
    from oracle import c_oracle
    from oracle import philox_ref as px
    table, radius = tables(m)["rt5"]
    n, k, seed = 65536, 2, 5
    ids = np.arange(n, dtype=np.uint64)
    eng = m.StepEngine(n, k, dh_table=table, radius=radius, substeps=substeps, obs_frame=1, ee_frame=-2, debug_zmin=True)
    assert "RtTableF<5>" in eng.step_kernel_name() and ("trig=1" if substeps > 26 else "trig=0") in eng.step_kernel_name()
    ora = c_oracle.COracle(n, k, table=np.asarray(table), radius=radius, substeps=substeps, threads=16, obs_frame=1, ee_frame=-2)
    eng.reset_random(seed, 0)
    ora.reset(eng.points().astype(np.float64))
    for t in range(3):
        ora.step(px.sample_actions(seed, ids, t, 5).astype(np.float64))
        eng.step_random(seed, t)
        err = np.abs(eng.zmin() - ora.zmin)
        assert err.max() <= POS_TOL, (t, err.max())
    record(test="custom_frames", S=substeps, kernel=eng.step_kernel_name(), max_err=float(err.max()))


def test_zmin_row_needs_the_flag_and_default_handles_carry_no_row(m):
    eng = m.StepEngine(64, 1)
    eng.reset_random(1, 0)
    eng.step_random(1, 0)
    with pytest.raises((ValueError, RuntimeError)):
        eng.zmin()
    with pytest.raises(RuntimeError):
        eng.device_ptr(m.lib.F_ZMIN)
