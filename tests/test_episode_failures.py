"""Episode-level failure behaviour (parity of what a whole episode DOES, not of one step).

In the reference a filter fails when predict() raises LinAlgError (robust_cholesky's ladder exhausted, dynamics.py:402-417) or
returns NaN (newton() gave up, farnocchia.py:337-353); filter_error() then overwrites it with the 1e20 sentinels
(ssa_tasker_simple_2.py:271-285, 369-382) and the next np.max(delta_pos) ends a 'jones' / 'shaped' episode (:325-343).  With the
env defaults (alpha = 1e-4) the reference loses 2-3 % of its filters over a 480-step round-robin episode: the prior covariance
sum_i Wc_i y_i y_i^T carries Wc_0 ~ -2e8, and for a diverged (hyperbolic) prior its cancellation noise makes P indefinite.

Measured on the oracle (reference order of operations) and on the HIP path, same inputs, FIVE workloads (seeds):
  * SSA_PROP_HYBRID + SSA_FLAG_REFERENCE_COV (the env default: what `fx_xyz_farnocchia` resolves to): the series solver on strong-elliptic
    states, the reference's formulas (fast primitives) on every other state -- the BEHAVIOUR-FAITHFUL variant at speed;
  * SSA_PROP_ELEMENTS + SSA_FLAG_REFERENCE_COV ('elements'): the reference's operation order throughout;
  * SSA_PROP_FG ('fg'): more accurate than the reference on diverged states, its filters survive -- zero failures, asserted as exactly
    that: a documented behavioural difference.
What "the same behaviour" can mean is bounded by the oracle itself: its two summation orders of the SAME arithmetic (reference order /
centred means) lose 52 vs 64 filters on seed 7 and agree on WHICH filters only to a Jaccard overlap of 0.07-0.12 -- which filter of the
susceptible population goes first is decided by rounding noise.  The gate therefore is: pooled counts per window (~270 failures: a
three-sigma band of +-6 %), the status-code mix, the 'jones' step, the overlap of the failed sets with the oracle's at the level of the
oracle's own yardstick, the POPULATION the failures come from (filters the oracle sees diverged by step 300: 13 % of the objects, 57 %
of the oracle's late failures) and the distribution of first-failure steps.  A variant that reproduced the count with the wrong filters
(random ones: 13 % from that population, overlap 0.01) fails it.
"""
import json
import os

import numpy as np
import pytest

import episode_workload as ew
from conftest import GOLDEN

FIXTURE = os.path.join(GOLDEN, "episode_failures_oracle.json")


def band(a, b):
    """two failure counts are 'the same statistics' when they differ by no more than three standard deviations of a
    Poisson count of their size (+3): the oracle's own two summation orders differ by 52 vs 64 at step 479"""
    return abs(a - b) <= 3.0 * np.sqrt(max(a, b)) + 3.0


def ks_distance(a, b):
    """two-sample Kolmogorov-Smirnov statistic of two samples of first-failure steps"""
    a, b = np.sort(np.asarray(a, dtype=float)), np.sort(np.asarray(b, dtype=float))
    grid = np.concatenate([a, b])
    return float(np.max(np.abs(np.searchsorted(a, grid, side="right") / len(a) - np.searchsorted(b, grid, side="right") / len(b))))


def late_in_population(run, population, after=ew.DIVERGED_STEP):
    """(failures after step `after` that lie in `population`, failures after step `after`)"""
    late = [j for j, s in zip(run["failed_ids"], run["failed_first_step"]) if s > after]
    return len(set(late) & set(population)), len(late)


@pytest.fixture(scope="module")
def fixture():
    return json.load(open(FIXTURE))


@pytest.fixture(scope="module")
def workload():
    return ew.workload(m=2000, seed=7)


@pytest.fixture(scope="module")
def oracle_runs(workload):
    import oracle as orc
    orc.build()
    return {"reference_order": ew.run_oracle(workload), "centred_means": ew.run_oracle(workload, centred=True)}


def test_oracle_episode_matches_the_committed_fixture(oracle_runs, fixture):
    """the fixture (tests/golden/gen_episode_failures.py, generated in the build container) pins the oracle's episode: same
    libm, same summation order -> same counts; a different host libm may move single filters across the ladder's edge"""
    fx = fixture["seed7"]
    for name in ("reference_order", "centred_means"):
        got, want = oracle_runs[name], fx[name]
        print(name, {k: got[k] for k in ("failed_at", "status_mix", "jones_done_step")})
        assert got["jones_done_step"] == want["jones_done_step"]
        for w in ew.WINDOWS:
            assert band(got["failed_at"][w], want["failed_at"][str(w)]), (name, w, got, want)
        assert ew.jaccard(got["failed_ids"], want["failed_ids"]) >= 0.8          # (same arithmetic, same machine class: the same filters)
    # the headline fact: the reference arithmetic loses filters late in the episode, almost all to LinAlgError
    ro = oracle_runs["reference_order"]
    assert ro["failed_at"][240] <= 2 and 30 <= ro["failed_at"][479] <= 90
    assert ro["status_mix"][ew.orc.ST_PREDICT_LINALG] >= 0.8 * ro["failed_at"][479]


def test_the_oracles_own_yardstick(fixture):
    """what two faithful evaluations of the SAME arithmetic agree on (fixture only, no GPU): the oracle in the reference's summation order
    against the oracle with centred means, five seeds -- pooled counts inside the band, WHICH filters: overlap 0.07-0.12 (chance, for
    ~55 of 2 000: 0.014), the population: more than half of the late failures from the 13 % of filters diverged by step 300."""
    J, cnt_a, cnt_b, pop_in, pop_n, div_n = [], 0, 0, 0, 0, 0
    for seed in ew.SEEDS:
        a, b = fixture["seed%d" % seed]["reference_order"], fixture["seed%d" % seed]["centred_means"]
        J.append(ew.jaccard(a["failed_ids"], b["failed_ids"]))
        cnt_a += a["failed_at"]["479"]
        cnt_b += b["failed_at"]["479"]
        k, n = late_in_population(b, a["diverged_at_%d" % ew.DIVERGED_STEP])
        pop_in, pop_n, div_n = pop_in + k, pop_n + n, div_n + len(a["diverged_at_%d" % ew.DIVERGED_STEP])
    print("[yardstick] pooled failures %d vs %d; overlap per seed %s; late failures of one order inside the other's diverged population: %d of %d "
          "(population = %.3f of the objects)" % (cnt_a, cnt_b, np.round(J, 3).tolist(), pop_in, pop_n, div_n / (2000.0 * len(ew.SEEDS))))
    assert band(cnt_a, cnt_b) and 200 <= cnt_a <= 350
    assert 0.04 <= np.mean(J) <= 0.3
    assert pop_in >= 0.4 * pop_n and div_n <= 0.2 * 2000 * len(ew.SEEDS)


@pytest.mark.gpu
def test_faithful_variants_reproduce_the_reference_failures_fg_does_not(fixture):
    import torch
    import ssa_gym_amd
    from ssa_gym_amd import _lib, device, host, engine
    ssa_gym_amd.build()
    _lib.load()

    class H:
        pass
    hip = H()
    hip.torch, hip.lib, hip.dev, hip.host, hip.engine = torch, _lib, device, host, engine
    runs = {name: {} for name in ("hybrid", "elements", "fg")}
    extra = {}
    for seed in ew.SEEDS:
        w = ew.workload(m=2000, seed=seed)
        for name in runs:
            runs[name][seed] = ew.run_hip(hip, w, name)
        if seed == 7:      # what each ingredient contributes
            extra = {"elements + centred covariance": ew.run_hip(hip, w, "elements", covariance="centred"),
                     "fg + reference covariance": ew.run_hip(hip, w, "fg", covariance="reference")}
    ref = {seed: fixture["seed%d" % seed]["reference_order"] for seed in ew.SEEDS}
    cen = {seed: fixture["seed%d" % seed]["centred_means"] for seed in ew.SEEDS}
    yard = [ew.jaccard(ref[s]["failed_ids"], cen[s]["failed_ids"]) for s in ew.SEEDS]
    ref_first = sum((ref[s]["failed_first_step"] for s in ew.SEEDS), [])
    cen_first = sum((cen[s]["failed_first_step"] for s in ew.SEEDS), [])
    print("[oracle] pooled failed @479: %d (reference order) / %d (centred means); overlap of the two per seed %s; first-failure step median %d / %d, "
          "KS distance between them %.3f" % (sum(ref[s]["failed_at"]["479"] for s in ew.SEEDS), sum(cen[s]["failed_at"]["479"] for s in ew.SEEDS),
                                          np.round(yard, 3).tolist(), np.median(ref_first), np.median(cen_first), ks_distance(ref_first, cen_first)))
    # ---- the behaviour-faithful variants
    for name in ("hybrid", "elements"):
        r = runs[name]
        # (1) pooled counts per window
        for wdw in ew.WINDOWS:
            a, b = sum(r[s]["failed_at"][wdw] for s in ew.SEEDS), sum(ref[s]["failed_at"][str(wdw)] for s in ew.SEEDS)
            assert band(a, b), (name, wdw, a, b)
        tot = sum(r[s]["failed_at"][479] for s in ew.SEEDS)
        assert tot >= 150
        # (2) the ladder, not Kepler; nothing from the update
        mix = np.sum([r[s]["status_mix"] for s in ew.SEEDS], axis=0)
        assert mix[_lib.ST_PREDICT_LINALG] >= 0.8 * tot and mix[_lib.ST_UPDATE_NAN] == 0 and mix[_lib.ST_UPDATE_LINALG] == 0
        # (3) a 'jones' episode ends where the reference's does
        for s in ew.SEEDS:
            assert abs(r[s]["jones_done_step"] - ref[s]["jones_done_step"]) <= 1, (name, s)
        # (4) WHICH filters: overlap with the oracle's failed sets at the level of the oracle's own yardstick (its two summation orders)
        J = [ew.jaccard(r[s]["failed_ids"], ref[s]["failed_ids"]) for s in ew.SEEDS]
        Jc = [ew.jaccard(r[s]["failed_ids"], cen[s]["failed_ids"]) for s in ew.SEEDS]
        # (5) the population they come from: filters the ORACLE sees diverged by step 300
        k = n = 0
        for s in ew.SEEDS:
            kk, nn = late_in_population(r[s], ref[s]["diverged_at_%d" % ew.DIVERGED_STEP])
            k, n = k + kk, n + nn
        # (6) WHEN: the distribution of first-failure steps
        first = sum((r[s]["failed_first_step"] for s in ew.SEEDS), [])
        ks = ks_distance(first, ref_first)
        print("[%s] pooled failed @479: %d; per seed %s; overlap with the oracle per seed %s (mean %.3f; with its centred-means run %.3f; the oracle's "
              "own yardstick %.3f); late failures inside the oracle's diverged population: %d of %d; first-failure step median %d (oracle %d), KS %.3f"
              % (name, tot, [r[s]["failed_at"][479] for s in ew.SEEDS], np.round(J, 3).tolist(), np.mean(J), np.mean(Jc), np.mean(yard), k, n,
                 np.median(first), np.median(ref_first), ks))
        assert np.mean(J) >= 0.5 * np.mean(yard) and np.mean(J) >= 0.03, (name, J, yard)       # (chance: 0.014)
        assert k >= 0.4 * n, (name, k, n)                                                         # (a random 13 % would give 0.13)
        assert abs(np.median(first) - np.median(ref_first)) <= 25 and ks <= 0.25, (name, np.median(first), np.median(ref_first), ks)
        assert all(150 <= r[s]["first_failure_step"] <= 400 for s in ew.SEEDS)
    # ---- fg: NOT the reference's failure behaviour (more accurate on diverged states; nothing fails), stated as such
    for s in ew.SEEDS:
        fg = runs["fg"][s]
        assert all(v == 0 for v in fg["failed_at"].values()), (s, fg["failed_at"])
        assert abs(fg["jones_done_step"] - ref[s]["jones_done_step"]) <= 1        # a 'jones' episode still ends at the same step
    # what each ingredient contributes: the reference's covariance arithmetic alone (with the accurate propagator) fails nothing,
    # its propagator alone (with the cancellation-free covariance) a tenth of the reference's count
    assert all(v == 0 for v in extra["fg + reference covariance"]["failed_at"].values())
    assert extra["elements + centred covariance"]["failed_at"][479] <= 0.4 * ref[7]["failed_at"]["479"]
