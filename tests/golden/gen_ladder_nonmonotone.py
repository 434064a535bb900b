"""Provenance of tests/golden/ladder_nonmonotone.npz (DATA: one prior covariance and filter mean).

robust_cholesky (envs/dynamics.py:402-417 of the reference) returns the factor of the FIRST matrix a + 10^i I, i = -6 .. 9, that
scipy.linalg.cholesky accepts.  For a diverged filter (n + lambda) P is numerically rank one -- here eigenvalues from -5e-6 to 7.5e14,
condition 1e20 -- and its low rungs' jitter (1e-6 .. 1e-2) is below the rounding noise of the last pivots: whether a rung succeeds is
decided by the last bits, and success is NOT monotone in the jitter.  Round 3 tried a two-pass search in the fused kernel (last rung of
each group of four, then the group: two factorisations whatever the rung) that assumes monotonicity; every test passed, and a
1 000-episode soak lost filters that should have survived.  build_ablate/ladder_ab.py isolated the first object / step where the two
searches part: object 8 900 of the 20 000-object bench workload at step 271.

This file was produced ON THE GPU BOX (the case is defined by the kernel's arithmetic, not by LAPACK's: on the host scipy accepts rungs
1 .. 15 of this matrix and rejects rung 0 -- monotone):

    hipcc ... -DSSA_LADDER_BY_PASSES -DSSA_LADDER_TWO_PASS -o build_ablate/libs/twopass.so ssa-gym_amd/csrc/ssa_kernels.hip
    LIB=build_ablate/libs/twopass.so OUT=/tmp/ladder_twopass.npz python build_ablate/ladder_ab.py          # record the two-pass build
    LIB=ssa-gym_amd/libssa_hip.so REF=/tmp/ladder_twopass.npz OUT=gpurun_out/r4b/ladder_case.npz python build_ablate/ladder_ab.py

and copied here (P = the object's prior covariance P_filter[270][8900], x = its mean, step, obj).  tests/test_hip_step.py::
test_ladder_first_success_on_the_non_monotone_case runs it through ssa_ladder_probe_f64 (the fused kernels' ladder + which rungs
factorise in their arithmetic), ssa_robust_cholesky6_f64 (the sequential register ladder) and the oracle."""
