"""Provenance of tests/golden/ladder_illconditioned_tile.npz (DATA: the prior covariances and filter means of one wavefront's four objects).

robust_cholesky (envs/dynamics.py:402-417 of the reference) returns the factor of the FIRST matrix a + 10^i I, i = -6 .. 9, that
scipy.linalg.cholesky accepts, a = (n + lambda) P.  For a diverged filter a is numerically rank one -- here eigenvalues from -5e-6 to
7.5e14, condition 1e20.  Round 3 tried a "two-pass" search in the fused kernel (last rung of each group of four, then the group), saw a
1 000-episode soak lose filters that should have survived, and reverted it with the explanation that success is not monotone in the
jitter.  Round 4 measured that explanation and it does not hold: ssa_ladder_probe_f64 -- the fused ladder plus, per rung, whether it
factorises in the kernel's arithmetic -- gives the SAME rung (1) for this tile under the two-pass build, the four-rungs-per-pass build and
the one-pass ladder, and the mask is monotone (rung 0 fails, 1 .. 15 succeed), also under 4 096 one-ulp perturbations.  What differed was
the FACTOR: rows 3 .. 5 by 1e-8 relative.  Under -ffp-contract=fast the compiler had fused scale * p + jit into one fma at one call
site of the factorisation and -- where it could share scale * p with a neighbouring factorisation of the same matrix -- not at another;
one rounding in a diagonal entry, amplified 1e8-fold by the conditioning.  The kernels now form that entry in the reference's own two
roundings (scaled_entry in csrc/ssa_kernels.hip: the product (n + lambda) P rounded, then + jitter: dynamics.py:410), contraction off,
and tests/test_hip_step.py::test_ladder_on_the_ill_conditioned_tile pins exactly that.

This file was produced ON THE GPU BOX (build_ablate/ladder_ab.py: first object / step at which the two builds' filter states part over a
20 000-object episode of the bench workload -- object 19 669, step 277, i.e. tile 4 917, row 1):

    hipcc ... -DSSA_LADDER_BY_PASSES -DSSA_LADDER_TWO_PASS -o build_ablate/libs/twopass.so ssa-gym_amd/csrc/ssa_kernels.hip
    LIB=build_ablate/libs/twopass.so OUT=/tmp/ladder_twopass.npz python build_ablate/ladder_ab.py          # record the two-pass build
    LIB=ssa-gym_amd/libssa_hip.so REF=/tmp/ladder_twopass.npz OUT=gpurun_out/r4e/ladder_case.npz python build_ablate/ladder_ab.py

P_tile [4][6][6] = P_filter[276] of objects 19 668 .. 19 671, x_tile their means, status_tile their status words (all healthy), P / x / obj /
step the object that parted.  profiles/r04_ladder_case.txt holds the probe's output under the three builds."""
