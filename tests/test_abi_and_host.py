"""CPU-only checks: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU),
loads, exports every symbol include/ssa_hip.h declares with matching struct layouts, refuses
CPU tensors (no fallback), and the init-time host logic matches the oracle / goldens.
No compute call is made here.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc
from conftest import ROOT, golden


@pytest.fixture(scope="module")
def pkg():
    import ssa_gym_amd
    ssa_gym_amd.build()
    return ssa_gym_amd


def test_library_builds_and_exports_every_declared_symbol(pkg):
    from ssa_gym_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ssa_hip.h")).read()
    declared = set(re.findall(r"\b(ssa_[a-z0-9_]+)\s*\(", header))
    declared -= {"ssa_consts", "ssa_step_params"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ssa_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.ssa_build_info()


def test_struct_layouts_match_the_header(pkg, tmp_path):
    """compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    from ssa_gym_amd import _lib
    structs = ("ssa_consts", "ssa_step_params", "ssa_rollout_params", "ssa_closed_loop_params")
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "ssa_hip.h"', 'int main(void){']
    want = []
    for name in structs:
        st = getattr(_lib, name)
        src.append('printf("%%zu\\n", sizeof(%s));' % name)
        want.append(C.sizeof(st))
        for f, _ in st._fields_:
            src.append('printf("%%zu\\n", offsetof(%s, %s));' % (name, f))
            want.append(getattr(st, f).offset)
    src.append('return 0;}')
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).decode().split()]
    assert out == want
    # constants mirrored from the header
    hdr = open(os.path.join(ROOT, "include", "ssa_hip.h")).read()
    for name in ("UPD_STRIDE", "UPD_OBS_TAKEN", "UPD_Z_TRUE", "UPD_Y", "UPD_S", "UPD_SIGMAS_H", "UPD_VISIBLE",
                 "UPD_ACTION", "STAT_STRIDE", "STAT_MAX_DPOS", "STAT_ARGMAX_SPOS", "STAT_N_FAILED", "ST_UPDATE_LINALG",
                 "OBS_XYZ", "PROP_FG", "FLAG_RESAMPLE", "ABI_VERSION", "LOOP_ARGMAX_SPOS", "LOOP_DEBUG_WITHHOLD"):
        m = re.search(r"#define SSA_%s\s+(\d+)u?" % name, hdr)
        assert m and int(m.group(1)) == getattr(_lib, name), name


def test_no_cpu_fallback(pkg):
    """the product refuses to compute without a GPU instead of silently using the CPU."""
    import torch
    from ssa_gym_amd import _lib, device
    with pytest.raises(_lib.SsaHipError):
        device.propagate(torch.zeros((4, 6), dtype=torch.float64), 20.0)
    if not torch.cuda.is_available():
        from ssa_gym_amd import engine, host
        g = golden("ukf_step_golden.npz")
        c = host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -1.57, g["obs_lla"])
        with pytest.raises(_lib.SsaHipError):
            engine.HotPathEngine(c, 4, 1, np.eye(3)[None], np.zeros((1, 1, 4, 3)), history=2)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under ssa-gym_amd/ may reference it."""
    pk = os.path.join(ROOT, "ssa-gym_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "ssa_oracle" not in txt and "ukf_numpy" not in txt, f


def test_host_constants_match_oracle_and_reference(pkg):
    from ssa_gym_amd import host
    g = golden("geometry_golden.npz")
    u = golden("ukf_step_golden.npz")
    # T2 lla2ecef vs reference golden
    for lla, ecef in zip(g["llas"], g["ecefs"]):
        np.testing.assert_allclose(host.lla2ecef(lla), ecef, rtol=1e-15, atol=1e-9)
    assert host.WGS84_A == g["wgs84"][0] and host.WGS84_F == g["wgs84"][1]
    # U4 Q vs composite golden (filterpy formula)
    Q = host.Q_discrete_white_noise(dim=2, dt=20.0, var=0.000025 ** 2, block_size=3, order_by_dim=False)
    assert np.array_equal(Q, u["Q"])
    # U1 weights
    for alpha in (1e-3, 1e-4, 1.0):
        Wm, Wc, scale = host.merwe_weights(alpha, 2.0, -3)
        Wm2, Wc2, scale2 = orc.merwe_weights(alpha, 2.0, -3)
        assert np.array_equal(Wm, Wm2) and np.array_equal(Wc, Wc2) and scale == scale2
        sm, sc = host.exact_weight_sums(Wm, Wc)
        assert abs(sm) < 1e-7 and abs(sc - (1 + 1 - alpha ** 2 + 2.0)) < 1e-7
        assert abs(sm - float(np.sum(Wm.astype(np.longdouble)) - 1)) < 1e-12
    c = host.make_consts(u["Q"], u["R"], 1e-4, 2.0, -3, 20.0, np.radians(-90), u["obs_lla"], 'aer', 'fg')
    np.testing.assert_allclose(np.array(c.obs_itrs), u["obs_itrs"], rtol=1e-15)
    assert c.Wm0 == orc.merwe_weights(1e-4, 2.0, -3)[0][0] and c.update_interval == 1
    # enu matrix is orthonormal and maps the local vertical to +w
    T = np.array(c.enu).reshape(3, 3)
    np.testing.assert_allclose(T.T @ T, np.eye(3), atol=1e-15)


def test_bench_cpu_share_respects_cgroup_quota(tmp_path, monkeypatch):
    """bench.host_cpu_share(): the all-cores CPU baseline must size its thread pool by the container's CPU quota, not
    by the CPUs it can see (256 visible / 16 granted on the GPU boxes)."""
    import builtins
    import importlib.util
    import os
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    real_open = builtins.open

    def fake(content):
        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                f = tmp_path / "cpu.max"
                f.write_text(content)
                return real_open(f, *a, **k)
            return real_open(path, *a, **k)
        return _open
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert bench.host_cpu_share() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert bench.host_cpu_share() == 256
    monkeypatch.setattr(builtins, "open", fake("50000 100000\n"))
    assert bench.host_cpu_share() == 1


# ---------------------------------------------------------------- register budget of the built code object
def _code_object(tmp_path):
    """the gfx950 code object embedded in the built libssa_hip.so (the file that ships), its metadata and disassembly"""
    import shutil
    import subprocess
    import ssa_gym_amd
    from ssa_gym_amd import _build
    ssa_gym_amd.build()
    b = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(b, "clang-offload-bundler")):
        pytest.skip("no ROCm LLVM tools here")
    fb, co = str(tmp_path / "fb.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([os.path.join(b, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, _build.LIB, str(tmp_path / "copy.so")])
    subprocess.check_call([os.path.join(b, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fb, "--output=" + co])
    notes = subprocess.check_output([os.path.join(b, "llvm-readelf"), "--notes", co], text=True)
    dis = subprocess.check_output([os.path.join(b, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    shutil.rmtree(str(tmp_path), ignore_errors=True)
    return notes, dis


def test_step_kernels_keep_their_register_budget(tmp_path):
    """The step / rollout kernels must fit 96 VGPRs (5 wavefronts per SIMD = the whole 20 000-object step resident in
    one round) and must not spill on the common path: the build relies on a whole-TU compiler switch
    (-disable-machine-licm, _build.py), so a toolchain change that brings the spills back (41.7 MB of scratch traffic
    per launch when it happened) has to fail HERE, not show up as a slower bench.  Scratch accesses are allowed only
    as the save / restore around the rare out-of-line calls of SSA_PROP_ELEMENTS (the complete farnocchia())."""
    import re
    notes, dis = _code_object(tmp_path)
    kern = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        kern[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in
                      ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "group_segment_fixed_size")}
    # the grid-stride instance re-derives its argument block from the kernarg segment pointer: the by-value StepK must sit
    # right behind the six preloaded scalar arguments (offset 40) (a wrong offset there is a wild-pointer GPU fault, not a wrong number)
    for blk in notes.split("- .agpr_count:")[1:]:
        if "step_fast_kernel" in re.search(r"\.name:\s+(\S+)", blk).group(1):
            offs = [int(v) for v in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+\d+\s+\.value_kind:\s+by_value", blk)]
            kinds = re.findall(r"\.value_kind:\s+(\w+)", blk)
            assert kinds[:6] == ["by_value"] * 2 + ["global_buffer"] * 4 and 40 in offs, (kinds[:7], offs)
    hot = [k for k in kern if "step_fast_kernel" in k or "rollout_kernel" in k or "closed_loop_kernel" in k]
    assert len(hot) == 16, hot                     # 4 propagators x {one tile, multi tile, rollout, closed loop}
    for k in hot:
        assert kern[k]["vgpr_count"] <= 96, (k, kern[k])
        assert kern[k]["private_segment_fixed_size"] <= 320, (k, kern[k])     # the callees' frames only (two levels: lean form, complete restatement)
        assert kern[k]["group_segment_fixed_size"] <= 160 * 1024 // 20, (k, kern[k])   # 20 wavefronts' tiles per CU
    # disassembly: every scratch access of a hot kernel lies next to an out-of-line call (s_swappc_b64)
    bodies = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", dis)
    checked = 0
    for name, body in zip(bodies[1::2], bodies[2::2]):
        if not ("step_fast_kernel" in name or "rollout_kernel" in name or "closed_loop_kernel" in name):
            continue
        ins = [ln.split()[0] for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(("//", ";"))]
        calls = [i for i, op in enumerate(ins) if op == "s_swappc_b64"]
        stray = [i for i, op in enumerate(ins) if op.startswith("scratch_") and not (calls and min(abs(i - c) for c in calls) <= 96)]
        # none in any instance -- the grid-stride one once spilled the object / env / action words across the propagator in
        # EVERY wavefront (24 bytes per lane and tile: 61 MB of scratch writes per 160 000-object step, found as write
        # traffic 1.58x the algorithmic bytes)
        # (the persistent closed loop of the two CALLING propagators carries the loop's state -- flags, rings, decision words -- across the
        # call as well: the allocator parks two or three of those values in scratch for the whole step, a store each per step on the common
        # path; everywhere else: none)
        allowed = 6 if ("closed_loop_kernel" in name and ("ILi0E" in name or "ILi3E" in name)) else 0
        assert len(stray) <= allowed, (name, stray[:8], "scratch access away from any call: a spill on the common path")
        if "ILi0E" not in name and "ILi3E" not in name:     # FG / J2 instances make no out-of-line call: no scratch at all
            assert kern[name]["private_segment_fixed_size"] == 0 and kern[name]["vgpr_spill_count"] == 0, (name, kern[name])
        checked += 1
    assert checked == 16


def test_acceleration_tokens_and_covariance_form_resolve_on_the_host(pkg):
    """config resolution is host logic (no GPU): the fx_xyz_cowell / ad tokens of envs/dynamics.py:168-201 and the
    covariance form that goes with each propagator (SSA_FLAG_REFERENCE_COV with 'elements': the behaviour-faithful variant)."""
    import functools
    from ssa_gym_amd import _lib, host
    from ssa_gym_amd.envs import dynamics as D, env_config
    from ssa_gym_amd.envs._config import kernel_consts, resolve_kernel_variant, resolve_perturbation
    fj = functools.partial(D.fx_xyz_cowell, ad=D.ad_j2, J2=1e-3, rtol=1e-11)
    assert resolve_kernel_variant(dict(env_config, fx=fj)) == ('aer', 'j2')
    assert resolve_perturbation(dict(env_config, fx=fj)) == (1e-3, host.R_EQ_EARTH)
    assert resolve_perturbation(dict(env_config, fx=D.fx_xyz_cowell)) == (0.0, host.R_EQ_EARTH)
    assert resolve_perturbation(dict(env_config)) is None
    with pytest.raises(NotImplementedError):
        resolve_perturbation(dict(env_config, fx=D.fx_xyz_cowell.with_ad(D.ad_j2, J3=1.0)))
    Q, R, lla = np.eye(6), np.eye(3), np.array([0.6, -1.3, 20.0])
    flags = {}
    for prop in ('fg', 'elements', 'j2', 'hybrid'):
        c, model = kernel_consts(dict(env_config, propagator=prop), Q, R, 20.0, 0.0, lla)
        flags[prop] = c.flags
        assert model == 'aer'
    assert flags['elements'] & _lib.FLAG_REFERENCE_COV and flags['hybrid'] & _lib.FLAG_REFERENCE_COV
    assert not flags['fg'] & _lib.FLAG_REFERENCE_COV and not flags['j2'] & _lib.FLAG_REFERENCE_COV
    # the env default (the reference's own token name): the behaviour-faithful variant
    c, _ = kernel_consts(dict(env_config), Q, R, 20.0, 0.0, lla)
    assert c.propagator == _lib.PROP_HYBRID and c.flags & _lib.FLAG_REFERENCE_COV
    assert resolve_kernel_variant(dict(env_config)) == ('aer', 'hybrid') and resolve_kernel_variant(dict(env_config, fx=D.fx_xyz_farnocchia_fg)) == ('aer', 'fg')
    c, _ = kernel_consts(dict(env_config, propagator='fg', covariance_form='reference', resample_sigmas=True), Q, R, 20.0, 0.0, lla)
    assert c.flags == (_lib.FLAG_REFERENCE_COV | _lib.FLAG_RESAMPLE)
    c, _ = kernel_consts(dict(env_config, propagator='elements', covariance_form='centred'), Q, R, 20.0, 0.0, lla)
    assert c.flags == 0
    with pytest.raises(ValueError):
        kernel_consts(dict(env_config, covariance_form='exact'), Q, R, 20.0, 0.0, lla)
