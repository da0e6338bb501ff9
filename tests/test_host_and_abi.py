"""CPU tests of the host layer (C++ libsart_host) and of the C-ABI surface (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sart_[a-z0-9_]+)\s*\(", txt)) - {"sart_accumulator_len", "sart_accumulator_len_spectra", "sart_mass_scan_len", "sart_angular_scan_len"})  # static inline helpers


def test_libsart_exports_every_declared_symbol():
    lib = L.load_sart()
    names = sorted(_declared("sart.h") + _declared("sart_emission.h"))
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(L.SART_SYMBOLS) == names          # the ctypes table binds exactly the header
    assert lib.sart_abi_version() == L.SART_ABI_VERSION == 5


def test_libsart_host_exports_every_declared_symbol():
    lib = L.load_host()
    names = _declared("sart_host.h")
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(L.SART_HOST_SYMBOLS) == names


def test_kernel_argument_layout_matches_the_code_object(tmp_path):
    """The ray kernels re-read their own arguments from the kernel-argument segment at hard-wired offsets
    (sart_kernels.hip: HistKernArgs).  Those offsets must be the ones the compiler recorded in the code object."""
    import shutil
    import subprocess
    lib = L.load_sart()
    want = (C.c_int32 * 8)()
    lib.sart_internal_kernarg_layout(want)
    obj = tmp_path / "sart_kernels.o"
    shutil.copy(os.path.join(ROOT, "solaraxionraytracing_amd", "csrc", "build", "sart_kernels.o"), obj)
    llvm = "/opt/rocm/lib/llvm/bin"
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(obj)], check=True, capture_output=True, cwd=tmp_path)
    dev = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert len(dev) == 1, dev
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / dev[0])], capture_output=True, text=True,
                           check=True).stdout
    # one "- .args:" list per kernel, followed (later) by its ".name:"
    kernels = re.split(r"\n  - \.a", notes)
    seen = 0
    for k in kernels:
        m = re.search(r"\.name:\s+(\S+)", k)
        if not m or not ("trace_histogram_kernel" in m.group(1) or "trace_angular_scan_kernel" in m.group(1)):
            continue
        args = re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+\.value_kind:\s+(\w+)", k)
        explicit = [(int(o), int(sz)) for o, sz, kind in args if not kind.startswith("hidden")]
        assert [o for o, _ in explicit] == list(want[:6]), (m.group(1), explicit, list(want))
        assert explicit[-1][0] + explicit[-1][1] <= want[7 if "angular_scan" in m.group(1) else 6]
        seen += 1
    # {specialised (vacuum and gas, each with and without the constant path; rotated), generic, generic rotated} x {f64, FIXED64}
    # + the fused mass scan for the four variants that can run the gas stage x {f64, FIXED64}
    # + the fused angular scan (same argument offsets: it shares the reload helpers) {specialised, generic} x {f64, FIXED64}
    assert seen == 14 + 8 + 4


def test_code_object_keeps_what_the_design_counts_on(tmp_path):
    """Properties of the compiled ray kernels that DESIGN.md 3.1 / 3.3 rest on, read from the gfx950 code object: no scratch
    (nothing spills to memory), at most 128 VGPRs (four waves per SIMD at 1024 threads), the LDS budget, and the Philox
    rounds' three-input xors as v_bitop3_b32 (the compiler does not form them by itself: a toolchain that stops accepting the
    builtin, or starts splitting it, shows up here and not as a silent 6 % of instructions)."""
    import shutil
    import subprocess
    obj = tmp_path / "sart_kernels.o"
    shutil.copy(os.path.join(ROOT, "solaraxionraytracing_amd", "csrc", "build", "sart_kernels.o"), obj)
    llvm = "/opt/rocm/lib/llvm/bin"
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(obj)], check=True, capture_output=True, cwd=tmp_path)
    dev = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert len(dev) == 1, dev
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / dev[0])], capture_output=True, text=True,
                           check=True).stdout
    seen = 0
    for k in re.split(r"\n  - \.a", notes):
        m = re.search(r"\.name:\s+(\S+)", k)
        if not m or "trace_" not in m.group(1):
            continue
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
        assert g("private_segment_fixed_size") == 0 and g("vgpr_spill_count") == 0, m.group(1)
        assert g("vgpr_count") <= 128, (m.group(1), g("vgpr_count"))
        if "trace_histogram_kernel" in m.group(1) or "trace_angular_scan_kernel" in m.group(1):
            assert 160 * 1024 - 4096 < g("group_segment_fixed_size") <= 160 * 1024, g("group_segment_fixed_size")
        seen += 1
    assert seen == 27   # fourteen histogram + eight mass-scan + four angular-scan instantiations + the record kernel
    asm = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=gfx950", str(tmp_path / dev[0])], capture_output=True,
                         text=True, check=True).stdout
    body = asm.split("<_ZN4sart22trace_histogram_kernelILi1024ELb1ELb0ELi0ELb1ELb0ELb0EEEvNS_4HotAEPKNS_7DevBlobENS_9TraceArgsEPdNS_4HotBENS_8ScanArgsE>:")
    assert len(body) == 2, "headline instantiation not found in the disassembly"
    head = body[1].split("s_endpgm")[0]
    # 2 per Philox round, 7 rounds (round 6: Philox4x32-7, one block per ray): one block per phase-A copy x 2 copies + the block of
    # stage A0 (shared by four rays) = 42; the three-input xor must stay one instruction (two v_xor_b32 otherwise)
    assert 40 <= head.count("v_bitop3_b32") <= 48 and head.count("v_xor_b32") <= 8, (head.count("v_bitop3_b32"), head.count("v_xor_b32"))
    assert head.count("v_mad_u64_u32") <= 48, head.count("v_mad_u64_u32")      # 2 per round as well: no second block crept back in


def test_nim_binding_declares_every_header_field():
    """integration/sart_ffi.nim (the binding a maintainer of the reference adds) names every field of the three structs and
    every entry point of include/sart.h, and assigns every sart_setup_t field in toSartSetup.  (No Nim compiler in the image:
    this is a text check; the file's own `static: doAssert sizeof` lines pin the layouts when it is compiled.)"""
    nim = open(os.path.join(ROOT, "integration", "sart_ffi.nim")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sart.h")).read(), flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(",")[0:1] + decl.split(",")[1:]:
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part.strip())
                if m:
                    names.append(m.group(1))
        return names

    setup_fields = fields("sart_setup_t")
    assert len(setup_fields) >= 60
    for f in setup_fields:
        if f.startswith("_pad"):
            continue
        assert re.search(r"\b%s\*" % f, nim), "SartSetup lacks field " + f
        assert "result.%s" % f in nim, "toSartSetup does not assign " + f
    for f in fields("sart_trace_params_t"):
        assert re.search(r"\b%s\*" % f, nim), "SartTraceParams lacks field " + f
        assert "result.%s" % f in nim, "sartParams does not set " + f
    for sym in _declared("sart.h"):
        assert re.search(r"proc %s\*\(" % sym, nim), "no importc proc for " + sym
    assert C.sizeof(L.Setup) == 2504 and "sizeof(SartSetup) == 2504" in nim
    assert C.sizeof(L.TraceParams) == 88 and "sizeof(SartTraceParams) == 88" in nim
    assert "..." not in nim                      # no elisions


def test_struct_sizes_match_c_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include "sart_emission.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sart_setup_t),'
                   ' sizeof(sart_axion_t), sizeof(sart_trace_params_t), sizeof(sart_summary_t), sizeof(sart_solar_zone_t),'
                   ' sizeof(sart_emission_params_t), sizeof(sart_opacity_tables_t));return 0;}')
    import subprocess
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(L.Setup), C.sizeof(L.Axion), C.sizeof(L.TraceParams), C.sizeof(L.Summary),
                                     C.sizeof(L.SolarZone), C.sizeof(L.EmissionParams), C.sizeof(L.OpacityTables)]
    assert C.sizeof(L.Axion) == 208


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="only meaningful without a GPU")
def test_product_path_fails_loudly_without_gpu():
    # no CPU fallback: creating a context without a device is an error, not a silent oracle run
    lib = L.load_sart()
    h = C.c_void_p()
    rc = lib.sart_create(0, C.byref(h))
    assert rc == L.SART_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in lib.sart_last_error() or b"hip" in lib.sart_last_error().lower()
    with pytest.raises(L.SartError):
        sa.RayTracer(sa.initFullSetup(n_radii=50, n_energies=40, refl_n_angles=20, refl_n_energies=20))


def _build_c_host(tmp_path):
    import subprocess
    exe = str(tmp_path / "trace_axion_wrapper")
    lib_dir = os.path.join(ROOT, "solaraxionraytracing_amd")
    subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "integration", "trace_axion_wrapper.c"), "-L", lib_dir, "-lsart_host", "-lsart", "-lm",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    return exe


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="only meaningful without a GPU")
def test_c_host_builds_against_the_headers_and_fails_loudly_without_gpu(tmp_path):
    """integration/trace_axion_wrapper.c: the boundary from plain C (C11, -Werror) - headers and libraries alone."""
    import subprocess
    r = subprocess.run([_build_c_host(tmp_path), "1000"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_host_records_agree_with_histogram(tmp_path):
    """The drop-in call from a C program (no Python, no torch in the process): 2e5 records binned by the caller against the
    fused histogram of the same ray ids - counters exactly, flux and image to rounding (the program's own exit code)."""
    import json
    import subprocess
    r = subprocess.run([_build_c_host(tmp_path), "200000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    out, scan, ascan = lines[0], lines[1]["mass_scan"], lines[2]["angular_scan"]
    assert out["agree"] is True and out["passed"] > 10000 and out["build"] == L.build_id() and out["abi"] == 5
    assert abs(out["flux_records"] - out["flux_histogram"]) <= 1e-11 * out["flux_histogram"]
    # step 3b: sart_trace_records_passed gives the records filterIt(it.passed) keeps, byte for byte, and the three counts
    assert out["passed_only_records_agree"] is True
    # step 6 of the program: the fused mass scan (gas stage) against one traceAxionWrapper per mass, from plain C
    assert scan["agree"] is True and scan["masses"] == 5 and scan["max_rel_diff_to_per_mass_records"] <= 1e-9
    # step 7: the fused angular scan against one traceAxionWrapper per telescope angle
    assert ascan["agree"] is True and ascan["angles"] == 4 and ascan["max_rel_diff_to_per_angle_records"] <= 1e-9


def test_product_does_not_link_or_import_the_oracle():
    import subprocess
    for lib in ("libsart.so", "libsart_host.so"):
        out = subprocess.run(["ldd", os.path.join(ROOT, "solaraxionraytracing_amd", lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out
    for f in os.listdir(os.path.join(ROOT, "solaraxionraytracing_amd")):
        if f.endswith(".py"):
            assert "oracle" not in re.sub(r'""".*?"""', "", open(os.path.join(ROOT, "solaraxionraytracing_amd", f)).read(), flags=re.S).lower(), f


def test_setup_constants_match_reference_tables():
    # SURVEY Appendix A (raytracer.nim:1098-1155, 1256-1346, 1381-1407, 1464-1490, 248-272)
    s = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
    assert (s.magnet_B, s.magnet_radiusCB, s.magnet_lengthColdbore, s.magnet_lengthB, s.magnet_tGas) == (2.0, 500.0, 11300.0, 11000.0, 100.0)
    assert (s.pipe_cb_vt3_length, s.pipe_cb_vt3_radius, s.pipe_vt3_xrt_length, s.pipes_turned_deg) == (225.0, 370.0, 250.0, 0.0)
    assert s.n_shells == 58 and s.all_r1[0] == 153.118 and s.all_r1[57] == 349.996 and s.all_r1[33] == 247.2855
    assert s.all_angles_deg[0] == 0.29 and s.all_angles_deg[57] == 0.661 and s.all_thickness[0] == 0.468 and s.all_thickness[57] == 1.070
    assert s.l_mirror == 300.0 and s.hole_in_optics == 0.2 and s.number_of_holes == 1 and s.hole_type == L.HT_NONE
    assert s.distance_detector_xrt == 7500.0 and s.reflectivity_kind == L.RK_SINGLE_COATING
    assert s.theta_rad == pytest.approx(np.deg2rad(20.0)) and s.radius_window == 7.0 and s.depth_det == 30.0
    assert (s.distance_sun_earth, s.radius_sun, s.m_axion, s.g_agamma, s.chip_x_max) == (1.5e14, 6.9e11, 0.0853, 1e-12, 14.0)
    assert (s.test_active, s.test_parallel, s.test_distance, s.test_radius, s.test_energy) == (0, 1, 2000.0, 350.0, 0.021)

    c = sa.newFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, L.CF_XRAY_TEST)
    assert (c.magnet_B, c.magnet_radiusCB, c.magnet_lengthColdbore, c.magnet_lengthB, c.magnet_tGas) == (9.0, 21.5, 9756.0, 9260.0, 1.7)
    assert (c.pipe_cb_vt3_length, c.pipe_cb_vt3_radius, c.pipe_vt3_xrt_length, c.pipe_vt3_xrt_radius, c.pipes_turned_deg) == (127.66, 39.89, 111.7, 23.935, 2.75)
    assert c.n_shells == 14 and list(c.all_r1[:14])[::13] == [63.006, 105.632] and c.all_xsep[8] == 4.306 and c.all_angles_deg[7] == 0.767
    assert list(c.optics_entrance) == [-83.0, 0.0, 0.0] and c.l_mirror == 225.0 and c.hole_type == L.HT_CROSS
    assert c.reflectivity_kind == L.RK_MULTI_COATING and list(c.coating_layers[:4]) == [2, 5, 9, 14]
    assert c.distance_detector_xrt == 1485.0 and c.theta_rad == pytest.approx(np.deg2rad(30.0))
    assert (c.test_active, c.test_distance, c.test_radius, c.test_off_axis_up, c.test_length_col, c.test_energy) == (1, 100.0, 10.0, 200.0, 50.0, 1.0)

    a = sa.newFullSetup(L.ES_CAST, L.DK_INGRID2017, L.SK_VACUUM, L.TK_ABRIXAS)
    assert a.n_shells == 27 and a.all_r1[0] == 38.125 and a.all_r1[26] == 81.443 and a.all_angles_deg[26] == 0.7120
    assert list(a.optics_entrance) == [0.0, -60.0, 0.0] and a.l_mirror == 150.0 and a.distance_detector_xrt == 1600.0
    assert (a.pipe_cb_vt3_length, a.pipe_cb_vt3_radius, a.pipe_vt3_xrt_length, a.pipe_vt3_xrt_radius) == (114.3, 66.65, 171.43, 47.62)


def test_unsupported_telescopes_raise_like_the_reference():
    # doAssert for tkCustomBabyIAXO (raytracer.nim:1233) and tkOther (:1348); ValueError for bad enums (:1027-1030)
    for tk in (L.TK_CUSTOM_BABYIAXO, L.TK_OTHER):
        with pytest.raises(L.SartError) as e:
            sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, tk)
        assert e.value.code == -4
    with pytest.raises(L.SartError):
        sa.newFullSetup(7, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)


def test_config_overrides():
    # maybeParseMagnetConfig / TestXraySource / DetectorInstallation (raytracer.nim:1032-1096) with config_default.toml values
    m = L.MagnetConfig(2.0, 350.0, 11300.0, 11000.0, 1.0, 100.0)
    t = L.TestSourceConfig(1, 0, 1.0, 2000.0, 350.0, 0.0, 0.0, 0.125, 0.021)
    d = L.DetectorInstallConfig(1485.0, 0.0, 0.0, 0.0)
    s = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM, 0, m, t, d)
    assert s.magnet_radiusCB == 350.0 and s.test_active == 1 and s.test_parallel == 0 and s.test_length_col == 0.021
    assert s.distance_detector_xrt == 1485.0


def test_build_cdfs_matches_numpy():
    rng = np.random.default_rng(5)
    n_r, n_e = 37, 53
    em = rng.random((n_r, n_e)) + 0.01
    radii, energies = tables.solar_grid(n_r, n_e)
    rcdf, ecdf = tables.build_cdfs(em, radii, energies)
    diff = em * energies[None, :] ** 2 * radii[:, None] ** 2          # raytracer.nim:2686
    cs = np.cumsum(diff, axis=1)
    np.testing.assert_allclose(ecdf, cs / cs[:, -1:], rtol=1e-13)
    rs = np.cumsum(cs[:, -1])
    np.testing.assert_allclose(rcdf, rs / rs[-1], rtol=1e-13)
    assert np.all(ecdf[:, -1] == 1.0) and rcdf[-1] == 1.0 and np.all(np.diff(rcdf) >= 0)


def test_default_tables_are_valid():
    full = sa.initFullSetup()
    assert full.diffFluxCDFs.shape == (1968, 1500) and full.fluxRadiusCDF.shape == (1968,)
    assert np.all(full.diffFluxCDFs[:, -1] == 1.0) and np.all(np.diff(full.diffFluxCDFs, axis=1) >= 0)
    assert full.energies[0] == pytest.approx(1e-3) and full.energies[-1] == 15.0
    # solar core dominates: half of the flux from within ~0.15 R_sun
    assert 0.05 < 0.0015 + 0.0005 * np.searchsorted(full.fluxRadiusCDF, 0.5) < 0.25
    r = full.reflectivity
    assert r.data.shape == (1, 1000, 1000) and 0.0 <= r.data.min() and r.data.max() <= 1.0
    d = full.detector_tables
    assert d.x_kev[0] == 0.0 and d.x_kev[-1] == 15.0 and d.gas_x_kev[-1] == 15.0
    assert np.all((d.window >= 0) & (d.window <= 1)) and np.all((d.gas_absorption >= 0) & (d.gas_absorption <= 1))
    llnl = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, n_radii=60, n_energies=50,
                            refl_n_angles=30, refl_n_energies=30)
    assert llnl.reflectivity.data.shape[0] == 4


def test_solar_csv_reader_roundtrip(tmp_path):
    radii, energies = tables.solar_grid(5, 7)
    em = np.arange(35, dtype=float).reshape(5, 7) + 1
    p = tmp_path / "solar_model_dataframe.csv"
    with open(p, "w") as f:
        f.write("Radius,Energy [keV],emRates\n")
        for i, r in enumerate(radii):
            for j, e in enumerate(energies):
                f.write("%s,%s,%s\n" % (repr(float(r)), repr(float(e)), repr(float(em[i, j]))))
    r2, e2, em2 = tables.read_solar_model_csv(str(p))
    np.testing.assert_array_equal(r2, radii); np.testing.assert_array_equal(e2, energies); np.testing.assert_array_equal(em2, em)


def test_build_id_names_the_sources_the_library_was_built_from():
    """sart_build_id() = hash of the device sources + flags (csrc/Makefile: BUILD_ID).  bench.py reports counter-derived
    roofline figures only when the committed PMC profile carries the same id."""
    import subprocess
    import sys
    want = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "solaraxionraytracing_amd", "csrc"), "print-build-id"],
                          capture_output=True, text=True, check=True).stdout.strip()
    assert re.fullmatch(r"[0-9a-f]{12}-[0-9a-f]{6}", want)
    assert L.build_id() == want, "libsart.so is older than the sources: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
    sys.path.insert(0, ROOT)
    import bench
    pmc, why, lib_id = bench.pmc_for_this_build("babyiaxo_xmm")
    assert lib_id == want
    assert (pmc is not None and pmc["build_id"] == want) or (pmc is None and "build" in why)
    blk = bench.roofline_block("babyiaxo_xmm", 1e9, 0.0153, 20, {"N_PASSED_TILL_WINDOW": 2.2e8, "N_SHELL_SELECTED": 3.3e8}, 1e9)
    for k in ("hbm_frac", "hbm_achieved_gbs", "fabric_bytes_per_ray", "build_id", "pmc_build_id"):   # flat scalars
        assert k in blk and not isinstance(blk[k], (dict, list))
    if pmc is None:
        assert blk["achieved"] is None and blk["frac"] is None and blk["hbm_frac"] is None and blk["traffic"] is None
    # a profile published for another build is refused
    stale = dict(bench.load_pmc("babyiaxo_xmm") or {}, build_id="000000000000-000000")
    orig = bench.load_pmc
    bench.load_pmc = lambda w: stale
    try:
        assert bench.pmc_for_this_build("babyiaxo_xmm")[0] is None
    finally:
        bench.load_pmc = orig


def test_python_constants_match_the_device_header():
    """_lib.py repeats a few sizes of csrc/sart_device.h / include/sart.h (buffers that Python allocates for the library):
    they must be the header's."""
    from solaraxionraytracing_amd import _lib
    dev = open(os.path.join(ROOT, "solaraxionraytracing_amd", "csrc", "sart_device.h")).read()
    hdr = open(os.path.join(ROOT, "include", "sart.h")).read()

    def const(name, text=dev):
        m = re.search(r"constexpr\s+\w+\s+%s\s*=\s*([^;]+);" % name, text)
        assert m, name
        return m.group(1).strip()
    k_guide, k_top = int(const("kRadiusGuide")), int(const("kRadiusGuideTop"))
    assert const("kRadiusGuideEntries").replace(" ", "") == "kRadiusGuide+1+kRadiusGuideTop+1"
    assert _lib.RADIUS_GUIDE_ENTRIES == k_guide + 1 + k_top + 1 == 3074
    assert "radius_guide_out[%d]" % _lib.RADIUS_GUIDE_ENTRIES in hdr and "[%d]" % _lib.ENERGY_GUIDE_ENTRIES in hdr
    assert int(const("kImageTileMax")) ** 2 <= int(const("kTileRingCells")) + int(const("kTileExtraCells"))
    assert int(const("kImageTileExtraMax")) ** 2 <= int(const("kTileExtraCells"))
    assert _lib.SART_ACC_COUNT == int(re.search(r"SART_ACC_COUNT\s*=\s*(\d+)", hdr).group(1))
