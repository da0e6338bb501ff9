## sart_ffi.nim — complete Nim binding of `libsart.so` (include/sart.h) for jovoy/SolarAxionRayTracing.
##
## `include`d into src/raytracer.nim behind its type section (it reads the private fields of `ExperimentSetup`,
## `DetectorSetup`, `Telescope` and the other types declared at raytracer.nim:59-184 and the module constants of :248-272), it routes the
## one hot path — `traceAxionWrapper` (raytracer.nim:2223-2244) and the accumulation behind it — through the MI355X
## library.  Compile the reference with
##     nim c -d:release --passC:"-I<repo>/include" --passL:"-L<repo>/solaraxionraytracing_amd -lsart" src/raytracer.nim
##
## STATUS: no Nim compiler exists in the build image of this repository; the file is written against include/sart.h field
## by field (every `sart_setup_t` / `sart_trace_params_t` / `sart_summary_t` field is declared and assigned, no elisions)
## but has never been compiled.  The `static: doAssert sizeof` lines make a layout slip a compile-time error.
## The C++ (`csrc/raytracer_host.cpp`) and Python (`_lib.py`) bindings of the same header are compiled and tested.

const
  sartH = "sart.h"
  SartMaxShells = 64    # SART_MAX_SHELLS
  SartMaxCoatings = 8   # SART_MAX_COATINGS
  SartAccCount = 24     # SART_ACC_COUNT (ABI version 2)
  SartScanRow = 8       # SART_SCAN_ROW: 8-byte slots per row of a mass-scan accumulator
  SartAScanRow = 8      # SART_ASCAN_ROW: the same for the fused angular scan (ABI version 3)
  SartErrAccumulator* = -7   # SART_ERR_ACCUMULATOR: a raw FIXED64 accumulator wrapped / does not resolve the weights

type
  SartContext* {.importc: "sart_context", header: sartH, incompleteStruct.} = object

  ## sart_setup_t (include/sart.h): ExperimentSetup + DetectorSetup + module constants, flattened.
  SartSetup* {.importc: "sart_setup_t", header: sartH, bycopy.} = object
    experiment*, stage*, telescope_kind*, detector_kind*: int32
    # Magnet, raytracer.nim:83-89
    magnet_B*, magnet_lengthB*, magnet_lengthColdbore*, magnet_radiusCB*, magnet_pGasRoom*, magnet_tGas*: cdouble
    # Pipes, raytracer.nim:125-140
    pipe_cb_vt3_length*, pipe_cb_vt3_radius*, pipe_vt3_xrt_length*, pipe_vt3_xrt_radius*: cdouble
    pipes_turned_deg*, distance_cb_axis_xrt_axis*: cdouble
    # Telescope, raytracer.nim:91-105
    optics_entrance*, optics_exit*: array[3, cdouble]
    telescope_turned_x_deg*, telescope_turned_y_deg*: cdouble
    n_shells*, hole_type*, number_of_holes*, reflectivity_kind*: int32
    all_r1*, all_thickness*, all_xsep*, all_angles_deg*: array[SartMaxShells, cdouble]
    l_mirror*, hole_in_optics*: cdouble
    n_coatings*: int32
    coating_layers*: array[SartMaxCoatings, int32]
    # DetectorInstallation, raytracer.nim:146-153
    distance_detector_xrt*, distance_window_focal_plane*, lateral_shift*, transversal_shift*: cdouble
    # DetectorSetup, raytracer.nim:170-184
    radius_window*: cdouble
    number_of_strips*: int32
    pad0 {.importc: "_pad0".}: int32
    open_aperture_ratio*, strip_dist_window*, strip_width_window*, theta_rad*, depth_det*: cdouble
    # TestXraySource, raytracer.nim:108-122
    test_active*, test_parallel*: int32
    test_energy*, test_distance*, test_radius*, test_off_axis_up*, test_off_axis_left*, test_activity*, test_length_col*: cdouble
    # module constants, raytracer.nim:248-272
    distance_sun_earth*, radius_sun*, room_temp*, m_axion*, g_agamma*, chip_x_max*, chip_y_max*: cdouble

  ## sart_trace_params_t
  SartTraceParams* {.importc: "sart_trace_params_t", header: sartH, bycopy.} = object
    n_rays*, seed*, ray_id_offset*: uint64
    flags*: uint32
    image_nx*, image_ny*, accumulate*: int32
    image_x_min*, image_x_max*, image_y_min*, image_y_max*: cdouble
    spectra*, n_radial_bins*: int32
    radial_max*: cdouble

  ## sart_summary_t: the scalars behind the image (SART_ACC_* indices below)
  SartSummary* {.importc: "sart_summary_t", header: sartH, bycopy.} = object
    v*: array[SartAccCount, cdouble]

  ## sart_fixed_quanta_t: what one count of a raw SART_ACCUM_FIXED64 accumulator is worth
  SartFixedQuanta* {.importc: "sart_fixed_quanta_t", header: sartH, bycopy.} = object
    weight*, weight_sq*, position*, reflect*: cdouble

const   # accumulation modes (include/sart.h)
  AccumF64* = 0
  AccumFixed64* = 1

const   # SART_ACC_* (include/sart.h)
  AccSumWeights* = 0
  AccNPassed* = 1
  AccNPassedTillWindow* = 2
  AccNHitNickel* = 3
  AccSumX* = 4
  AccSumY* = 5
  AccSumR* = 6
  AccSumWeightsSq* = 7
  AccNRays* = 8
  AccNReachedTelescope* = 9
  AccNShellSelected* = 10
  AccNOutsideImage* = 11
  AccSumWeightsHi* = 12    # raw FIXED64 accumulators only: high limbs (units of 2^40 quanta)
  AccSumXHi* = 13
  AccSumYHi* = 14
  AccSumRHi* = 15

# ---- entry points (include/sart.h, in its order) ----------------------------------------------------------------------
proc sart_abi_version*(): cint {.importc, header: sartH.}
proc sart_last_error*(): cstring {.importc, header: sartH.}
proc sart_create*(deviceOrdinal: cint, ctx: ptr ptr SartContext): cint {.importc, header: sartH.}
proc sart_destroy*(ctx: ptr SartContext): cint {.importc, header: sartH.}
proc sart_set_stream*(ctx: ptr SartContext, hipStream: pointer): cint {.importc, header: sartH.}
proc sart_synchronize*(ctx: ptr SartContext): cint {.importc, header: sartH.}
proc sart_set_setup*(ctx: ptr SartContext, s: ptr SartSetup): cint {.importc, header: sartH.}
proc sart_get_setup*(ctx: ptr SartContext, s: ptr SartSetup): cint {.importc, header: sartH.}
proc sart_set_telescope_angles*(ctx: ptr SartContext, turnedXDeg, turnedYDeg: cdouble): cint {.importc, header: sartH.}
proc sart_set_axion_mass*(ctx: ptr SartContext, mAxionEv: cdouble): cint {.importc, header: sartH.}
proc sart_set_solar_tables*(ctx: ptr SartContext, fluxRadiusCdf, diffFluxCdfs, energiesKev: ptr cdouble,
                            nRadii, nEnergies: int32): cint {.importc, header: sartH.}
proc sart_set_solar_tables_device*(ctx: ptr SartContext, emRatesDevice: pointer, radii, energiesKev: ptr cdouble,
                                   nRadii, nEnergies: int32): cint {.importc, header: sartH.}
proc sart_get_solar_tables*(ctx: ptr SartContext, fluxRadiusCdfOut, diffFluxCdfsOut: ptr cdouble,
                            radiusGuideOut, energyGuideOut: ptr uint16): cint {.importc, header: sartH.}
proc sart_set_reflectivity*(ctx: ptr SartContext, nCoatings, nAngles, nEnergies: int32,
                            angleMinDeg, angleMaxDeg, energyMinKev, energyMaxKev: cdouble, data: ptr cdouble): cint {.importc, header: sartH.}
proc sart_set_detector_tables*(ctx: ptr SartContext, strongbackX, strongbackY: ptr cdouble, nStrongback: int32,
                               windowX, windowY: ptr cdouble, nWindow: int32,
                               gasAbsX, gasAbsY: ptr cdouble, nGasAbs: int32): cint {.importc, header: sartH.}
proc sart_trace_records*(ctx: ptr SartContext, p: ptr SartTraceParams, axBuf: pointer): cint {.importc, header: sartH.}
proc sart_trace_records_device*(ctx: ptr SartContext, p: ptr SartTraceParams, axBufDevice: pointer): cint {.importc, header: sartH.}
type SartRecordCounts* {.importc: "sart_record_counts_t", header: sartH, bycopy.} = object
  n_rays*, n_passed*, n_passed_till_window*, n_hit_nickel*: uint64
proc sart_trace_records_passed*(ctx: ptr SartContext, p: ptr SartTraceParams, axBuf: pointer, capacity: uint64,
                                counts: ptr SartRecordCounts): cint {.importc, header: sartH.}
proc sart_trace_records_passed_device*(ctx: ptr SartContext, p: ptr SartTraceParams, axBufDevice: pointer, capacity: uint64,
                                       countsDevice: pointer): cint {.importc, header: sartH.}
proc sart_release_scratch*(ctx: ptr SartContext): cint {.importc, header: sartH.}
proc sart_trace_histogram_device*(ctx: ptr SartContext, p: ptr SartTraceParams, accumulatorDevice: ptr cdouble): cint {.importc, header: sartH.}
proc sart_trace_histogram*(ctx: ptr SartContext, p: ptr SartTraceParams, imageOutHost: ptr cdouble, summary: ptr SartSummary): cint {.importc, header: sartH.}
proc sart_trace_histogram_spectra*(ctx: ptr SartContext, p: ptr SartTraceParams, imageOutHost: ptr cdouble, summary: ptr SartSummary,
                                   spectraOutHost: ptr cdouble): cint {.importc, header: sartH.}
proc sart_set_accumulation_mode*(ctx: ptr SartContext, mode, headroomBits: cint): cint {.importc, header: sartH.}
proc sart_get_accumulation_mode*(ctx: ptr SartContext, modeOut: ptr cint): cint {.importc, header: sartH.}
proc sart_get_fixed_quanta*(ctx: ptr SartContext, quanta: ptr SartFixedQuanta): cint {.importc, header: sartH.}
proc sart_finalize_accumulator_device*(ctx: ptr SartContext, p: ptr SartTraceParams, accFixedDevice: pointer,
                                       outF64Device: ptr cdouble): cint {.importc, header: sartH.}
## long FIXED64 accumulations: a second limb per slot (value = (hi 2^40 + lo) quantum), folded between launches
proc sart_rollover_accumulator_device*(ctx: ptr SartContext, p: ptr SartTraceParams, accFixedDevice: pointer,
                                       hiLimbsDevice: pointer): cint {.importc, header: sartH.}
proc sart_finalize_accumulator_limbs_device*(ctx: ptr SartContext, p: ptr SartTraceParams, accFixedDevice: pointer,
                                             hiLimbsDevice: pointer, outF64Device: ptr cdouble): cint {.importc, header: sartH.}
## fused axion-mass scan (gas stage): every ray traced once, weighed for nMasses masses; (nMasses + 1) rows of SartScanRow slots
proc sart_trace_mass_scan_device*(ctx: ptr SartContext, p: ptr SartTraceParams, massesEv: ptr cdouble, nMasses: int32,
                                  scanAccDevice: ptr cdouble): cint {.importc, header: sartH.}
proc sart_trace_mass_scan*(ctx: ptr SartContext, p: ptr SartTraceParams, massesEv: ptr cdouble, nMasses: int32,
                           scanOutHost: ptr cdouble): cint {.importc, header: sartH.}
proc sart_finalize_mass_scan_device*(ctx: ptr SartContext, p: ptr SartTraceParams, massesEv: ptr cdouble, nMasses: int32,
                                     scanFixedDevice: pointer, outF64Device: ptr cdouble): cint {.importc, header: sartH.}
## fused angular scan: every ray sampled and cut once, turned through nAngles telescope angles; (nAngles + 1) rows of SartAScanRow slots
proc sart_trace_angular_scan_device*(ctx: ptr SartContext, p: ptr SartTraceParams, turnedYDeg: ptr cdouble, nAngles: int32,
                                     scanAccDevice: ptr cdouble): cint {.importc, header: sartH.}
proc sart_trace_angular_scan*(ctx: ptr SartContext, p: ptr SartTraceParams, turnedYDeg: ptr cdouble, nAngles: int32,
                              scanOutHost: ptr cdouble): cint {.importc, header: sartH.}
proc sart_finalize_angular_scan_device*(ctx: ptr SartContext, p: ptr SartTraceParams, nAngles: int32, scanFixedDevice: pointer,
                                        outF64Device: ptr cdouble): cint {.importc, header: sartH.}
proc sart_reduce_across_devices*(contexts: ptr ptr SartContext, accumulatorsDevice: ptr ptr cdouble, n: int32, nDoubles: csize_t,
                                 root: int32): cint {.importc, header: sartH.}
proc sart_enable_kernel_timing*(ctx: ptr SartContext, enable: cint): cint {.importc, header: sartH.}
proc sart_get_kernel_timing*(ctx: ptr SartContext, totalMs: ptr cdouble, nLaunches: ptr int64): cint {.importc, header: sartH.}
proc sart_build_id*(): cstring {.importc, header: sartH.}
proc sart_device_info*(ctx: ptr SartContext, nCu, waveSize: ptr int32, nameBuf: cstring, nameBufLen: csize_t): cint {.importc, header: sartH.}

static:
  doAssert sizeof(Axion) == 208            # the record layout the library writes (raytracer.nim:192-221; SURVEY 8a a1)
  doAssert sizeof(SartSetup) == 2504
  doAssert sizeof(SartTraceParams) == 88
  doAssert sizeof(SartSummary) == 8 * SartAccCount

template sartCheck*(rc: cint) =
  if rc != 0: raise newException(IOError, "sart: " & $sart_last_error())

# ---- ExperimentSetup + DetectorSetup + module constants -> sart_setup_t, one assignment per field ------------------------
proc toSartSetup*(e: ExperimentSetup, d: DetectorSetup): SartSetup =
  result.experiment = e.kind.ord.int32                       # ExperimentSetupKind :16-18
  result.stage = e.stage.ord.int32                           # StageKind :39-41
  result.telescope_kind = e.telescope.kind.ord.int32         # TelescopeKind :31-37
  result.detector_kind = (case d.windowYear                  # informational only
                          of wy2017: 0'i32
                          of wy2018: 1'i32
                          of wyIAXO: 2'i32)
  # Magnet :83-89
  result.magnet_B = e.magnet.B.float
  result.magnet_lengthB = e.magnet.lengthB.float
  result.magnet_lengthColdbore = e.magnet.lengthColdbore.float
  result.magnet_radiusCB = e.magnet.radiusCB.float
  result.magnet_pGasRoom = e.magnet.pGasRoom.float
  result.magnet_tGas = e.magnet.tGas.float
  # Pipes :125-140
  result.pipe_cb_vt3_length = e.pipes.coldBoreToVT3.length.float
  result.pipe_cb_vt3_radius = e.pipes.coldBoreToVT3.radius.float
  result.pipe_vt3_xrt_length = e.pipes.vt3ToXRT.length.float
  result.pipe_vt3_xrt_radius = e.pipes.vt3ToXRT.radius.float
  result.pipes_turned_deg = e.pipes.pipesTurned.float
  result.distance_cb_axis_xrt_axis = e.pipes.distanceCBAxisXRTAxis.float
  # Telescope :91-105
  let tel = e.telescope
  doAssert tel.optics_entrance.len == 3 and tel.optics_exit.len == 3
  for i in 0 ..< 3:
    result.optics_entrance[i] = tel.optics_entrance[i].float
    result.optics_exit[i] = tel.optics_exit[i].float
  result.telescope_turned_x_deg = tel.telescope_turned_x.float
  result.telescope_turned_y_deg = tel.telescope_turned_y.float
  let nShells = tel.allR1.len
  doAssert nShells <= SartMaxShells and tel.allThickness.len == nShells and tel.allXsep.len == nShells and tel.allAngles.len == nShells
  result.n_shells = nShells.int32
  result.hole_type = tel.holeType.ord.int32                  # HoleType :23-29
  result.number_of_holes = tel.numberOfHoles.int32
  result.reflectivity_kind = tel.reflectivity.kind.ord.int32 # ReflectivityKind :59-64
  for i in 0 ..< nShells:
    result.all_r1[i] = tel.allR1[i].float
    result.all_thickness[i] = tel.allThickness[i].float
    result.all_xsep[i] = tel.allXsep[i].float
    result.all_angles_deg[i] = tel.allAngles[i].float
  result.l_mirror = tel.lMirror.float
  result.hole_in_optics = tel.holeInOptics.float
  case tel.reflectivity.kind                                 # Reflectivity.layers :78-81
  of rkMultiCoating:
    doAssert tel.reflectivity.layers.len <= SartMaxCoatings
    result.n_coatings = tel.reflectivity.layers.len.int32
    for i, l in tel.reflectivity.layers: result.coating_layers[i] = l.int32
  else:
    result.n_coatings = 1
    result.coating_layers[0] = nShells.int32
  # DetectorInstallation :146-153
  result.distance_detector_xrt = e.detectorInstall.distanceDetectorXRT.float
  result.distance_window_focal_plane = e.detectorInstall.distanceWindowFocalPlane.float
  result.lateral_shift = e.detectorInstall.lateralShift.float
  result.transversal_shift = e.detectorInstall.transversalShift.float
  # DetectorSetup :170-184
  result.radius_window = d.radiusWindow.float
  result.number_of_strips = d.numberOfStrips.int32
  result.open_aperture_ratio = d.openApertureRatio
  result.strip_dist_window = d.stripDistWindow.float
  result.strip_width_window = d.stripWidthWindow.float
  result.theta_rad = d.theta.float
  result.depth_det = d.depthDet.float
  # TestXraySource :108-122
  result.test_active = e.testSource.active.int32
  result.test_parallel = e.testSource.parallel.int32
  result.test_energy = e.testSource.energy.float
  result.test_distance = e.testSource.distance.float
  result.test_radius = e.testSource.radius.float
  result.test_off_axis_up = e.testSource.offAxisUp.float
  result.test_off_axis_left = e.testSource.offAxisLeft.float
  result.test_activity = e.testSource.activity.float
  result.test_length_col = e.testSource.lengthCol.float
  # module constants :248-272
  result.distance_sun_earth = DistanceSunEarth.float
  result.radius_sun = RadiusSun.float
  result.room_temp = RoomTemp.float
  result.m_axion = mAxion
  result.g_agamma = g_aγ.float
  result.chip_x_max = ChipXMax.float
  result.chip_y_max = ChipYMax.float

# ---- the raw tables the library interpolates itself ----------------------------------------------------------------------
## The reference keeps only the interpolators (numericalnim closures); the library needs the raw columns / tensors they were
## built from.  Keep them next to the interpolators where they are read:
##   initReflectivity (:1174-1231): `reflectivity` tensors [angle][energy] per coating + (min, max) of `Angles` / `Energy`
##   newDetectorSetup (:1499-1527): the (energy / keV, transmission) columns of the three 1-D tables
type
  SartRawTables* = object
    reflData*: seq[cdouble]            # [coating][angle][energy], row-major
    reflNC*, reflNA*, reflNE*: int32
    reflAMin*, reflAMax*, reflEMin*, reflEMax*: cdouble
    sbX*, sbY*, winX*, winY*, gasX*, gasY*: seq[cdouble]

proc sartUpload*(ctx: ptr SartContext, s: FullRaytraceSetup, raw: var SartRawTables) =
  ## Everything `traceAxionWrapper` captures (:2223-2232), once per setup.
  var setup = toSartSetup(s.expSetup, s.detectorSetup)
  sartCheck sart_set_setup(ctx, addr setup)
  let nR = s.diffFluxCDFs.len
  let nE = s.energies.len
  var rcdf = newSeq[cdouble](nR)
  var ecdf = newSeq[cdouble](nR * nE)                     # row-major [radius][energy]
  var energies = newSeq[cdouble](nE)
  for r in 0 ..< nR:
    rcdf[r] = s.fluxRadiusCDF[r]
    doAssert s.diffFluxCDFs[r].len == nE
    for i in 0 ..< nE: ecdf[r * nE + i] = s.diffFluxCDFs[r][i]
  for i in 0 ..< nE: energies[i] = s.energies[i].float
  sartCheck sart_set_solar_tables(ctx, addr rcdf[0], addr ecdf[0], addr energies[0], nR.int32, nE.int32)
  sartCheck sart_set_reflectivity(ctx, raw.reflNC, raw.reflNA, raw.reflNE, raw.reflAMin, raw.reflAMax, raw.reflEMin, raw.reflEMax,
                                  addr raw.reflData[0])
  sartCheck sart_set_detector_tables(ctx, addr raw.sbX[0], addr raw.sbY[0], raw.sbX.len.int32,
                                     addr raw.winX[0], addr raw.winY[0], raw.winX.len.int32,
                                     addr raw.gasX[0], addr raw.gasY[0], raw.gasX.len.int32)

# ---- the replacement of traceAxionWrapper (raytracer.nim:2223-2244) ------------------------------------------------------
proc sartParams*(nRays: int, flags: set[ConfigFlags], seed = 299792458'u64, rayIdOffset = 0'u64): SartTraceParams =
  result.n_rays = nRays.uint64
  result.seed = seed                        # replaces randomize(299792458) (:276): Philox key; counter = global ray id
  result.ray_id_offset = rayIdOffset
  result.flags = cast[uint8](flags).uint32  # set[ConfigFlags] (:223-230) is a bitset, bit i = i-th enum value = SART_CF_*
  result.image_nx = 256; result.image_ny = 256          # heatmaptable2 (:2629)
  result.accumulate = 0
  result.image_x_min = ChipXMin.float; result.image_x_max = ChipXMax.float   # :2622-2625
  result.image_y_min = ChipYMin.float; result.image_y_max = ChipYMax.float
  result.spectra = 0; result.n_radial_bins = 0; result.radial_max = 0.0

proc traceAxionWrapperGpu*(ctx: ptr SartContext, axBuf: ptr UncheckedArray[Axion], bufLen: int, flags: set[ConfigFlags],
                           seed = 299792458'u64, rayIdOffset = 0'u64) =
  ## Literal drop-in: `bufLen` Axion records into the caller's buffer.
  var p = sartParams(bufLen, flags, seed, rayIdOffset)
  sartCheck sart_trace_records(ctx, addr p, axBuf)

proc traceAxionsPassedGpu*(ctx: ptr SartContext, bufLen: int, flags: set[ConfigFlags], counts: var SartRecordCounts,
                           seed = 299792458'u64, rayIdOffset = 0'u64, capacity = 0): seq[Axion] =
  ## `axions.filterIt(it.passed)` of the buffer traceAxionWrapper would fill (what generateResultPlots :2252 and the scan
  ## sum :2800 go on with), in ray order, and the counts echoed at :2253-2257 - without the records of the other rays
  ## crossing PCIe.  `capacity` = records to make room for (0: bufLen, always enough; a caller that knows its pass
  ## fraction allocates less and checks counts.n_passed <= capacity).
  let cap = if capacity > 0: capacity else: bufLen
  result = newSeq[Axion](cap)
  var p = sartParams(bufLen, flags, seed, rayIdOffset)
  let buf: pointer = if cap > 0: cast[pointer](addr result[0]) else: nil
  sartCheck sart_trace_records_passed(ctx, addr p, buf, cap.uint64, addr counts)
  result.setLen(min(counts.n_passed.int, cap))

proc traceHistogramGpu*(ctx: ptr SartContext, nRays: int, flags: set[ConfigFlags], image: var seq[cdouble],
                        seed = 299792458'u64, rayIdOffset = 0'u64): SartSummary =
  ## traceAxionWrapper + prepareHeatmap(256, 256) (:2629) + the flux sum (:2800) + the counters of :2252-2257 without a
  ## seq[Axion]: the form to use for N >> 1e6.  `image` is [y][x] row-major, 256 x 256.
  var p = sartParams(nRays, flags, seed, rayIdOffset)
  image.setLen(256 * 256)
  sartCheck sart_trace_histogram(ctx, addr p, addr image[0], addr result)

proc performAngularScanGpu*(ctx: ptr SartContext, angles: seq[float], nRaysPerAngle: int, flags: set[ConfigFlags]): seq[float] =
  ## performAngularScan (:2778-2802): one full run per angle, flux = sum of weights of the passed rays.
  var setup: SartSetup
  sartCheck sart_get_setup(ctx, addr setup)
  for i, a in angles:
    sartCheck sart_set_telescope_angles(ctx, NaN, a)       # tel.telescope_turned_y = angle (:2796)
    var p = sartParams(nRaysPerAngle, flags, rayIdOffset = (i * nRaysPerAngle).uint64)
    p.image_nx = 0; p.image_ny = 0                         # flux-only launch: the scan reads the sum of the weights alone (:2800)
    var s: SartSummary
    sartCheck sart_trace_histogram(ctx, addr p, nil, addr s)
    result.add s.v[AccSumWeights]
  sartCheck sart_set_telescope_angles(ctx, NaN, setup.telescope_turned_y_deg)   # the reference scans a copy (:2794)

proc performAngularScanFusedGpu*(ctx: ptr SartContext, angles: seq[float], nRays: int, flags: set[ConfigFlags],
                                 seed = 299792458'u64, rayIdOffset = 0'u64): tuple[flux, fluxSq, nPassed: seq[float]] =
  ## The same curve in ONE pass over the rays (sart_trace_angular_scan): the telescope's angle enters a ray at the
  ## transformation into the telescope's frame (:1878-1899) and nowhere before it, so every ray is sampled and taken through
  ## bore and pipes once and turned through every angle.  All angles see the same rays; the context's setup is not changed.
  doAssert angles.len > 0
  var p = sartParams(nRays, flags, seed, rayIdOffset)
  var a = newSeq[cdouble](angles.len)
  for i, x in angles: a[i] = x
  var rows = newSeq[cdouble]((angles.len + 1) * SartAScanRow)
  sartCheck sart_trace_angular_scan(ctx, addr p, addr a[0], a.len.int32, addr rows[0])
  for k in 0 ..< angles.len:
    result.flux.add rows[k * SartAScanRow + 0]       # SART_ASCAN_SUM_WEIGHTS (= fluxes[i] of :2800)
    result.fluxSq.add rows[k * SartAScanRow + 1]     # SART_ASCAN_SUM_WEIGHTS_SQ
    result.nPassed.add rows[k * SartAScanRow + 2]    # SART_ASCAN_N_PASSED

proc performAxionMassScanGpu*(ctx: ptr SartContext, massesEv: seq[float], nRays: int, flags: set[ConfigFlags],
                               seed = 299792458'u64, rayIdOffset = 0'u64): tuple[flux, fluxSq, nPassed: seq[float]] =
  ## Axion-mass scan in the gas stage (stageSetup = "gas"): the reference would change `mAxion` (:255) and re-run
  ## calculateFluxFractions per mass; here every ray is traced ONCE and weighed for every mass (sart_trace_mass_scan: the mass
  ## enters a ray's weight through axionConversionProb2 alone, :1599-1625).  All masses see the same rays.
  ## flux[k] = sum of weights of the passed rays for massesEv[k]; sqrt(fluxSq[k]) = its Monte-Carlo error.
  doAssert massesEv.len > 0
  var p = sartParams(nRays, flags, seed, rayIdOffset)
  var m = newSeq[cdouble](massesEv.len)
  for i, x in massesEv: m[i] = x
  var rows = newSeq[cdouble]((massesEv.len + 1) * SartScanRow)
  sartCheck sart_trace_mass_scan(ctx, addr p, addr m[0], m.len.int32, addr rows[0])
  for k in 0 ..< massesEv.len:
    result.flux.add rows[k * SartScanRow + 0]        # SART_SCAN_SUM_WEIGHTS
    result.fluxSq.add rows[k * SartScanRow + 1]      # SART_SCAN_SUM_WEIGHTS_SQ
    result.nPassed.add rows[k * SartScanRow + 2]     # SART_SCAN_N_PASSED

proc traceHistogramDeterministic*(ctx: ptr SartContext, nRays: int, flags: set[ConfigFlags], image: var seq[cdouble],
                                  seed = 299792458'u64, rayIdOffset = 0'u64): SartSummary =
  ## The same image and sums, bit for bit reproducible: integer accumulation (SART_ACCUM_FIXED64) instead of f64 atomics - the
  ## result does not depend on the number of GPUs, on the replica layout or on how the rays are split over calls.  The call
  ## returns doubles like traceHistogramGpu (the library converts the integer accumulator itself).
  sartCheck sart_set_accumulation_mode(ctx, AccumFixed64.cint, 0)
  result = traceHistogramGpu(ctx, nRays, flags, image, seed, rayIdOffset)
  sartCheck sart_set_accumulation_mode(ctx, AccumF64.cint, 0)

proc sartUploadSolarTablesFromDevice*(ctx: ptr SartContext, emRatesDevice: pointer, radii, energies: seq[float]) =
  ## The CDF loops of initFullSetup (:2670-2705) on an emission table that already lives on the GPU (sart_emission_table_device):
  ## cumulative sums in the reference's order, normalisation and the library's guide tables, built on the device.
  var r = newSeq[cdouble](radii.len)
  var e = newSeq[cdouble](energies.len)
  for i, x in radii: r[i] = x
  for i, x in energies: e[i] = x
  sartCheck sart_set_solar_tables_device(ctx, emRatesDevice, addr r[0], addr e[0], r.len.int32, e.len.int32)

## calculateFluxFractions (:2755-2776) then reads:
##   var ctx: ptr SartContext
##   sartCheck sart_create(0, addr ctx)
##   sartUpload(ctx, raytraceSetup, rawTables)
##   traceAxionWrapperGpu(ctx, cast[ptr UncheckedArray[Axion]](axions[0].addr), NumberOfPointsSun, raytraceSetup.flags)
## in place of lines 2763-2772 (init(Weave) to exit(Weave)); generateResultPlots is unchanged.
