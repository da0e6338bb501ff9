/*
 * sart.h — C-ABI of the MI355X-native solar-axion ray tracer ("sart").
 *
 * This is the drop-in boundary for ONE hot path of jovoy/SolarAxionRayTracing:
 * the per-ray loop `traceAxionWrapper` -> `traceAxion`
 * (reference src/raytracer.nim:2223-2244 and :1736-2221) plus the accumulation
 * that follows it (`prepareHeatmap` :818-842, flux sum :2800).
 *
 * Plain C: extern "C", PODs, pointers and sizes only.  Every entry point
 * returns 0 on success and a negative sart_status on error; the message of the
 * last error on the calling thread is available from sart_last_error().
 * A context is bound to one GPU; it is not thread-safe; distinct contexts are
 * independent.  All calls are blocking unless stated otherwise.
 *
 * All floating point is IEEE f64, as in the reference.  Lengths are mm, angles
 * degrees unless the field name says otherwise, energies keV.
 */
#ifndef SART_H
#define SART_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SART_ABI_VERSION 5   /* 2: SART_ACC_COUNT 16 -> 24 (SUM_WEIGHTS_SQ_HI), fused mass scan, FIXED64 status;
                                3: SART_ERR_ACCUMULATOR, fused angular scan, flux-only launches, accumulator roll-over;
                                4: sart_trace_records_passed;
                                5: sart_release_scratch (additive); the (seed, ray id) -> uniforms mapping changed
                                   (Philox4x32-7, one block per ray): a v4 caller runs unchanged and gets other rays */
#define SART_MAX_SHELLS 64
#define SART_MAX_COATINGS 8

typedef enum sart_status {
  SART_OK = 0,
  SART_ERR_INVALID_ARGUMENT = -1,
  SART_ERR_NO_DEVICE = -2,      /* no HIP device / HIP runtime error             */
  SART_ERR_NOT_READY = -3,      /* setup or tables missing                       */
  SART_ERR_UNSUPPORTED = -4,    /* e.g. telescope kind the reference asserts on  */
  SART_ERR_OUT_OF_MEMORY = -5,
  SART_ERR_INTERNAL = -6,
  SART_ERR_ACCUMULATOR = -7     /* a raw SART_ACCUM_FIXED64 accumulator no longer means what it should: a slot wrapped (or is
                                   about to), or its quanta do not resolve the weights ("accumulation mode" below) - the
                                   arguments of the call that reports it are fine */
} sart_status;

/* ---- enums: numeric values follow the declaration order of the reference ---- */

/* ExperimentSetupKind, raytracer.nim:16-18 */
enum { SART_ES_CAST = 0, SART_ES_BABYIAXO = 1 };
/* StageKind, raytracer.nim:39-41 */
enum { SART_SK_VACUUM = 0, SART_SK_GAS = 1 };
/* TelescopeKind, raytracer.nim:31-37 */
enum { SART_TK_LLNL = 0, SART_TK_XMM = 1, SART_TK_CUSTOM_BABYIAXO = 2,
       SART_TK_ABRIXAS = 3, SART_TK_OTHER = 4 };
/* DetectorSetupKind, raytracer.nim:164-167 */
enum { SART_DK_INGRID2017 = 0, SART_DK_INGRID2018 = 1, SART_DK_INGRIDIAXO = 2 };
/* HoleType, raytracer.nim:23-29 */
enum { SART_HT_NONE = 0, SART_HT_CROSS = 1, SART_HT_STAR = 2, SART_HT_CIRCLE = 3,
       SART_HT_SQUARE = 4, SART_HT_DIAMOND = 5 };
/* ReflectivityKind, raytracer.nim:59-64 */
enum { SART_RK_EFFECTIVE_AREA = 0, SART_RK_SINGLE_COATING = 1, SART_RK_MULTI_COATING = 2 };
/* MaterialKind, raytracer.nim:186-189 */
enum { SART_MK_SI3N4 = 0, SART_MK_SI = 1, SART_MK_AR = 2 };

/* ConfigFlags, raytracer.nim:223-230: bit i <=> i-th enum value (a Nim set[ConfigFlags]) */
enum {
  SART_CF_IGNORE_DET_WINDOW = 1u << 0,
  SART_CF_IGNORE_GAS_ABS = 1u << 1,
  SART_CF_IGNORE_REFLECTION = 1u << 2,
  SART_CF_IGNORE_CONV_PROB = 1u << 3,
  SART_CF_XRAY_TEST = 1u << 4,
  SART_CF_READ_MAGNET_CONFIG = 1u << 5,
  SART_CF_READ_DET_INSTALL_CONFIG = 1u << 6
};

/*
 * sart_setup_t — everything `traceAxion` reads from `ExperimentSetup`,
 * `DetectorSetup`, `CenterVectors` and the module-level constants, flattened.
 * Replaces: ExperimentSetup raytracer.nim:155-162 (Magnet :83-89, Telescope
 * :91-105, TestXraySource :108-122, Pipes :131-140, DetectorInstallation
 * :146-153), DetectorSetup :170-184 and the `let` constants :248-272.
 * The centre vectors of :278-320 are functions of these fields and are derived
 * by the callee.
 */
typedef struct sart_setup_t {
  int32_t experiment;      /* SART_ES_*   */
  int32_t stage;           /* SART_SK_*   */
  int32_t telescope_kind;  /* SART_TK_*   */
  int32_t detector_kind;   /* SART_DK_*   (informational; theta etc. below are what is used) */

  /* Magnet, raytracer.nim:83-89 */
  double magnet_B;              /* T   */
  double magnet_lengthB;        /* mm  */
  double magnet_lengthColdbore; /* mm  */
  double magnet_radiusCB;       /* mm  */
  double magnet_pGasRoom;       /* bar */
  double magnet_tGas;           /* K   */

  /* Pipes, raytracer.nim:125-140 */
  double pipe_cb_vt3_length, pipe_cb_vt3_radius;
  double pipe_vt3_xrt_length, pipe_vt3_xrt_radius;
  double pipes_turned_deg;
  double distance_cb_axis_xrt_axis;

  /* Telescope, raytracer.nim:91-105 */
  double optics_entrance[3];
  double optics_exit[3];
  double telescope_turned_x_deg;
  double telescope_turned_y_deg;
  int32_t n_shells;
  int32_t hole_type;        /* SART_HT_* */
  int32_t number_of_holes;
  int32_t reflectivity_kind; /* SART_RK_* */
  double all_r1[SART_MAX_SHELLS];
  double all_thickness[SART_MAX_SHELLS];
  double all_xsep[SART_MAX_SHELLS];
  double all_angles_deg[SART_MAX_SHELLS];
  double l_mirror;
  double hole_in_optics;
  /* Reflectivity.layers, raytracer.nim:78-81, :1167 ("boundary layers") */
  int32_t n_coatings;
  int32_t coating_layers[SART_MAX_COATINGS];

  /* DetectorInstallation, raytracer.nim:146-153 */
  double distance_detector_xrt;
  double distance_window_focal_plane;
  double lateral_shift;
  double transversal_shift;

  /* DetectorSetup, raytracer.nim:170-184 */
  double radius_window;
  int32_t number_of_strips;
  int32_t _pad0;
  double open_aperture_ratio;
  double strip_dist_window;
  double strip_width_window;
  double theta_rad;
  double depth_det;

  /* TestXraySource, raytracer.nim:108-122 */
  int32_t test_active;
  int32_t test_parallel;
  double test_energy;       /* keV */
  double test_distance;
  double test_radius;
  double test_off_axis_up;
  double test_off_axis_left;
  double test_activity;     /* GBq */
  double test_length_col;

  /* module constants, raytracer.nim:248-272 — runtime parameters here */
  double distance_sun_earth; /* 1.5e14 mm  */
  double radius_sun;         /* 6.9e11 mm  */
  double room_temp;          /* 293.15 K   */
  double m_axion;            /* eV         */
  double g_agamma;           /* GeV^-1     */
  double chip_x_max;         /* 14 mm (ChipXMax; ChipCenterX = chip_x_max/2) */
  double chip_y_max;
} sart_setup_t;

/*
 * sart_axion_t — binary image of the reference's `Axion` object
 * (raytracer.nim:192-221; a Nim object is a C struct in declaration order):
 * 3 bools, 23 f64, 2 one-byte enums, 1 int => 208 bytes.
 */
typedef struct sart_axion_t {
  uint8_t passed;
  uint8_t passedTillWindow;
  uint8_t hitNickel;
  uint8_t _pad0[5];
  double pointdataX;
  double pointdataY;
  double pointdataXBefore;
  double pointdataYBefore;
  double pointdataR;
  double weights;
  double weightsAll;
  double transmissionMagnet;
  double yawAngles;
  double pixvalsX;
  double pixvalsY;
  double radii;
  double energiesAx;
  double energiesAxAll;
  double energiesAxWindow;
  uint8_t kinds;
  uint8_t kindsWindow;
  uint8_t _pad1[6];
  double transProbWindow;
  double transProbArgon;
  double transProbDetector;
  double transProbMagnet;
  double deviationDet;
  int64_t shellNumber;
  double energiesPre;
  double emratesPre;
  double reflect;
} sart_axion_t;

/* Parameters of one trace call. */
typedef struct sart_trace_params_t {
  uint64_t n_rays;         /* bufLen of traceAxionWrapper                               */
  uint64_t seed;           /* Philox4x32-7 key (replaces randomize(299792458), :276); the six uniforms of a ray are a
                              function of (seed, global ray id) only, see oracle/sart_oracle.c: sart_oracle_uniforms.  The
                              mapping changed in ABI 5 (one seven-round block per ray, 32 / 21 / 22-bit fractions): the same
                              (seed, id) names other rays than under ABI <= 4 - statistically equivalent, not identical */
  uint64_t ray_id_offset;  /* global id of ray 0 of this call: counter = offset + i      */
  uint32_t flags;          /* SART_CF_* bitset                                          */
  int32_t image_nx;        /* columns (x bins) of the focal-plane image, 256 in :2629    */
  int32_t image_ny;        /* rows    (y bins).  image_nx == image_ny == 0: FLUX-ONLY launch - no image is accumulated (no
                              pixel atomics, no LDS tile, no pilot launch that places it), the accumulator is the
                              SART_ACC_COUNT scalars (+ spectra) alone and every passed ray counts as outside the image
                              (N_OUTSIDE_IMAGE = N_PASSED): what a scan that reads SUM_WEIGHTS alone needs (:2800) */
  int32_t accumulate;      /* histogram mode: 0 = zero the accumulator first, 1 = add    */
  double image_x_min, image_x_max;  /* 0 .. ChipXMax in :2622-2625 */
  double image_y_min, image_y_max;
  /* Optional post-processing histograms (generateResultPlots): 0 = image + scalars only. */
  int32_t spectra;         /* 1: also accumulate the radial and per-energy histograms described below          */
  int32_t n_radial_bins;   /* bins of pointdataR over [0, radial_max): the reference bins at 0.001 mm (:2386)   */
  double radial_max;       /* mm                                                                                */
} sart_trace_params_t;

/*
 * Layout of the fused f64 accumulator a histogram trace adds into
 * (one buffer so that a multi-GPU reduce is a single RCCL call):
 *   [0, nx*ny)                     image, row-major img[y][x]  (prepareHeatmap :838-842, norm = 1)
 *   [nx*ny + SART_ACC_*]           scalars below
 */
enum {
  SART_ACC_SUM_WEIGHTS = 0,   /* sum of weights over passed rays (:2800)                 */
  SART_ACC_N_PASSED = 1,      /* axions.filterIt(it.passed).len (:2252)                  */
  SART_ACC_N_PASSED_TILL_WINDOW = 2, /* :2254 */
  SART_ACC_N_HIT_NICKEL = 3,  /* :2256 */
  SART_ACC_SUM_X = 4,         /* sums of pointdataX/Y/R over passed rays (means :2276-2278) */
  SART_ACC_SUM_Y = 5,
  SART_ACC_SUM_R = 6,
  SART_ACC_SUM_WEIGHTS_SQ = 7, /* for the Monte-Carlo error of the flux                   */
  SART_ACC_N_RAYS = 8,        /* rays traced into this accumulator                        */
  SART_ACC_N_REACHED_TELESCOPE = 9, /* survived bore + pipes (:1813-1868)                 */
  SART_ACC_N_SHELL_SELECTED = 10,   /* survived opaque structures + shell selection (:1910-1957) */
  SART_ACC_N_OUTSIDE_IMAGE = 11,    /* passed rays whose (x,y) fall outside the image range */
  /* SART_ACCUM_FIXED64 only (raw accumulators; 0 in every f64 accumulator): high limbs of the four sums that every
   * passed ray adds to (and, slot 16, of the sum of squared weights), in units of 2^40 quanta - see "accumulation mode" below */
  SART_ACC_SUM_WEIGHTS_HI = 12,
  SART_ACC_SUM_X_HI = 13,
  SART_ACC_SUM_Y_HI = 14,
  SART_ACC_SUM_R_HI = 15,
  SART_ACC_SUM_WEIGHTS_SQ_HI = 16,   /* FIXED64 raw accumulators only, like 12 .. 15 */
  /* FIXED64 raw accumulators only (two limbs; 0 in every f64 accumulator): sum of the weights of the passed rays that fall outside
   * the image - with it sart_finalize_accumulator_device checks that the pixels add up to SUM_WEIGHTS exactly (a wrapped slot cannot) */
  SART_ACC_SUM_WEIGHTS_OUTSIDE = 17,
  SART_ACC_SUM_WEIGHTS_OUTSIDE_HI = 18,
  /* 19 .. 23: reserved, 0 */
  SART_ACC_COUNT = 24
};

static inline size_t sart_accumulator_len(int32_t nx, int32_t ny) {
  return (size_t)nx * (size_t)ny + (size_t)SART_ACC_COUNT;
}
/*
 * With params.spectra != 0 the accumulator continues behind the scalars (all f64, passed rays only):
 *   radial_counts [n_radial_bins]     number of rays per bin of pointdataR (last bin collects r >= radial_max)
 *   radial_weights[n_radial_bins]     sum of weights per bin  ("Radial distribution", raytracer.nim:2378-2389)
 *   energy_counts [n_energies + 1]    rays per energy index (index n_energies = the X-ray test source's energy)
 *   energy_weights[n_energies + 1]    sum of weights per energy index   (flux after the experiment, :2595-2601)
 *   energy_reflect[n_energies + 1]    sum of `reflect` per energy index (reflectivity vs energy, :2295-2315)
 */
static inline size_t sart_accumulator_len_spectra(int32_t nx, int32_t ny, int32_t n_radial_bins, int32_t n_energies) {
  return sart_accumulator_len(nx, ny) + 2u * (size_t)n_radial_bins + 3u * ((size_t)n_energies + 1u);
}

/* Host-side view of the scalar tail. */
typedef struct sart_summary_t {
  double v[SART_ACC_COUNT];
} sart_summary_t;

typedef struct sart_context sart_context;

/* ---- lifecycle --------------------------------------------------------- */
int sart_abi_version(void);
const char* sart_last_error(void);
/* device_ordinal: HIP device index of this process. Fails with SART_ERR_NO_DEVICE without a GPU. */
int sart_create(int device_ordinal, sart_context** out);
int sart_destroy(sart_context* ctx);
/* Use an existing HIP stream (hipStream_t cast to void*) for all launches; NULL = the context's own stream. */
int sart_set_stream(sart_context* ctx, void* hip_stream);
/* Waits for the context's stream.  Also the place where problems found asynchronously surface: SART_ERR_ACCUMULATOR if a
 * FIXED64 finalize queued before it found unresolved weights or a wrapped slot ("accumulation mode" below; reported once). */
int sart_synchronize(sart_context* ctx);

/* ---- inputs (the captures of traceAxionWrapper, raytracer.nim:2223-2232) -- */
/* expSetup + detectorSetup + centerVecs. The library copies; caller keeps ownership.
 * SART_ERR_UNSUPPORTED: a telescope / reflectivity kind the reference asserts on (:1233, :1347).  SART_ERR_INVALID_ARGUMENT (the
 * message names the field): n_shells outside [9, 64], shell radii not ascending or glass thicker than the spacing, an enum out
 * of range, number_of_holes outside [0, 64], impossible magnet geometry, or any double of the struct (shell arrays: the first
 * n_shells entries) that is NaN or infinite - such a setup would trace to zero flux without a word. */
int sart_set_setup(sart_context* ctx, const sart_setup_t* setup);
int sart_get_setup(sart_context* ctx, sart_setup_t* out);
/* Cheap updates used by the scan drivers (performAngularScan :2796; m_a scan). */
/* A NaN leaves the corresponding angle unchanged. */
int sart_set_telescope_angles(sart_context* ctx, double turned_x_deg, double turned_y_deg);
int sart_set_axion_mass(sart_context* ctx, double m_axion_ev);
/* fluxRadiusCDF[nR], diffFluxCDFs[nR][nE] row-major, energies[nE] (FullRaytraceSetup :237-241). */
int sart_set_solar_tables(sart_context* ctx, const double* flux_radius_cdf,
                          const double* diff_flux_cdfs, const double* energies_kev,
                          int32_t n_radii, int32_t n_energies);
/*
 * The same tables built ON THE DEVICE from an emission-rate table that already lives there (e.g. the output of
 * sart_emission_table_device): the CDF construction of initFullSetup (raytracer.nim:2670-2705) - per radius row the
 * running sum of emRate * E^2 * r^2 over the energies in the reference's order, normalised by its last element; the running
 * sum of the row sums, normalised - and the guide tables in front of both, without a device -> host -> device round trip.
 * Bit-identical to sart_host_build_cdfs + sart_set_solar_tables on the same numbers.
 *   em_rates_device [n_radii][n_energies] row-major, DEVICE memory (not modified)
 *   radii [n_radii] (fractions of the solar radius, :2651), energies_kev [n_energies]: HOST memory
 * Blocking.  SART_ERR_INVALID_ARGUMENT if a row does not give a CDF (negative or non-finite rates, a row that sums to 0).
 */
int sart_set_solar_tables_device(sart_context* ctx, const double* em_rates_device, const double* radii,
                                 const double* energies_kev, int32_t n_radii, int32_t n_energies);
/*
 * Host copies of the sampling tables the context holds, whichever call set them (any pointer may be NULL):
 * flux_radius_cdf_out[n_radii], diff_flux_cdfs_out[n_radii][n_energies], and the library's guide tables
 * radius_guide_out[3074], energy_guide_out[n_radii][2594] (u16; layouts in csrc/sart_device.h: kRadiusGuideEntries,
 * kEnergyGuideEntries - for tests and debugging).
 */
int sart_get_solar_tables(sart_context* ctx, double* flux_radius_cdf_out, double* diff_flux_cdfs_out,
                          uint16_t* radius_guide_out, uint16_t* energy_guide_out);
/* Reflectivity grids as read by initReflectivity (:1160-1231): data[coating][angle][energy],
 * uniform grid defined by (min,max) of the axes (newBilinearSpline :1181/:1204/:1226). */
int sart_set_reflectivity(sart_context* ctx, int32_t n_coatings, int32_t n_angles,
                          int32_t n_energies, double angle_min_deg, double angle_max_deg,
                          double energy_min_kev, double energy_max_kev, const double* data);
/* The three newLinear1D tables of newDetectorSetup (:1522-1527): x in keV ascending. */
int sart_set_detector_tables(sart_context* ctx,
                             const double* strongback_x, const double* strongback_y, int32_t n_strongback,
                             const double* window_x, const double* window_y, int32_t n_window,
                             const double* gas_abs_x, const double* gas_abs_y, int32_t n_gas_abs);

/* ---- the hot path ------------------------------------------------------- */
/*
 * Literal drop-in for traceAxionWrapper (:2223-2244): traces params->n_rays rays and writes
 * one Axion record per ray to `ax_buf` (HOST memory, caller-allocated, n_rays * 208 bytes).
 * Every field of every record is written (the reference relies on zero-initialised seqs).
 * Above 2^20 records the rays are traced in chunks into two device buffers and copied out on a second stream; the
 * interior of `ax_buf` is advised MADV_HUGEPAGE and faulted in ahead of the copy by helper threads (every byte they touch
 * is overwritten with records; SART_NO_HOST_PREFAULT=1 leaves the buffer alone): 2.3e8 records/s into a fresh buffer,
 * 2.66e8 into mapped pages, PCIe bound 2.75e8.
 */
int sart_trace_records(sart_context* ctx, const sart_trace_params_t* params, sart_axion_t* ax_buf);
/* Same, but `ax_buf_device` is DEVICE memory and the call only enqueues on the stream. */
int sart_trace_records_device(sart_context* ctx, const sart_trace_params_t* params,
                              sart_axion_t* ax_buf_device);

/*
 * The passed rays only.  What the reference does with the buffer traceAxionWrapper filled (generateResultPlots :2252-2283, the
 * scan sum :2800): `axions.filterIt(it.passed)` for everything it plots, writes or sums, and the number of records with
 * passedTillWindow / hitNickel set, echoed.  This call returns exactly that: ax_buf[k] = the record of the k-th ray, in ray-id
 * order, whose `passed` is set - byte for byte the record sart_trace_records writes for that ray - and the four counts.
 * `capacity` = the records ax_buf has room for; as with snprintf, counts->n_passed is the number of passed rays whatever the
 * capacity, and min(n_passed, capacity) records are written (n_passed <= capacity: all of them).  Why: BabyIAXO passes 21 % of
 * its rays, so 79 % of the 208 bytes per ray that sart_trace_records moves across PCIe are never read; the rays are traced in
 * chunks of 2^20 into device scratch, compacted on the device (integer scan: the order does not depend on the launch
 * geometry) and only the passed records travel, overlapped with the next chunk.
 */
typedef struct sart_record_counts {
  uint64_t n_rays;                /* records looked at = params->n_rays                                   */
  uint64_t n_passed;              /* ... with passed (axionsPass.len, :2253)                               */
  uint64_t n_passed_till_window;  /* ... with passedTillWindow (:2255)                                     */
  uint64_t n_hit_nickel;          /* ... with hitNickel (:2257)                                            */
} sart_record_counts_t;
int sart_trace_records_passed(sart_context* ctx, const sart_trace_params_t* params, sart_axion_t* ax_buf, uint64_t capacity,
                              sart_record_counts_t* counts);
/* Same with DEVICE memory for the records and for the counts (four uint64_t in the order of sart_record_counts_t); enqueues on
 * the stream.  params->accumulate != 0: the records are appended behind the counts_device[1] records already there and the
 * counts grow (several launches into one buffer); 0: the counts start from zero. */
int sart_trace_records_passed_device(sart_context* ctx, const sart_trace_params_t* params, sart_axion_t* ax_buf_device,
                                     uint64_t capacity, uint64_t* counts_device);
/*
 * Device scratch of the record entries.  sart_trace_records and sart_trace_records_passed[_device] trace into buffers the
 * context owns: one of min(n_rays, 2^20) records (208 B each: 218 MB at full size), a second one as soon as a call needs more
 * than one chunk, and - host form of the passed-only call above one chunk - a third for the compacted records (436 - 654 MB in
 * all).  The buffers only grow (a call never frees: hipFree would synchronise the device) and stay until sart_destroy - or
 * until this call, which waits for the context's streams and frees them; the next record call allocates again.  The
 * histogram / scan entries do not use them.
 */
int sart_release_scratch(sart_context* ctx);

/*
 * Fused trace + accumulation: traceAxionWrapper + prepareHeatmap(256,256,...,norm=1) (:2629)
 * + flux sum (:2800) + the counters echoed at :2253-2257, without materialising records.
 * `accumulator_device` is DEVICE memory of sart_accumulator_len(nx,ny) doubles; the call is
 * asynchronous on the context's stream (pair with sart_synchronize) - except for the first launch after the geometry,
 * the tables or the image binning changed: it is preceded by a 2e5-ray pilot launch whose centroid places the LDS image
 * tile, read back with one stream synchronisation (~60 us).  That holds for every kernel variant (rotated
 * telescope and gas stage included): a host loop that changes the telescope's angle between image launches pays it per angle.
 * Launches without an image do not: flux-only launches (image_nx = image_ny = 0) and the fused scans (sart_trace_mass_scan,
 * sart_trace_angular_scan) place no tile and never synchronise.
 */
int sart_trace_histogram_device(sart_context* ctx, const sart_trace_params_t* params,
                                double* accumulator_device);
/* Blocking convenience form with HOST outputs (image may be NULL, summary may be NULL). */
int sart_trace_histogram(sart_context* ctx, const sart_trace_params_t* params,
                         double* image_out_host, sart_summary_t* summary_out);
/* Same, with params->spectra != 0: spectra_out_host receives the 2*n_radial_bins + 3*(n_energies+1) doubles behind the
 * scalars (layout above). */
int sart_trace_histogram_spectra(sart_context* ctx, const sart_trace_params_t* params, double* image_out_host,
                                 sart_summary_t* summary_out, double* spectra_out_host);

/* ---- accumulation mode -------------------------------------------------- */
/*
 * SART_ACCUM_F64 (default): the accumulator holds IEEE doubles and the kernel adds with f64 atomics, as prepareHeatmap
 * does on the CPU (raytracer.nim:838-842).  Counters are exact; sums depend on the order in which the hardware retires
 * the atomics (relative differences ~1e-13 between two runs, GPU counts, replica counts or launch splittings).
 *
 * SART_ACCUM_FIXED64: deterministic accumulation.  Every 8-byte slot of the accumulator is a two's-complement int64:
 * counters count, every sum is held in integer multiples of a power-of-two quantum fixed per context, and each ray adds
 * rint(value / quantum) with integer atomics - integer addition is associative, so the result is BITWISE independent of
 * the number of GPUs, of the replica / LDS-tile placement and of how the rays are split over launches.  A multi-GPU
 * reduce is an int64 sum of the raw buffers (ncclInt64 / torch.int64 view; sart_reduce_across_devices does this by
 * itself); sart_finalize_accumulator_device converts a raw buffer to the f64 layout documented above.
 *   quanta   weights (image pixels, SUM_WEIGHTS, radial / energy weight spectra): q_w = 2^e with
 *            w_bound < 2^(e + 63 - headroom_bits), w_bound = the host's scale of the largest weight of one ray for the setup,
 *            tables and flags of the launch that fixes the quantum (exposure x conversion probability x max reflectivity^2 x
 *            max window transmission x max gas absorption; the conversion probability over lengthB in the vacuum stage, and
 *            in the gas stage its bound for the context's axion mass, (g B / 2)^2 min(L^2, 4 / (q^2 + Gamma^2 / 4)) maximised
 *            over the energy table - it follows the mass, so a far-off-resonance point resolves as well as the resonance).
 *            headroom_bits (default 27) = log2 of the number of w_bound-weight rays a slot can take before it wraps: a pixel
 *            holds 2^27 = 1.3e8 of them (measured on the 256 x 256 BabyIAXO / XMM image: the brightest pixel passes 2^62 - where
 *            the status check starts to fail - after 1.1e12 traced rays and wraps after 2.1e12; with headroom_bits = 31 it
 *            stands at 2^59.3 after 2.6e12 rays: profiles/r04_v48_fixed64_long_run.txt); the resolution of one
 *            ray's weight is 2^-36 w_bound, and a pixel that n rays hit carries a rounding error of ~q_w sqrt(n / 12): an
 *            image of 2e7 rays agrees with the f64 image to < 1e-12 of its largest pixel, larger images better.
 *            SUM_WEIGHTS_SQ: q = 2^(2 e' - 39) with w_bound < 2^e' (two limbs; the quantum leaves room for the rays of one
 *            workgroup of one launch).  SUM_X / SUM_Y / SUM_R: 2^-32 mm.  energy_reflect spectrum: 2^-40.
 *            The bound sees table maxima (for the X-ray test source: the values at its one energy), not what the rays
 *            actually meet: every ray's rounding error is <= q_w / 2 in absolute terms whatever its weight.
 *   limbs    SUM_WEIGHTS, SUM_X, SUM_Y, SUM_R, SUM_WEIGHTS_SQ (and SUM_WEIGHTS_OUTSIDE) receive every passed ray of every
 *            launch: value = (hi * 2^40 + lo) * q with lo in slot SART_ACC_SUM_*, hi in slot SART_ACC_SUM_*_HI (after a launch
 *            0 <= lo < 2^40; limb-wise int64 sums over up to 2^22 ranks stay exact).
 *   freezing the quanta are computed by the first histogram launch after a change of the accumulation mode (or by any launch
 *            with accumulate == 0) and kept for the following accumulate == 1 launches of the context, so that launches that
 *            add into one accumulator share them; a launch whose weight bound no longer fits fails with
 *            SART_ERR_INVALID_ARGUMENT.  In the gas stage the bound follows the axion mass: start the accumulator of every
 *            scan point with an accumulate == 0 launch (an accumulate == 1 launch after sart_set_axion_mass keeps the frozen
 *            quanta: refused if the new bound does not fit them, and reported by the checks below if the new weights fall
 *            under their resolution).  Contexts with equal inputs compute equal quanta (ranks of a multi-GPU job).
 *   checks   integers can stop meaning what they should in two ways, and neither passes silently.  The finalize kernels
 *            (sart_finalize_accumulator_device, sart_finalize_mass_scan_device, and the blocking host-output calls, which
 *            finalize internally) examine the raw accumulator and record what they find in a status word of the context;
 *            the next sart_synchronize - and every blocking host-output call - returns SART_ERR_ACCUMULATOR for it:
 *              unresolved  the accumulated weights average below 2^12 quanta per passed ray (an outlier in a table inflated
 *                          the bound): use a smaller headroom.  Judged once 256 rays have passed (the average of a handful
 *                          of faint rays says nothing about the bound; each of them is still exact to half a quantum).  Squared weights that average below 2^6 quanta are not an
 *                          error - SUM_WEIGHTS_SQ (an error estimate; nothing else depends on it) then reads NaN.
 *              wrapped     a slot is negative or >= 2^62, or the pixels (plus SUM_WEIGHTS_OUTSIDE), the radial weight bins or
 *                          the energy weight bins do not add up to SUM_WEIGHTS - every passed ray adds the same integer to
 *                          each of them, so the sums agree exactly unless a slot wrapped, however often (each wrap takes 2^64
 *                          out of its sum): use a larger headroom, or finalize and start a new accumulator earlier.
 * The mode applies to sart_trace_histogram_device (raw accumulators) and to the blocking host-output calls, which
 * finalize internally and keep returning doubles.
 */
enum { SART_ACCUM_F64 = 0, SART_ACCUM_FIXED64 = 1 };
typedef struct sart_fixed_quanta_t {
  double weight;        /* q_w: pixels, SUM_WEIGHTS (both limbs), radial_weights, energy_weights */
  double weight_sq;     /* SUM_WEIGHTS_SQ */
  double position;      /* SUM_X, SUM_Y, SUM_R (2^-32 mm) */
  double reflect;       /* energy_reflect (2^-40) */
} sart_fixed_quanta_t;
/* headroom_bits: 0 = default (27); otherwise 16 .. 44.  Waits for the stream; a CHANGE of mode or headroom releases the frozen
 * quanta (setting what is already set changes nothing: an accumulator may hold data in them). */
int sart_set_accumulation_mode(sart_context* ctx, int mode, int headroom_bits);
int sart_get_accumulation_mode(sart_context* ctx, int* mode_out);
/* The frozen quanta (SART_ERR_NOT_READY before the first FIXED64 launch). */
int sart_get_fixed_quanta(sart_context* ctx, sart_fixed_quanta_t* out);
/*
 * Raw FIXED64 accumulator -> the f64 accumulator layout (image, scalars [the *_HI slots read 0], spectra if
 * params->spectra), with the context's frozen quanta.  Both pointers are DEVICE memory of the accumulator's length;
 * out_f64_device may equal acc_fixed_device (in place).  Asynchronous on the context's stream; what the conversion finds
 * wrong with the integers ("checks" above) is returned by the next sart_synchronize.
 */
int sart_finalize_accumulator_device(sart_context* ctx, const sart_trace_params_t* params, const void* acc_fixed_device,
                                     double* out_f64_device);

/*
 * Roll-over for long accumulations.  An int64 slot ends where 2^headroom_bits bound-weight rays have met on one pixel or bin
 * (default 27: BabyIAXO's brightest pixel passes 2^62 after 1.1e12 rays).  Instead of hand-picking a larger headroom - and a
 * coarser quantum - a caller can give every slot a second limb: `hi_limbs_device` is DEVICE memory of the accumulator's length
 * (int64, zeroed by the caller when the accumulator is); this call moves the bits of every slot of acc_fixed_device above 2^40
 * into the slot's limb (slot i then stands for (hi[i] 2^40 + acc[i]) quanta, 0 <= acc[i] < 2^40) - one pass over the slots,
 * ~5 us for a 256 x 256 image.  Called between launches - after every one, or every few hundred 1e9-ray launches - an
 * accumulation runs as long as anybody likes at the fine quantum; a slot that wrapped between two calls is still caught (this
 * call checks sign and size of every slot it folds; the finalize's conservation sums include the limbs).  Both arrays reduce
 * over ranks as int64 sums.  Asynchronous on the context's stream; problems surface as SART_ERR_ACCUMULATOR from the next
 * sart_synchronize.
 */
int sart_rollover_accumulator_device(sart_context* ctx, const sart_trace_params_t* params, void* acc_fixed_device,
                                     void* hi_limbs_device);
/* sart_finalize_accumulator_device for an accumulator with roll-over limbs (hi_limbs_device may be NULL: no limbs).
 * out_f64_device may equal acc_fixed_device, not hi_limbs_device. */
int sart_finalize_accumulator_limbs_device(sart_context* ctx, const sart_trace_params_t* params, const void* acc_fixed_device,
                                           const void* hi_limbs_device, double* out_f64_device);

/* ---- fused axion-mass scan (BASELINE configs[4]) -------------------------- */
/*
 * The reference has ONE constant axion mass (`mAxion`, raytracer.nim:255); a mass scan on it is a host loop that re-runs
 * calculateFluxFractions per mass.  In the gas stage the mass enters a ray's weight through the conversion probability
 * alone (computeMagnetTransmission :1599-1625 -> axionConversionProb2, axionMassforMagnet.nim:75-98): sampling, cuts,
 * mirrors, reflectivities, window and absorption do not depend on it.  These entry points trace every ray ONCE and
 * evaluate the conversion probability for n_masses masses in the last stage of the kernel (per mass and surviving ray:
 * q = |m_gamma^2 - m_a^2| / 2E, one reciprocal, one cosine; Gamma(E), exp(-Gamma L / 2) and the rest of the weight are
 * shared), accumulating per mass the flux, its sum of squares and the number of passed rays - the same ray ids for every
 * mass (common random numbers: the curve's point-to-point noise is the noise of the weights' ratio, not of two samples).
 * Per mass the results equal a sart_trace_histogram launch with sart_set_axion_mass(m) on the same ray ids: bit for bit
 * in SART_ACCUM_FIXED64, up to the summation order in SART_ACCUM_F64.
 *
 * Scan accumulator: (n_masses + 1) rows of SART_SCAN_ROW 8-byte slots (f64, or int64 when raw SART_ACCUM_FIXED64):
 *   row k < n_masses   SART_SCAN_SUM_WEIGHTS, SART_SCAN_SUM_WEIGHTS_SQ, SART_SCAN_N_PASSED of masses_ev[k]
 *                      (raw FIXED64: + the high limbs SART_SCAN_SUM_WEIGHTS_HI / _SQ_HI, value = (hi 2^40 + lo) quantum)
 *   row n_masses       mass-independent counters SART_SCAN_N_* below
 * Only params->n_rays, seed, ray_id_offset, flags and accumulate are read (no image is accumulated).
 * SART_ERR_INVALID_ARGUMENT unless the setup's stage is SART_SK_GAS (the vacuum probability :363-365 has no m_a in it).
 * Masses are processed in groups of 32 per kernel launch (the per-mass accumulators of a workgroup live in LDS), and every
 * group traces the rays again: a scan of 40 masses costs two traces, one of 64 masses two as well.
 * FIXED64: the quanta of mass k are a function of (setup, tables, flags, headroom, masses_ev[k]) alone - every rank of a
 * multi-GPU job computes the same ones; a reduce is an int64 sum of the raw scan accumulators;
 * sart_finalize_mass_scan_device (same masses) converts to doubles.
 */
enum { SART_SCAN_SUM_WEIGHTS = 0, SART_SCAN_SUM_WEIGHTS_SQ = 1, SART_SCAN_N_PASSED = 2,
       SART_SCAN_SUM_WEIGHTS_HI = 4, SART_SCAN_SUM_WEIGHTS_SQ_HI = 5, SART_SCAN_ROW = 8 };
enum { SART_SCAN_N_RAYS = 0, SART_SCAN_N_REACHED_TELESCOPE = 1, SART_SCAN_N_SHELL_SELECTED = 2, SART_SCAN_N_HIT_NICKEL = 3,
       SART_SCAN_N_ON_DETECTOR = 4 /* rays on the chip whose mass-independent weight factor is non-zero */ };
static inline size_t sart_mass_scan_len(int32_t n_masses) { return ((size_t)n_masses + 1u) * (size_t)SART_SCAN_ROW; }
/* scan_acc_device: DEVICE memory of sart_mass_scan_len(n_masses) 8-byte slots; asynchronous on the context's stream. */
int sart_trace_mass_scan_device(sart_context* ctx, const sart_trace_params_t* params, const double* masses_ev,
                                int32_t n_masses, double* scan_acc_device);
/* Blocking form with a HOST output of sart_mass_scan_len(n_masses) doubles (finalized in FIXED64 mode). */
int sart_trace_mass_scan(sart_context* ctx, const sart_trace_params_t* params, const double* masses_ev, int32_t n_masses,
                         double* scan_out_host);
/* Raw FIXED64 scan accumulator -> doubles (device pointers; in place allowed); asynchronous.  Resolution / overflow
 * problems are reported by the next sart_synchronize (see "accumulation mode"). */
int sart_finalize_mass_scan_device(sart_context* ctx, const sart_trace_params_t* params, const double* masses_ev,
                                   int32_t n_masses, const void* scan_fixed_device, double* out_f64_device);

/* ---- fused angular scan (BASELINE configs[3]) ----------------------------- */
/*
 * performAngularScan (raytracer.nim:2778-2802) re-runs calculateFluxFractions per telescope angle: it copies the setup, sets
 * telescope_turned_y (:2796) and traces NumberOfPointsSun fresh rays.  The angle enters a ray at the transformation into the
 * telescope's frame (:1878-1899) and nowhere before it: sampling from the solar model, the point on the bore exit, bore,
 * cold-bore exit and the pipe cuts (:1746-1868) do not depend on it.  These entry points take every ray through that part ONCE
 * and run telescope frame -> opaque structures -> shell selection -> mirrors -> weight once per angle, accumulating per angle
 * the flux, its sum of squares and four counters - the same ray ids for every angle (common random numbers: the curve's
 * point-to-point noise is the noise of the weights' differences, not of two samples).  Per angle the results equal a
 * sart_trace_histogram launch after sart_set_telescope_angles(ctx, NaN, turned_y_deg[k]) on the same ray ids: bit for bit in
 * SART_ACCUM_FIXED64 (for an angle that leaves the telescope unrotated - turned x and y both 0 - the single launch runs the
 * unrotated kernel, whose frame change is exact where the rotation by 0 rounds: equal to ~1e-13 then), up to the summation
 * order in SART_ACCUM_F64.  turned_x stays what the context's setup says; the setup itself is not changed.
 *
 * Scan accumulator: (n_angles + 1) rows of SART_ASCAN_ROW 8-byte slots (f64, or int64 when raw SART_ACCUM_FIXED64):
 *   row k < n_angles   SART_ASCAN_SUM_WEIGHTS, _SUM_WEIGHTS_SQ, _N_PASSED, _N_SHELL_SELECTED, _N_HIT_NICKEL,
 *                      _N_PASSED_TILL_WINDOW of turned_y_deg[k] (raw FIXED64: + the high limbs _SUM_WEIGHTS_HI / _SQ_HI,
 *                      value = (hi 2^40 + lo) quantum)
 *   row n_angles       angle-independent counters SART_ASCAN_N_RAYS, SART_ASCAN_N_REACHED_TELESCOPE
 * Only params->n_rays, seed, ray_id_offset, flags and accumulate are read (no image is accumulated).
 * Angles are processed in groups of up to 32 per kernel launch (balanced: 50 angles = 25 + 25); every group traces
 * the rays again.  FIXED64: the quanta are a function of (setup, tables, flags, headroom) alone, the same for every angle and
 * every rank of a multi-GPU job; a reduce is an int64 sum of the raw scan accumulators; sart_finalize_angular_scan_device
 * converts to doubles.  The common quantum comes from the on-axis weight bound: an angle far off axis whose few passed rays carry
 * weights it does not resolve (on average below 2^12 quanta per passed ray) reads NaN in ITS row's SUM_WEIGHTS (and
 * SUM_WEIGHTS_SQ); the other rows and the call's status are not affected - use a smaller headroom or SART_ACCUM_F64 for such
 * a scan.
 * Angles: every turned_y_deg[k] must be finite and inside (-90, 90) degrees (SART_ERR_INVALID_ARGUMENT otherwise; the host loop
 * sart_set_telescope_angles + sart_trace_histogram takes any finite angle).  Stage A0 retires, before anything is sampled, the
 * rays whose bore-exit radius alone proves them dead at EVERY angle of a launch group (pipes; inner disc, ring and the annulus
 * beyond the last shell narrowed by the group's largest tilt) - a wide scan is best given in ascending |angle|, so that the
 * groups of 32 have small ranges.
 */
enum { SART_ASCAN_SUM_WEIGHTS = 0, SART_ASCAN_SUM_WEIGHTS_SQ = 1, SART_ASCAN_N_PASSED = 2, SART_ASCAN_N_SHELL_SELECTED = 3,
       SART_ASCAN_SUM_WEIGHTS_HI = 4, SART_ASCAN_SUM_WEIGHTS_SQ_HI = 5, SART_ASCAN_N_HIT_NICKEL = 6,
       SART_ASCAN_N_PASSED_TILL_WINDOW = 7, SART_ASCAN_ROW = 8 };
enum { SART_ASCAN_N_RAYS = 0, SART_ASCAN_N_REACHED_TELESCOPE = 1 };
static inline size_t sart_angular_scan_len(int32_t n_angles) { return ((size_t)n_angles + 1u) * (size_t)SART_ASCAN_ROW; }
/* scan_acc_device: DEVICE memory of sart_angular_scan_len(n_angles) 8-byte slots; asynchronous on the context's stream. */
int sart_trace_angular_scan_device(sart_context* ctx, const sart_trace_params_t* params, const double* turned_y_deg,
                                   int32_t n_angles, double* scan_acc_device);
/* Blocking form with a HOST output of sart_angular_scan_len(n_angles) doubles (finalized in FIXED64 mode). */
int sart_trace_angular_scan(sart_context* ctx, const sart_trace_params_t* params, const double* turned_y_deg, int32_t n_angles,
                            double* scan_out_host);
/* Raw FIXED64 scan accumulator -> doubles (device pointers; in place allowed); asynchronous.  Resolution / overflow
 * problems are reported by the next sart_synchronize (see "accumulation mode"). */
int sart_finalize_angular_scan_device(sart_context* ctx, const sart_trace_params_t* params, int32_t n_angles,
                                      const void* scan_fixed_device, double* out_f64_device);

/* ---- multi-GPU ---------------------------------------------------------- */
/*
 * Sum the fused accumulators of n contexts (one per GPU of this process) into the one of contexts[root]:
 * a single RCCL ncclReduce(ncclSum) of n_doubles 8-byte elements over xGMI (grouped over the devices): f64 elements, or
 * int64 when the contexts are in SART_ACCUM_FIXED64 mode (all of them must be in the same mode).
 * accumulators_device[i] must live on the device of contexts[i].  Blocking.  n == 1 waits for the context's stream and returns
 * (one accumulator: nothing to add up, no RCCL call).  librccl is loaded at the first call with n > 1 (dlopen: the library has
 * no link-time dependency on it); one communicator set per ordered device list is created on first use (ncclCommInitAll) and
 * kept for the life of the process; a collective that fails drops its communicators (SART_ERR_INTERNAL with RCCL's message), the
 * next call builds new ones.  In SART_ACCUM_FIXED64 the two-limb sums (SUM_WEIGHTS, ...: hi 2^40 + lo) add limb by limb without
 * carrying - the same integers in another representation; sart_finalize_accumulator_device reads either.
 * A host that runs one process per GPU (e.g. under MPI / torchrun) reduces with its own communicator instead
 * (bench.py: torch.distributed, backend "nccl" = RCCL).  Returns SART_ERR_UNSUPPORTED if librccl cannot be loaded.
 */
int sart_reduce_across_devices(sart_context* const* contexts, double* const* accumulators_device, int32_t n,
                               size_t n_doubles, int32_t root);

/* ---- measurement -------------------------------------------------------- */
/*
 * HIP-event timing of the hot-path kernels on the stream they are launched on.
 * enable != 0: every subsequent sart_trace_*_device launch is bracketed by events.
 * sart_get_kernel_timing synchronises the stream, returns the sum of kernel durations and
 * the number of launches since the last reset, and resets.
 */
int sart_enable_kernel_timing(sart_context* ctx, int enable);
int sart_get_kernel_timing(sart_context* ctx, double* total_ms, int64_t* n_launches);

/*
 * Identity of this build of the library: hash of the device sources, their headers and the compile flags
 * (csrc/Makefile: BUILD_ID).  Hardware-counter profiles are collected in separate runs (rocprofv3 --pmc cannot run
 * inside the timed process); tools/pmc_summary.py stores this id beside the figures it publishes and bench.py
 * reports counter-derived numbers only when the id of the library it times is the same.  Static string.
 */
const char* sart_build_id(void);

/* Device properties the bench reports (CU count etc.). Any pointer may be NULL. */
int sart_device_info(sart_context* ctx, int32_t* n_cu, int32_t* wave_size,
                     char* name_buf, size_t name_buf_len);

#ifdef __cplusplus
}
#endif
#endif /* SART_H */
