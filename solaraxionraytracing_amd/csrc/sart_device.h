// sart_device.h — PODs shared by the host side of libsart (sart_api.hip) and the gfx950 kernels
// (sart_kernels.hip).  Everything here is *derived* from sart_setup_t + the tables by the hoist_*
// functions of sart_api.hip: per-setup / per-shell / per-energy-index constants are evaluated once on
// the host with the reference's formulas (cited there) so that the per-ray kernel contains no
// setup-only transcendental.
//
// Where the data lives on the device:
//   DevParams        by value in the kernel arguments (scalar registers; loop-invariant)
//   ShellDev[], shell LUT, fluxRadiusCDF + its guide table
//                    HBM -> staged into LDS once per workgroup (coalesced loads)
//   diffFluxCDFs (23.6 MB), its guide table, EnergyDev[], reflectivity (12 MB / coating)
//                    HBM, served by L2 / Infinity Cache (random, dependent gathers)
#pragma once
#include <stdint.h>

namespace sart {

constexpr int kMaxShells = 64;
constexpr int kMaxStrips = 16;       // half the number of window strips that are looped over
constexpr int kRadiusGuide = 2048;   // buckets of the guide table in front of fluxRadiusCDF
// The CDF's entries crowd towards 1 (the outer radii emit next to nothing): uniform buckets of 1/2048 hold up to hundreds of them
// there, and a wave with ONE such lane walks a binary search.  u >= 31/32 therefore has a second guide of 1024 buckets of 1/32768
// behind the first: entries 0 .. 2048 = lowerBound(cdf, k / 2048), entries 2049 .. 3073 = lowerBound(cdf, 31/32 + j / 32768).
// With K = floor(u 2^32) (Uniforms::u2_hi): bucket K >> 21, or 2049 + ((K - kRadiusGuideTopStart) >> 17) for K >= kRadiusGuideTopStart.
constexpr int kRadiusGuideTop = 1024;
constexpr uint32_t kRadiusGuideTopStart = 0xF8000000u;   // 31/32 of 2^32
constexpr int kRadiusGuideEntries = kRadiusGuide + 1 + kRadiusGuideTop + 1;   // 3074
// Guide table in front of every row of diffFluxCDFs (lowerBound of the energy draw, raytracer.nim:464-468).  Buckets in the
// uniform u of the draw: width 1/kEnergyGuideDiv below u = 31/32; above, where a solar spectrum's CDF creeps towards 1 over hundreds
// of energies, buckets of constant RELATIVE width in v = 1 - u (64 per octave of v, read off the bits of the double v) down
// to v = 2^-30.  A bucket then holds 0-2 table entries almost everywhere, and the draw resolves with ONE gather of four
// consecutive CDF values instead of a data-dependent binary search (a chain of dependent HBM / Infinity-Cache round trips).
// 1024 rather than 2048 uniform buckets: the guide shrinks from 14.1 to 10.2 MB (its hot part from 2 to 1 MB per 500 radius
// rows) while a bucket still holds fewer than four entries wherever the CDF rises; measured -1.3 % on CAST / LLNL, -1 % on
// BabyIAXO, 512 buckets: no further gain (profiles/r03_exp_guide_density.txt).  Results do not depend on the density.
#ifndef SART_ENERGY_GUIDE_DIV
#define SART_ENERGY_GUIDE_DIV 1024
#endif
constexpr int kEnergyGuideDiv = SART_ENERGY_GUIDE_DIV;    // uniform buckets per unit of u (a power of two)
constexpr int kEnergyGuideUniform = kEnergyGuideDiv / 32 * 31;   // buckets [k / Div, (k + 1) / Div), k < Div * 31/32
constexpr int kEnergyGuideLogMax = 25 * 64;               // log buckets j = 1 .. 1600 (j = 0: the single point u = 31/32)
constexpr int kEnergyGuideBuckets = kEnergyGuideUniform + kEnergyGuideLogMax + 1;
constexpr int kEnergyGuideEntries = kEnergyGuideBuckets + 1;   // u16 per row: bucket k is bracketed by entries k, k + 1
constexpr uint32_t kEnergyGuideCode0 = (1023u - 5u) << 6;  // (high word of the double 1/32) >> 14
constexpr int kEnergyCdfPad = 4;                          // 1.0-valued pad behind every CDF row (four-wide candidate gather)
constexpr int kShellLutMax = 1024;   // cells of the radial look-up table of the shell selection
constexpr int kMaxRadii = 2048;      // fluxRadiusCDF entries that fit the LDS stage
constexpr int kSinCosEntries = 129;  // (cos, sin)(pi k / 64), k = 0 .. 128: table behind the sampling angles

// Per-shell constants (Wolter-I pair j).  Quadratics are expressed in the telescope frame with
// the ray parametrised by z:  X(z) = X0 + sx z,  Y(z) = Y0 + sy z, so that with
//   A = sx^2 + sy^2,  D = X0 sx + Y0 sy,  Q = X0^2 + Y0^2
// a = A - k,  half_b = D + hb,  c = Q - cc   (findPosCone/Parabolic/Hyperbolic, raytracer.nim:628-729).
struct ShellDev {
  double r1;        // allR1[j]
  double r1_outer;  // allR1[j] + allThickness[j]
  // mirror 1 (cone, angle beta | paraboloid)
  double m1_k, m1_hb, m1_cc, m1_zhi;   // z range (0, m1_zhi)
  // mirror 2 (cone r4, angle 3 beta | hyperboloid)
  double m2_k, m2_hb, m2_cc, m2_zlo, m2_zhi;
  double m2_rc;     // cone 2: radius at z = m2_zlo (r4)
  // surface normals (calcNormalVec, raytracer.nim:731-759)
  double n1_tan;    // cone: tan(beta)
  double n1_r3t;    // paraboloid: r3 * tan(beta);    n1_r3sq = r3^2, n1_e = 2 r3 tan(beta)
  double n1_r3sq, n1_e;
  double n2_tan;    // cone: tan(3 beta)
  double n2_r3t;    // hyperboloid: r3 * tan(3 beta); n2_r3sq = r3^2, n2_e = 2 r3 tan(3 beta)
  double n2_r3sq, n2_e, n2_invF;       // invF = 1 / (f + r3 cot(2 beta))
  // detector distance for this shell (raytracer.nim:2070-2072), already divided by cos(pipe angle)
  double dist_det, dist_det_end;
  double dist_det_raw;                 // distDet before the 1/cos (xray-test straight-through, :2131)
  // lineHitsNickel (raytracer.nim:1722): r1 - (R1[j-1] + t[j-1]); unused for j == 0
  double nickel_num;
  // reflectivity grid of this shell: layers.lowerBound(j) (raytracer.nim:1573)
  int32_t coating;
  int32_t refl_row0;   // coating * (n_energies + 1): first row of this shell's grid in refl[][n_angles]
};
static_assert(sizeof(ShellDev) % 8 == 0, "ShellDev is staged into LDS as 8-byte words");

// Per-energy-index constants: the energy of a ray is one of the n_energies table values
// (raytracer.nim:470-471: energies[idx], clamped to >= 0.03 keV), so every E-only factor is a table.
// Row n_energies holds the same quantities for the X-ray test source's fixed energy (:1771).
struct EnergyDev {
  double energy;       // max(0.03, energies[idx])
  double t_window;     // windowTransmission.eval(E)      (:2179)
  double t_strongback; // strongbackTransmission.eval(E)  (:2170)
  double a_gas;        // gasAbsorption.eval(E)           (:2190)
  // gas stage (axionMassforMagnet.nim:75-113)
  double gamma;        // Gamma(E) in eV
  double inv_two_e_ev; // 1 / (2 * E[eV])  (momentumTransfer, :68: q = |m_gamma^2 - m_a^2| / (2 E), as a multiplication)
  double mu_pipe;      // massAtt * rhoPipe   * 100   [1/m]
  double mu_magnet;    // massAtt * rhoMagnet * 100   [1/m]
};

// Loop-invariant scalars, passed by value as a kernel argument.
struct DevParams {
  // ---- sampling (raytracer.nim:412-471) ----
  double sun_distance, sun_radius;
  double radius_cb, radius_cb_sq, length_b, length_coldbore;
  double pipe1_len, pipe2_len, pipe1_radius_sq;
  int32_t n_radii, n_energies;
  int32_t radius_span;        // max entries of fluxRadiusCDF between two guide marks a draw can meet (informational: the kernel reads four
                              // candidates and searches on only in a wider bracket)
  int32_t _pad0;
  // ---- X-ray test source (raytracer.nim:1765-1806) ----
  int32_t test_active, test_parallel;
  double test_x, test_y, test_z, test_radius, test_radius_sq, test_collimator_z;
  // ---- telescope frame (raytracer.nim:1878-1899) ----
  int32_t telescope_wolter;   // 1: paraboloid + hyperboloid (XMM, Abrixas); 0: cones (LLNL)
  int32_t telescope_kind;     // SART_TK_*
  int32_t n_shells, rotated;
  double l_mirror;
  double rx_c, rx_s, ry_c, ry_s, half_length_telescope;
  double entrance_x, entrance_y;
  // ---- shell selection (raytracer.nim:1932-1957) ----
  double r1_last;             // allR1[^1]
  double lut_inv_step;
  int32_t lut_n, _pad1;
  // A ray that selects the innermost shell (none below it, so no nickel test, :1719) from a radial distance below this value
  // provably stays inside that shell's first mirror over its whole length: neither root of findPosParabolic lies in the
  // mirror (:677-682), the input point comes back and the ray ends at the no-hit test (:2055) with every counter untouched.
  // Phase A counts it as "shell selected" and does not hand it to phase B.  -1: not proved for this setup (sart_api.hip:
  // shell0_miss_radius_of).
  double shell0_miss_radius;
  // ---- opaque structures (raytracer.nim:1635-1704) ----
  double spider_z;            // -85 (XMM) / -35 (Abrixas)
  double inner_radius;        // XMM: 64.7 (<=) ; Abrixas: 37.5 (<)
  double ring_lo, ring_hi;    // XMM: 130.7 .. 151.6 (exclusive)
  // spokes every 360/spoke_n degrees, half width w:  blocked <=> cos(spoke_n phi) >= cos(spoke_n w)
  int32_t spoke_n;            // 16 (XMM: 22.5 deg) | 6 (Abrixas: 60 deg)
  int32_t inner_blocks;       // XMM: 1 if the hole loop always blocks (htNone), -1: evaluate it
  double spoke_cos_thr;
  int32_t hole_type, number_of_holes;
  double hole_in_optics;
  // ---- detector plane (raytracer.nim:797-814, 2133-2204) ----
  double pipe_c, pipe_s, d_cb_xray;
  double lateral_shift, transversal_shift;
  double radius_window_sq, chip_cx, chip_cy;
  double theta_c, theta_s;
  int32_t n_half_strips, stage_gas;
  double strip_lo[kMaxStrips], strip_hi[kMaxStrips];
  // ---- weights (raytracer.nim:363-365, 1582-1625, 2207-2212) ----
  double conv_k;              // P(a->gamma) = conv_k * pathCB^2 (vacuum)
  double exposure;            // 3.585e3*3600*1.5*90 | 9.5e6*3600*12*90
  double gas_m_gamma_sq, gas_term1, gas_inv_hbarc_m;   // m_gamma^2, (g B / 2)^2, 1e-3 / 1.97e-7 (mm -> 1/eV)
  // |m_gamma^2 - m_a^2| (one IEEE subtraction of the two squares; the mass scan's table holds the same expression per mass).
  // (Takes the place of m_a^2 in this block: the LDS copy of the blob keeps its size, and with it the rings their addresses -
  // 16 bytes more and every ring access needs an extra address add, +2 vector instructions per loop iteration.)
  double gas_dm2_abs;
  // ---- reflectivity (raytracer.nim:1533-1580) ----
  int32_t refl_n_angles, n_coatings;
  double refl_angle_min, refl_inv_dangle, refl_dangle;
};
static_assert(sizeof(DevParams) < 2048, "DevParams travels in the kernel-argument segment");

// The few scalars phase A touches for every ray: by value in the kernel arguments (SGPRs).
struct HotA {
  double sun_distance, sun_radius;
  double radius_cb, radius_cb_sq, length_b, length_coldbore;
  double dz1, dz2, dz3;        // z of cold-bore exit / pipe exits relative to the magnetic-field exit plane
  double pipe1_radius_sq;
  double entrance_x, entrance_y;
  double r1_last, lut_inv_step;
  double spider_z, spoke_cos_thr, inner_radius, ring_lo, ring_hi;
  int32_t test_active, rotated, telescope_kind, spoke_n;
  int32_t n_shells, lut_n, radius_span, inner_blocks;
  // Stage A0 (early rejection on the bore-exit radius alone): the hi word w of the uniform u3 that sets the
  // radius of the point on the bore exit (r = R sqrt(u3), raytracer.nim:418) lies in zone z if
  // zone_lo[z] <= w <= zone_hi[z]; rays in a zone provably die before the mirrors.  Bit z of zone_reached:
  // they provably pass bore + pipes first (they still count as "reached the telescope").  0 zones: stage off.
  int32_t n_zones;
  uint32_t zone_reached;
  uint32_t zone_lo[4], zone_hi[4];
};
constexpr int kMaxZones = 4;

// What phase B needs as *scalars* at its gathers: the bases of the HBM-resident tables and the sizes their offsets are
// built from.  Passed by value as a kernel argument and re-read with scalar loads at the start of a phase-B pass, so
// that every gather is `global_load ... v_offset32, s[base]` instead of 64-bit vector address arithmetic on pointers
// fetched from LDS.  All tables are < 4 GB (checked on the host), offsets are 32-bit.
struct HotB {
  const double* diff_flux_cdfs;       // [n_radii][cdf_stride]: every row followed by kEnergyCdfPad entries of 1.0
  // The same table as the upper 32 of the 52 bits of floor(cdf * 2^52), same stride: what the four-candidate count of the energy
  // draw gathers (16 bytes instead of 32; half the cache footprint of the table that competes hardest for L2).  The draw's
  // uniform is k * 2^-52 with an integer k, so  cdf[i] < u  <=>  floor(cdf[i] 2^52) < k, and the upper 32 bits decide that unless
  // they are equal (one draw in ~1e9): then, and in buckets wider than four entries, the f64 row decides.
  const uint32_t* cdf_hi32;
  const uint16_t* energy_guide;       // [n_radii][kEnergyGuideEntries]
  const EnergyDev* energy_tab;        // [n_energies + 1]
  const double* refl;                 // [n_coatings][n_energies + 1][n_angles]
  int32_t n_energies, refl_n_angles;
  int32_t cdf_stride, _pad;           // n_energies + kEnergyCdfPad
};

// Device pointers of one context.
struct DevTables {
  const double* sincos_tab;           // [kSinCosEntries][2]: (cos, sin)(pi k / 64), correctly rounded
  const ShellDev* shells;             // [n_shells]
  const uint8_t* shell_lut;           // [lut_n]: first shell with R1 > k * lut_step
  const double* flux_radius_cdf;      // [n_radii]
  const uint16_t* radius_guide;       // [kRadiusGuide + 1]
  const double* diff_flux_cdfs;       // [n_radii][n_energies + kEnergyCdfPad] (rows padded with 1.0)
  const uint16_t* energy_guide;       // [n_radii][kEnergyGuideEntries]
  const EnergyDev* energy_tab;        // [n_energies + 1]
  // reflectivity re-tabulated per energy index: refl[coating][e_idx][angle] (see hoist_reflectivity)
  const double* refl;                 // [n_coatings][n_energies + 1][n_angles]
};

// DevParams + DevTables as one blob in HBM; staged into LDS by every workgroup.
struct DevBlob {
  DevParams P;
  DevTables T;
};
static_assert(sizeof(DevBlob) % 8 == 0, "DevBlob is staged into LDS as 8-byte words");

constexpr int kMaxImageReplicas = 128;   // power of two

struct TraceArgs {
  uint64_t n_rays, ray_id_offset;
  // Image atomics go to replica (wave id & replica_mask) of a scratch image: contended f64 atomics on a small
  // focal spot serialise at the memory side.  fold_replicas_kernel adds the replicas into the caller's
  // accumulator.  replica_mask = 0 (wide images): `replicas` is the accumulator itself, no fold.
  double* replicas;
  uint32_t replica_mask;
  uint32_t replica_stride;   // doubles between two replicas (image size + padding, see sart_api.hip)
  // Per-workgroup partial sums of the SART_ACC_COUNT scalars (plain stores; fold_scalars_kernel adds them to the
  // accumulator): thousands of f64 atomics on the same 11 addresses serialise for hundreds of microseconds.
  double* partials;
  uint32_t seed_lo, seed_hi;
  uint32_t flags;
  int32_t image_nx, image_ny;
  double image_x_min, image_y_min, image_inv_step_x, image_inv_step_y;
  // optional spectra behind the scalars (include/sart.h: sart_accumulator_len_spectra)
  int32_t spectra, n_radial_bins;
  double radial_inv_bin;
  // Image tile in LDS: pixels [tile_x0, tile_x0 + tile_n) x [tile_y0, tile_y0 + tile_n) are accumulated per workgroup with
  // ds_add_f64 and flushed once at the end of the kernel; everything else goes to global atomics.  tile_n = 0: off.  Pixel
  // (tx, ty) of the tile is cell tile_base + ty tile_n + tx of the tile space (cells 0 .. kTileRingCells - 1: ring 0, free when stage
  // A0 is off, or ring 1's path column in the constant-path variants; kTileExtraCells more behind the tables): tile_base = 0 and
  // tile_n <= kImageTileMax when the ring cells are free, else tile_base = kTileRingCells and tile_n <= kImageTileExtraMax.
  int32_t tile_x0, tile_y0, tile_n, tile_base;
  // SART_ACCUM_FIXED64 (include/sart.h "accumulation mode"): 1 / quantum of the weights and of the squared weights (powers
  // of two); positions use kFixedPositionScale, the reflectivity spectrum kFixedReflectScale.  Unused by the f64 kernels.
  double fx_scale_w, fx_scale_w2;
};
// Fused axion-mass scan (include/sart.h: sart_trace_mass_scan): phase B evaluates the gas-stage conversion probability of every
// surviving ray for the masses of this table and accumulates per mass.  One launch takes up to kScanMaxMasses masses: their
// accumulators ([mass][sum of w, sum of w^2][kScanLanes] f64 or int64 = 512 B per mass; lanes l and l + 32 of a wave share a cell) live
// in the 16 KB of LDS that the image tile uses in the histogram kernels (a scan accumulates no image).
constexpr int kScanMaxMasses = 32;
constexpr int kScanLanes = 32;
struct ScanMass {
  double dm2_abs;                 // |m_gamma^2 - m_a^2| in eV^2 (host: the same IEEE subtraction the single-mass kernel performs)
  double fx_scale_w, fx_scale_w2; // SART_ACCUM_FIXED64: 1 / quantum of the weights and of the squared weights of THIS mass
  double _pad;
};
struct ScanArgs {
  int32_t n_masses, _pad;         // masses of this launch (<= kScanMaxMasses); 0 in every launch that is not a scan
  double* partials;               // [n_blocks][kScanMaxMasses][4] per-workgroup {sum w, sum w^2, rays whose weight vanishes for this mass only, 0}
  ScanMass m[kScanMaxMasses];
};
static_assert(sizeof(ScanMass) == 32 && sizeof(ScanArgs) == 16 + 32 * kScanMaxMasses, "the kernel re-reads ScanArgs with scalar loads at these offsets");
constexpr int kScanPartialSlots = 4;

// Fused angular scan (include/sart.h: sart_trace_angular_scan): the telescope's angles enter a ray at the transformation into the
// telescope's frame (raytracer.nim:1878-1899) and nowhere before it, so trace_angular_scan_kernel samples a ray and takes it through
// bore and pipes ONCE and runs telescope frame -> opaque structures -> shell selection -> mirrors -> weight once per angle of this
// table.  One launch takes up to kAScanMaxAngles angles: per angle a workgroup keeps [sum of w, sum of w^2][kScanLanes] f64 / int64
// cells - in the u5 column of ring 1, which this kernel does not use (the energy draw happens in front of the angle loop: the ring
// carries the energy index) - and four counters in LDS.
constexpr int kAScanMaxAngles = 32;
struct AScanAngle {
  // rotateInY(rotateInX(., turnedX), turnedY) about (0, 0, lT/2) as hoist_setup() evaluates it for this angle (TelRot)
  double rx_c, rx_s, ry_c, ry_s, half_length_telescope;
  double shell0_miss_radius;      // DevParams::shell0_miss_radius of this angle (the bound follows the tilt)
  // third column of the rotation: mx = rx_s, my = -(rx_c ry_s), mz = rx_c ry_c - what phase B needs to find z of pointExitCB in
  // the rotated frame (staged into LDS: a phase-B pass holds rays of several angles)
  double mx, my, mz;
  double _pad;
};
struct AScanArgs {
  int32_t n_angles, _pad;         // angles of this launch (1 .. kAScanMaxAngles)
  double* partials;               // [n_blocks][kAScanMaxAngles][kAScanPartialSlots] per-workgroup sums (plain stores; fold_ascan_kernel adds them)
  AScanAngle a[kAScanMaxAngles];
};
static_assert(sizeof(AScanAngle) == 80 && sizeof(AScanArgs) == 16 + 80 * kAScanMaxAngles, "the kernel re-reads AScanArgs with scalar loads at these offsets");
// per-workgroup partial row of one angle: sum w, sum w^2, N_PASSED, N_HIT_NICKEL, N_PASSED_TILL_WINDOW, N_SHELL_SELECTED,
// (angle 0 only) N_REACHED_TELESCOPE of the workgroup, unused
constexpr int kAScanPartialSlots = 8;

constexpr double kFixedPositionScale = 4294967296.0;        // 2^32 per mm
constexpr double kFixedReflectScale = 1099511627776.0;      // 2^40
constexpr int kFixedLimbBits = 40;                          // two-limb sums: value = hi * 2^40 + lo
// LDS image tile: 16 waves x 128 doubles of ring space (ring 0, or ring 1's path column) + kTileExtraCells doubles behind the tables
// (what the radius CDF leaves free in LDS as 32-bit words, and the end of the 160 KB): 56 x 56 <= 2048 + 1090 cells
constexpr int kTileRingCells = 2048;
constexpr int kTileExtraCells = 1090;
constexpr int kImageTileMax = 56;
constexpr int kImageTileExtraMax = 33;   // 33 x 33 <= kTileExtraCells: the tile of the variants whose rings are all in use

}  // namespace sart
