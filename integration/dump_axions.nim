## dump_axions.nim — writes the `seq[Axion]` of one `calculateFluxFractions` run of jovoy/SolarAxionRayTracing to a raw file,
## for the parity kit of the MI355X port (tools/make_nim_parity_kit.py, tools/nim_raw_to_npz.py, tests/test_nim_stream.py).
##
## STATUS: like integration/sart_ffi.nim this file has never been compiled (no Nim toolchain in the port's build image).
## It uses nothing beyond system / std/os.
##
## Hook (src/raytracer.nim), three lines:
##   1. behind the type section (after `type Axion`, raytracer.nim:192-221):     include dump_axions
##   2. in `calculateFluxFractions`, directly behind `exit(Weave)` (:2772):       maybeDumpAxions(axions)
##   3. run with ONE weave thread, so that the global random stream (randomize(299792458), :276) is consumed in ray order:
##        WEAVE_NUM_THREADS=1 SART_DUMP_AXIONS=/tmp/axions.raw ./raytracer --noPlots
## (`generateResultPlots` may be skipped for the dump: pass generatePlots = false or ignore its output.)
##
## File format: axions.len records of sizeof(Axion) = 208 bytes each, the object as it lies in memory (a Nim object is a C
## struct in declaration order: include/sart.h `sart_axion_t` names the same fields at the same offsets).

import std/os

proc dumpAxions*(axions: seq[Axion], path: string) =
  doAssert sizeof(Axion) == 208, "Axion layout differs from the one the converter expects (include/sart.h: sart_axion_t)"
  doAssert axions.len > 0
  var f: File
  doAssert open(f, path, fmWrite), "cannot open " & path
  defer: f.close()
  let nBytes = axions.len * sizeof(Axion)
  let written = f.writeBuffer(unsafeAddr axions[0], nBytes)
  doAssert written == nBytes, "short write to " & path
  echo "dumped ", axions.len, " Axion records (", nBytes, " bytes) to ", path

proc maybeDumpAxions*(axions: seq[Axion]) =
  ## no-op unless SART_DUMP_AXIONS names a file
  let path = getEnv("SART_DUMP_AXIONS")
  if path.len > 0:
    dumpAxions(axions, path)
