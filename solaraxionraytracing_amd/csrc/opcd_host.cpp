// opcd_host.cpp — reader of the OPCD 3.3 monochromatic opacity files for the emission-table producer (libsart_host.so).
//
// Mirrors the file handling of the reference's solar-model pre-processor (src/readOpacityFile.nim): `parseOpacityOriginal`
// (:235-258) with `parseTableHeader` (:164-178), `parseDensityTab` (:184-217) and `parseTableLine` (:146-162) for the files
// `fmZZ.TTT`, `readMeshFile` (:284-296) for `fm01.mesh`, and the file selection of `calculateOpacities` (:731-745).  What
// the reference keeps as a table of tables of interpolator objects is flattened here into the arrays the GPU kernel
// indexes (sart_opacity_tables_t, include/sart_emission.h): only the density tables the zones of the solar model use
// are converted to numbers, the others are skipped line by line.
//
// The data files are not redistributable and not in the reference repository; tests write files of the same format.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/sart_host.h"

namespace sart_host {
int fail(int code, const std::string& msg);   // raytracer_host.cpp: sets sart_host_last_error()
}
using sart_host::fail;

namespace {

// The elements `calculateOpacities` evaluates an opacity for (:827-834): the proton numbers of `ElementKind` (:28-52) that
// are not in `noElement` (:636).  Hydrogen and helium are looked up like the others (a missing file or density table is
// the reference's KeyError) but only Z > 2 enters the sum (:833).
constexpr int kLookedUp[] = {1, 2, 6, 7, 8, 10, 11, 12, 13, 14, 16, 18, 20, 24, 25, 26, 28};
constexpr int kSummed[] = {6, 7, 8, 10, 11, 12, 13, 14, 16, 18, 20, 24, 25, 26, 28};
constexpr int kNSummed = (int)(sizeof(kSummed) / sizeof(kSummed[0]));

struct Mapped {   // read-only mapping of a whole file
  const char* p = nullptr;
  size_t n = 0;
  int fd = -1;
  ~Mapped() {
    if (p && n) munmap(const_cast<char*>(p), n);
    if (fd >= 0) close(fd);
  }
  bool open_file(const std::string& path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) return false;
    n = (size_t)st.st_size;
    if (n == 0) return true;
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { n = 0; return false; }
    p = static_cast<const char*>(m);
    (void)madvise(m, n, MADV_SEQUENTIAL);
    return true;
  }
};

struct Cursor {   // `nextSlice` of the reference: the text up to the next '\n'
  const char* p;
  const char* end;
  bool more() const { return p < end; }
  void line(const char*& b, const char*& e) {
    b = p;
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    e = nl ? nl : end;
    p = nl ? nl + 1 : end;
    if (e > b && e[-1] == '\r') --e;   // files that went through a DOS transfer
  }
};

bool blank(const char* b, const char* e) {
  for (; b < e; ++b)
    if (*b != ' ' && *b != '\t' && *b != '\r') return false;
  return true;
}

// strutils.parseInt on a stripped line / scanf("$s$i"): optional blanks, optional sign, digits.  `rest` = first unread char.
bool parse_int(const char* b, const char* e, long& v, const char** rest = nullptr) {
  while (b < e && (*b == ' ' || *b == '\t')) ++b;
  const char* s = b;
  if (s < e && (*s == '+' || *s == '-')) ++s;
  if (s >= e || *s < '0' || *s > '9') return false;
  long x = 0;
  for (; s < e && *s >= '0' && *s <= '9'; ++s) {
    if (x > 100000000L) return false;
    x = x * 10 + (*s - '0');
  }
  v = (*b == '-') ? -x : x;
  if (rest) *rest = s;
  return true;
}

// Correctly rounded decimal -> f64 (Fortran's `D` exponent letter accepted).
bool parse_double(const char* b, const char* e, double& v) {
  if (b < e && *b == '+') ++b;
  const char* d = static_cast<const char*>(memchr(b, 'D', (size_t)(e - b)));
  if (!d) d = static_cast<const char*>(memchr(b, 'd', (size_t)(e - b)));
  if (d) {
    char buf[64];
    const size_t n = (size_t)(e - b);
    if (n >= sizeof(buf)) return false;
    memcpy(buf, b, n);
    buf[d - b] = 'E';
    const auto r = std::from_chars(buf, buf + n, v);
    return r.ec == std::errc() && r.ptr == buf + n;
  }
  const auto r = std::from_chars(b, e, v);
  return r.ec == std::errc() && r.ptr == e;
}

// parseTableLine (:146-162): blank-separated numbers; one number = the opacity, two = abscissa and opacity; more is an error.
bool parse_table_line(const char* b, const char* e, double& x, double& y, bool& has_x) {
  int n = 0;
  double v[2] = {0.0, 0.0};
  while (b < e) {
    while (b < e && *b == ' ') ++b;
    if (b >= e) break;
    const char* t = b;
    while (t < e && *t != ' ') ++t;
    if (n == 2) return false;
    if (!parse_double(b, t, v[n])) return false;
    ++n;
    b = t;
  }
  if (n == 0) return false;
  has_x = (n == 2);
  x = has_x ? v[0] : 0.0;
  y = has_x ? v[1] : v[0];
  return true;
}

struct DensityTable {
  int density = 0;
  bool implicit_x = false;   // 10000-line table: abscissa = line count + 1 (:206-208)
  std::vector<double> x, y;
};

struct OpcdFile {
  int element = 0, temp = 0;
  std::vector<int> densities;          // every table of the file, in file order
  std::vector<int> lengths;
  std::vector<DensityTable> tables;    // the ones asked for
};

std::string base_name(const std::string& path) {
  const size_t s = path.find_last_of('/');
  return s == std::string::npos ? path : path.substr(s + 1);
}

// parseOpacityOriginal (:235-258).  `want` = densities to convert (nullptr: all).  Deviations from the reference, all on
// input it would mis-read silently: a table cut short by the end of the file is an error (the reference keeps the short
// table), blank lines after the last table are ignored (the reference raises on them).
int parse_file(const std::string& path, const std::set<int>* want, OpcdFile& out, std::string& err) {
  const std::string name = base_name(path);
  // fname[2 .. 3] = element, fname[5 .. ^1] = temperature index (:238-242)
  long z = 0, t = 0;
  const char *rz = nullptr, *rt = nullptr;
  if (name.size() < 6 || name[0] != 'f' || name[1] != 'm' || name[4] != '.' || !parse_int(name.data() + 2, name.data() + 4, z, &rz) ||
      rz != name.data() + 4 || !parse_int(name.data() + 5, name.data() + name.size(), t, &rt) || rt != name.data() + name.size() || z < 0 || t < 0) {
    err = "cannot read element and temperature from the file name `" + name + "` (expected fmZZ.TTT)";
    return SART_ERR_INVALID_ARGUMENT;
  }
  out.element = (int)z;
  out.temp = (int)t;
  Mapped m;
  if (!m.open_file(path)) {
    err = "Could not open file " + path;   // IOError of :266
    return SART_ERR_INVALID_ARGUMENT;
  }
  Cursor c{m.p, m.p + m.n};
  const char *b, *e;
  if (c.more()) c.line(b, e);   // file header
  while (c.more()) {
    c.line(b, e);
    if (blank(b, e)) {
      bool only_blank = true;
      Cursor look = c;
      while (look.more()) {
        const char *b2, *e2;
        look.line(b2, e2);
        if (!blank(b2, e2)) { only_blank = false; break; }
      }
      if (only_blank) break;
    }
    long density = 0;
    if (!parse_int(b, e, density)) {
      err = name + ": Could not parse header line 1: " + std::string(b, e);   // :175
      return SART_ERR_INVALID_ARGUMENT;
    }
    if (!c.more()) { err = name + ": file ends inside a table header"; return SART_ERR_INVALID_ARGUMENT; }
    c.line(b, e);   // header line 2: unused (:176-178)
    if (!c.more()) { err = name + ": file ends inside a table header"; return SART_ERR_INVALID_ARGUMENT; }
    c.line(b, e);
    long count = 0;
    const char* rest = nullptr;
    if (!parse_int(b, e, count, &rest) || !blank(rest, e) || count < 0) {
      err = name + ": cannot read the number of table lines from `" + std::string(b, e) + "`";
      return SART_ERR_INVALID_ARGUMENT;
    }
    if (count == 0) count = 10000;   // :256
    const bool implicit_x = (count == 10000);
    const bool keep = !want || want->count((int)density) != 0;
    out.densities.push_back((int)density);
    out.lengths.push_back((int)count);
    DensityTable tab;
    tab.density = (int)density;
    tab.implicit_x = implicit_x;
    if (keep) {
      tab.y.reserve((size_t)count);
      if (!implicit_x) tab.x.reserve((size_t)count);
    }
    for (long i = 0; i < count; ++i) {
      if (!c.more()) {
        err = name + ": table of density " + std::to_string(density) + " ends after " + std::to_string(i) + " of " + std::to_string(count) + " lines";
        return SART_ERR_INVALID_ARGUMENT;
      }
      c.line(b, e);
      if (!keep) continue;
      double x = 0.0, y = 0.0;
      bool has_x = false;
      if (!parse_table_line(b, e, x, y, has_x)) {
        err = name + ": Parsing opacity table in line `" + std::string(b, e) + "` failed!";   // :161
        return SART_ERR_INVALID_ARGUMENT;
      }
      if (!implicit_x) {
        if (!has_x) {
          err = name + ": table of density " + std::to_string(density) + " has " + std::to_string(count) + " lines and no abscissa column";
          return SART_ERR_INVALID_ARGUMENT;
        }
        tab.x.push_back(x);
      }
      tab.y.push_back(y);
    }
    if (keep) {
      // densityTab[h1.density] = ... (:258): a later table of the same density replaces the earlier one
      auto it = std::find_if(out.tables.begin(), out.tables.end(), [&](const DensityTable& d) { return d.density == tab.density; });
      if (it != out.tables.end()) *it = std::move(tab);
      else out.tables.push_back(std::move(tab));
    }
  }
  return 0;
}

// readMeshDataFile (:278-282): blank-separated columns with a header line; the column `u`.
int read_mesh(const std::string& path, std::vector<double>& u, std::string& err) {
  Mapped m;
  if (!m.open_file(path)) {
    err = "Could not read mesh file `fm01.mesh` at path: " + path;   // :292-294
    return SART_ERR_INVALID_ARGUMENT;
  }
  Cursor c{m.p, m.p + m.n};
  const char *b, *e;
  if (!c.more()) { err = path + ": empty mesh file"; return SART_ERR_INVALID_ARGUMENT; }
  c.line(b, e);
  int col = -1, k = 0;
  for (const char* s = b; s < e;) {
    while (s < e && *s == ' ') ++s;
    if (s >= e) break;
    const char* t = s;
    while (t < e && *t != ' ') ++t;
    if (t - s == 1 && *s == 'u') col = k;
    ++k;
    s = t;
  }
  if (col < 0) { err = path + ": no column `u` in the header line"; return SART_ERR_INVALID_ARGUMENT; }
  while (c.more()) {
    c.line(b, e);
    if (blank(b, e)) continue;
    int i = 0;
    bool got = false;
    for (const char* s = b; s < e;) {
      while (s < e && *s == ' ') ++s;
      if (s >= e) break;
      const char* t = s;
      while (t < e && *t != ' ') ++t;
      if (i == col) {
        double v;
        if (!parse_double(s, t, v)) { err = path + ": cannot read `" + std::string(s, t) + "`"; return SART_ERR_INVALID_ARGUMENT; }
        u.push_back(v);
        got = true;
        break;
      }
      ++i;
      s = t;
    }
    if (!got) { err = path + ": line without a value in column `u`: " + std::string(b, e); return SART_ERR_INVALID_ARGUMENT; }
  }
  return 0;
}

bool file_exists(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

std::string mono_dir(const char* opcd_path) { return std::string(opcd_path) + "/OPCD_3.3/mono"; }   // :290, :739

std::string file_of(const std::string& dir, int z, int temp) {
  char buf[32];
  snprintf(buf, sizeof buf, "/fm%02d.%d", z, temp);   // &"/OPCD_3.3/mono/fm{Z:02}.{temp}" (:739)
  return dir + buf;
}

}  // namespace

struct sart_opcd_set {
  std::vector<double> u_mesh;
  std::vector<int32_t> slot_of_zone, element_z, table_len, slot_temp, slot_ne;
  std::vector<int64_t> y_begin, x_begin;
  std::vector<double> table_x, table_y;
  sart_opacity_tables_t view{};
};

extern "C" {

// n_Z of the first loop of calculateOpacities (:655-679), per proton number.
int sart_host_solar_number_densities(const double* rho, const double* mass_fractions, int32_t n_radii, double* n_z_out) {
  if (!rho || !mass_fractions || !n_z_out || n_radii < 1) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_solar_number_densities: bad argument");
  static const double a[29] = {1.0078,  4.0026,  3.0160,  12.0000, 13.0033, 14.0030, 15.0001, 15.9949, 16.9991, 17.9991,
                               20.1797, 22.9897, 24.3055, 26.9815, 28.085,  30.9737, 32.0675, 35.4515, 39.8775, 39.0983,
                               40.078,  44.9559, 47.867,  50.9415, 51.9961, 54.9380, 55.845,  58.9331, 58.6934};   // atomicMass :120-124
  const double amu = 1.6605e-24;   // :651
  for (int32_t i = 0; i < n_radii; ++i) {
    const double* x = mass_fractions + (size_t)i * 29;
    double* n = n_z_out + (size_t)i * 29;
    for (int k = 0; k < 29; ++k) n[k] = 0.0;   // newSeq[float](29): entries the loop never writes stay 0
    auto single = [&](int idx) { return (x[idx] / a[idx]) * (rho[i] / amu); };   // template n(idx) :658-659
    n[1] = single(0);   // hydrogen :661
    // iZmult = 1, 2, 3: the isotope pairs (He4, He3), (C12, C13), (N14, N15) with their mean mass -> n_Z[2], n_Z[6], n_Z[7] (:662-670)
    for (int m = 1; m <= 3; ++m) {
      const int iz = m * 2;
      const double v = (x[iz - 1] + x[iz]) / ((a[iz - 1] * x[iz - 1] + a[iz] * x[iz]) / (x[iz - 1] + x[iz])) * rho[i] / amu;
      if (m == 1) n[iz] = v;
      else n[m + 4] = v;
    }
    n[8] = (x[7] + x[8] + x[9]) / ((x[7] * a[7] + x[8] * a[8] + x[9] * a[9]) / (x[7] + x[8] + x[9])) * rho[i] / amu;   // oxygen :672-675
    for (int iz = 10; iz < 29; ++iz) n[iz] = single(iz);   // :676-677 (from neon on the column index equals the proton number)
  }
  return 0;
}

int sart_host_opcd_read_mesh(const char* path, double* u_out, int32_t capacity, int32_t* n_out) {
  if (!path || !n_out) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_opcd_read_mesh: bad argument");
  std::vector<double> u;
  std::string err;
  if (int rc = read_mesh(path, u, err)) return fail(rc, err);
  *n_out = (int32_t)u.size();
  if (u_out) {
    if ((size_t)capacity < u.size()) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_opcd_read_mesh: buffer too small");
    std::copy(u.begin(), u.end(), u_out);
  }
  return 0;
}

int sart_host_opcd_file_info(const char* path, int32_t* element, int32_t* temp_index, int32_t* n_tables, int32_t* densities_out,
                             int32_t* lengths_out, int32_t capacity) {
  if (!path || !n_tables) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_opcd_file_info: bad argument");
  OpcdFile f;
  std::string err;
  const std::set<int> none;
  if (int rc = parse_file(path, &none, f, err)) return fail(rc, err);
  if (element) *element = f.element;
  if (temp_index) *temp_index = f.temp;
  *n_tables = (int32_t)f.densities.size();
  for (size_t i = 0; i < f.densities.size() && (int32_t)i < capacity; ++i) {
    if (densities_out) densities_out[i] = f.densities[i];
    if (lengths_out) lengths_out[i] = f.lengths[i];
  }
  return 0;
}

int sart_host_opcd_read_table(const char* path, int32_t density, double* x_out, double* y_out, int32_t capacity, int32_t* n_out) {
  if (!path || !n_out) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_opcd_read_table: bad argument");
  OpcdFile f;
  std::string err;
  const std::set<int> want{density};
  if (int rc = parse_file(path, &want, f, err)) return fail(rc, err);
  if (f.tables.empty()) return fail(SART_ERR_INVALID_ARGUMENT, base_name(path) + ": no table of density " + std::to_string(density));
  const DensityTable& t = f.tables[0];
  *n_out = (int32_t)t.y.size();
  if ((x_out || y_out) && (size_t)capacity < t.y.size()) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_opcd_read_table: buffer too small");
  for (size_t i = 0; i < t.y.size(); ++i) {
    if (x_out) x_out[i] = t.implicit_x ? (double)(i + 1) : t.x[i];
    if (y_out) y_out[i] = t.y[i];
  }
  return 0;
}

int sart_host_opcd_load(const char* opcd_path, const sart_solar_zone_t* zones, int32_t n_radii, int32_t n_threads, sart_opcd_set** out) {
  if (!opcd_path || !zones || !out || n_radii < 1) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_opcd_load: bad argument");
  *out = nullptr;
  auto set = std::make_unique<sart_opcd_set>();
  const std::string dir = mono_dir(opcd_path);
  {
    std::string err;
    if (int rc = read_mesh(dir + "/fm01.mesh", set->u_mesh, err)) return fail(rc, err);
    if (set->u_mesh.size() != 10001)   // doAssert df.len == 10001 (:281)
      return fail(SART_ERR_INVALID_ARGUMENT, "fm01.mesh: " + std::to_string(set->u_mesh.size()) + " lines, expected 10001");
    for (size_t i = 1; i < set->u_mesh.size(); ++i)
      if (!(set->u_mesh[i] > set->u_mesh[i - 1])) return fail(SART_ERR_INVALID_ARGUMENT, "fm01.mesh: column u does not ascend at line " + std::to_string(i));
  }
  // slots = distinct (temperature, density) pairs in the order the zones meet them
  std::map<std::pair<int, int>, int> slot_of;
  std::map<int, std::set<int>> want;   // temperature -> densities
  set->slot_of_zone.resize((size_t)n_radii);
  for (int32_t r = 0; r < n_radii; ++r) {
    const std::pair<int, int> key{zones[r].temp_index, zones[r].ne_index};
    auto it = slot_of.find(key);
    if (it == slot_of.end()) {
      it = slot_of.emplace(key, (int)slot_of.size()).first;
      set->slot_temp.push_back(key.first);
      set->slot_ne.push_back(key.second);
    }
    set->slot_of_zone[(size_t)r] = it->second;
    want[key.first].insert(key.second);
  }
  // the files: for temp in toSet(temperatures), for Z in ElementKind: if existsFile(...): parse (:734-745); an element the
  // cell loop looks up without a file is the reference's KeyError (:831)
  struct Job { int temp, z; const std::set<int>* want; std::string path; OpcdFile file; int rc = 0; std::string err; };
  std::vector<Job> jobs;
  for (const auto& tw : want)
    for (int z : kLookedUp) {
      Job j;
      j.temp = tw.first;
      j.z = z;
      j.want = &tw.second;
      j.path = file_of(dir, z, tw.first);
      if (!file_exists(j.path))
        return fail(SART_ERR_INVALID_ARGUMENT, "no opacity file " + j.path + " (element " + std::to_string(z) + " at temperature index " + std::to_string(tw.first) + ")");
      jobs.push_back(std::move(j));
    }
  {
    int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::min<int>(nt, (int)jobs.size());
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= jobs.size()) return;
        Job& j = jobs[i];
        const bool summed = j.z > 2;
        const std::set<int> none;
        // hydrogen and helium: the table must exist, its numbers are not used (:833)
        j.rc = parse_file(j.path, summed ? j.want : &none, j.file, j.err);
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  std::map<std::pair<int, int>, const OpcdFile*> file_at;   // (temp, z)
  for (const Job& j : jobs) {
    if (j.rc) return fail(j.rc, j.err);
    if (j.file.element != j.z || j.file.temp != j.temp) return fail(SART_ERR_INVALID_ARGUMENT, "unexpected file name " + j.path);
    file_at[{j.temp, j.z}] = &j.file;
    for (int ne : *j.want)
      if (std::find(j.file.densities.begin(), j.file.densities.end(), ne) == j.file.densities.end())
        return fail(SART_ERR_INVALID_ARGUMENT, base_name(j.path) + ": no table of density " + std::to_string(ne) + " (needed by the solar model)");
  }
  const size_t n_slots = slot_of.size();
  set->element_z.assign(kSummed, kSummed + kNSummed);
  set->y_begin.resize(n_slots * kNSummed);
  set->x_begin.resize(n_slots * kNSummed);
  set->table_len.resize(n_slots * kNSummed);
  for (size_t s = 0; s < n_slots; ++s)
    for (int k = 0; k < kNSummed; ++k) {
      const OpcdFile* f = file_at[{set->slot_temp[s], kSummed[k]}];
      const int ne = set->slot_ne[s];
      const auto it = std::find_if(f->tables.begin(), f->tables.end(), [&](const DensityTable& d) { return d.density == ne; });
      const DensityTable& t = *it;   // present: checked above
      if (t.y.size() < 2) return fail(SART_ERR_INVALID_ARGUMENT, "a table of fewer than two lines cannot be interpolated");
      const size_t cell = s * kNSummed + (size_t)k;
      set->y_begin[cell] = (int64_t)set->table_y.size();
      set->table_len[cell] = (int32_t)t.y.size();
      set->table_y.insert(set->table_y.end(), t.y.begin(), t.y.end());
      if (t.implicit_x) {
        set->x_begin[cell] = -1;
      } else {
        for (size_t i = 1; i < t.x.size(); ++i)
          if (!(t.x[i] > t.x[i - 1])) return fail(SART_ERR_INVALID_ARGUMENT, "abscissae of a table do not ascend (density " + std::to_string(ne) + ")");
        set->x_begin[cell] = (int64_t)set->table_x.size();
        set->table_x.insert(set->table_x.end(), t.x.begin(), t.x.end());
      }
    }
  sart_opacity_tables_t& v = set->view;
  v.u_mesh = set->u_mesh.data();
  v.n_mesh = (int32_t)set->u_mesh.size();
  v.n_slots = (int32_t)n_slots;
  v.slot_of_zone = set->slot_of_zone.data();
  v.element_z = set->element_z.data();
  v.n_elements = kNSummed;
  v._pad = 0;
  v.table_y_begin = set->y_begin.data();
  v.table_x_begin = set->x_begin.data();
  v.table_len = set->table_len.data();
  v.table_x = set->table_x.empty() ? nullptr : set->table_x.data();
  v.table_y = set->table_y.data();
  v.n_table_x = (int64_t)set->table_x.size();
  v.n_table_y = (int64_t)set->table_y.size();
  *out = set.release();
  return 0;
}

const sart_opacity_tables_t* sart_host_opcd_tables(const sart_opcd_set* set) { return set ? &set->view : nullptr; }

int sart_host_opcd_slot(const sart_opcd_set* set, int32_t slot, int32_t* temp_index, int32_t* ne_index) {
  if (!set || slot < 0 || (size_t)slot >= set->slot_temp.size()) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_opcd_slot: bad argument");
  if (temp_index) *temp_index = set->slot_temp[(size_t)slot];
  if (ne_index) *ne_index = set->slot_ne[(size_t)slot];
  return 0;
}

void sart_host_opcd_free(sart_opcd_set* set) { delete set; }

}  // extern "C"
