// uvol_host.cpp — see uvol_host.hpp.  Mirrors scripts/Encoder.py (file:line cited per function).
#include "uvol_host.hpp"
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <dlfcn.h>
#include <fstream>
#include <sstream>
#include <sys/stat.h>
#include <thread>
#include <zlib.h>

namespace uvolh {

// ------------------------------------------------------------------ JSON
namespace {
struct P {
  const std::string &s; size_t i = 0; std::string err;
  explicit P(const std::string &t) : s(t) {}
  void ws() {
    for (;;) {
      while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
      if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') i++; continue; }
      if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') { i += 2; while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) i++; i += 2; continue; }
      if (i < s.size() && s[i] == '#') { while (i < s.size() && s[i] != '\n') i++; continue; }
      break;
    }
  }
  bool fail(const char *m) { if (err.empty()) { err = m; err += " at offset " + std::to_string(i); } return false; }
  bool str(std::string &o) {
    if (s[i] != '"') return fail("expected string");
    i++; o.clear();
    while (i < s.size() && s[i] != '"') {
      char c = s[i++];
      if (c == '\\' && i < s.size()) {
        char e = s[i++];
        switch (e) { case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break; case 'b': o += '\b'; break; case 'f': o += '\f'; break;
          case 'u': { unsigned v = 0; for (int k = 0; k < 4 && i < s.size(); k++) v = v * 16 + (unsigned)(std::isdigit((unsigned char)s[i]) ? s[i] - '0' : (std::tolower(s[i]) - 'a' + 10)), i++;
            if (v < 0x80) o += (char)v; else if (v < 0x800) { o += (char)(0xC0 | (v >> 6)); o += (char)(0x80 | (v & 63)); } else { o += (char)(0xE0 | (v >> 12)); o += (char)(0x80 | ((v >> 6) & 63)); o += (char)(0x80 | (v & 63)); } break; }
          default: o += e; }
      } else o += c;
    }
    if (i >= s.size()) return fail("unterminated string");
    i++; return true;
  }
  bool val(Json &j) {
    ws(); if (i >= s.size()) return fail("unexpected end");
    char c = s[i];
    if (c == '{') {
      i++; j.type = Json::Obj; ws();
      if (i < s.size() && s[i] == '}') { i++; return true; }
      for (;;) {
        ws(); if (i < s.size() && s[i] == '}') { i++; return true; }          // trailing comma
        std::string k; if (!str(k)) return false; ws();
        if (i >= s.size() || s[i] != ':') return fail("expected ':'");
        i++;
        Json v; if (!val(v)) return false; j.set(k, std::move(v)); ws();
        if (i < s.size() && s[i] == ',') { i++; continue; }
        if (i < s.size() && s[i] == '}') { i++; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      i++; j.type = Json::Arr; ws();
      if (i < s.size() && s[i] == ']') { i++; return true; }
      for (;;) {
        ws(); if (i < s.size() && s[i] == ']') { i++; return true; }
        Json v; if (!val(v)) return false; j.arr.push_back(std::move(v)); ws();
        if (i < s.size() && s[i] == ',') { i++; continue; }
        if (i < s.size() && s[i] == ']') { i++; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') { j.type = Json::Str; return str(j.str); }
    if (!s.compare(i, 4, "true")) { i += 4; j.type = Json::Bool; j.b = true; return true; }
    if (!s.compare(i, 5, "false")) { i += 5; j.type = Json::Bool; j.b = false; return true; }
    if (!s.compare(i, 4, "null")) { i += 4; j.type = Json::Null; return true; }
    size_t st = i; bool isint = true;
    if (i < s.size() && (s[i] == '-' || s[i] == '+')) i++;
    while (i < s.size() && (std::isdigit((unsigned char)s[i]) || s[i] == '.' || s[i] == 'e' || s[i] == 'E' || s[i] == '-' || s[i] == '+')) { if (!std::isdigit((unsigned char)s[i])) isint = false; i++; }
    if (i == st) return fail("unexpected character");
    j.type = Json::Num; j.num = std::strtod(s.substr(st, i - st).c_str(), nullptr); j.is_int = isint; return true;
  }
};
void dump(const Json &j, std::string &o) {
  switch (j.type) {
    case Json::Null: o += "null"; break;
    case Json::Bool: o += j.b ? "true" : "false"; break;
    case Json::Num: { char b[64]; if (j.is_int || j.num == std::floor(j.num)) std::snprintf(b, sizeof b, "%lld", (long long)j.num); else std::snprintf(b, sizeof b, "%.17g", j.num); o += b; break; }
    case Json::Str: o += '"'; for (char c : j.str) { if (c == '"' || c == '\\') { o += '\\'; o += c; } else if (c == '\n') o += "\\n"; else if (c == '\t') o += "\\t"; else o += c; } o += '"'; break;
    case Json::Arr: o += '['; for (size_t k = 0; k < j.arr.size(); k++) { if (k) o += ", "; dump(j.arr[k], o); } o += ']'; break;
    case Json::Obj: o += '{'; for (size_t k = 0; k < j.obj.size(); k++) { if (k) o += ", "; o += '"'; o += j.obj[k].first; o += "\": "; dump(j.obj[k].second, o); } o += '}'; break;
  }
}
}  // namespace
bool json_parse(const std::string &text, Json &out, std::string &err) { P p(text); out = Json(); if (!p.val(out)) { err = p.err; return false; } p.ws(); if (p.i != text.size()) { err = "trailing characters"; return false; } return true; }
std::string json_dump(const Json &j) { std::string o; dump(j, o); return o; }

// ------------------------------------------------------------------ patterns
std::string convert_pounds_to_c_style(const std::string &s) {       // scripts/Encoder.py:16-19
  size_t n = std::count(s.begin(), s.end(), '#');
  if (!n) { std::string o = "%00u"; for (char c : s) { o += c; o += "%00u"; } return o; }   // python: s.replace('', '%00u')
  std::string run(n, '#'), rep = "%0" + std::to_string(n) + "u", out = s;
  for (size_t p = out.find(run); p != std::string::npos; p = out.find(run, p + rep.size())) out.replace(p, n, rep);
  return out;
}
bool match_pattern(const std::string &pattern, const std::string &fn) {   // scripts/Encoder.py:87-100
  const size_t pad = std::count(pattern.begin(), pattern.end(), '#');
  const std::string run(pad, '#');
  size_t pi = pattern.find(run);                          // python str.find(''): 0
  if (pi == std::string::npos) {                          // python find() == -1: slices with -1
    // pattern[:-1] == file_name[:-1] and pattern[-1+pad:] == file_name[-1+pad:] and file_name[-1:-1+pad].isdigit()
    auto sl = [](const std::string &t, long a, long b) { long n = (long)t.size(); if (a < 0) a += n; if (b < 0) b += n; a = std::max(0L, std::min(a, n)); b = std::max(0L, std::min(b, n)); return a < b ? t.substr((size_t)a, (size_t)(b - a)) : std::string(); };
    auto from = [](const std::string &t, long a) { long n = (long)t.size(); if (a < 0) a += n; a = std::max(0L, std::min(a, n)); return t.substr((size_t)a); };
    std::string mid = sl(fn, -1, -1 + (long)pad);
    bool dig = !mid.empty() && std::all_of(mid.begin(), mid.end(), [](char c) { return std::isdigit((unsigned char)c); });
    return sl(pattern, 0, -1) == sl(fn, 0, -1) && from(pattern, -1 + (long)pad) == from(fn, -1 + (long)pad) && dig;
  }
  auto sub = [](const std::string &t, size_t a, size_t b) { a = std::min(a, t.size()); b = std::min(b, t.size()); return a < b ? t.substr(a, b - a) : std::string(); };
  std::string mid = sub(fn, pi, pi + pad);
  bool dig = !mid.empty() && std::all_of(mid.begin(), mid.end(), [](char c) { return std::isdigit((unsigned char)c); });
  return sub(pattern, 0, pi) == sub(fn, 0, pi) && sub(pattern, pi + pad, std::string::npos) == sub(fn, pi + pad, std::string::npos) && dig;
}
static std::string strip_brackets(const std::string &pattern) {
  const size_t pad = std::count(pattern.begin(), pattern.end(), '#');
  if (!pad) return pattern;
  const std::string br = "[" + std::string(pad, '#') + "]";
  size_t p = pattern.find(br);
  if (p == std::string::npos) return pattern;
  std::string o = pattern; o.replace(p, br.size(), std::string(pad, '#')); return o;
}
bool match_pattern_lenient(const std::string &pattern, const std::string &fn) { return match_pattern(pattern, fn) || match_pattern(strip_brackets(pattern), fn); }
std::string format_index(const std::string &pattern, unsigned index) {
  std::string p = strip_brackets(pattern);
  const size_t pad = std::count(p.begin(), p.end(), '#');
  if (!pad) return p;
  char buf[64]; std::snprintf(buf, sizeof buf, "%0*u", (int)pad, index);
  size_t at = p.find(std::string(pad, '#'));
  if (at == std::string::npos) return p;
  p.replace(at, pad, buf); return p;
}

// ------------------------------------------------------------------ config
std::string check_all_fields(const Json &cfg) {                       // scripts/Encoder.py:45-84
  static const char *mand[] = { "name", "GEOMETRY_FRAME_RATE", "TEXTURE_FRAME_RATE", "OutputDirectory", "KTX2_BATCH_SIZE" };
  std::string missing;
  for (const char *k : mand) { const Json *v = cfg.get(k); if (!v || v->type == Json::Null) { if (!missing.empty()) missing += ", "; missing += std::string("'") + k + "'"; } }
  if (!missing.empty()) return "Missing mandatory fields:  [" + missing + "]";
  auto truthy = [&](const char *k) { const Json *v = cfg.get(k); return v && v->truthy(); };
  if (!(truthy("ABCFilePath") || truthy("OBJFilesPath") || truthy("DRACOFilesPath"))) return "Path to Geometry data is not specified";
  if (truthy("ImagesPath")) {
    const Json *a = cfg.get("KTX2_FIRST_FILE"), *b = cfg.get("KTX2_FILE_COUNT");
    auto is_int = [](const Json *v) { return v && ((v->type == Json::Num && v->is_int) || v->type == Json::Bool); };   // isinstance(True, int) is True in python
    if (!(is_int(a) && is_int(b))) return "When ImagesPath is given, you must specify `KTX2_FIRST_FILE` and `KTX2_FILE_COUNT`";
  } else if (!truthy("KTX2FilesPath")) return "Path to Texture data is not specified";
  return "";
}
bool load_config(const std::string &text, Config &c, std::string &err) {
  if (!json_parse(text, c.raw, err)) return false;
  if (c.raw.type != Json::Obj) { err = "config is not a JSON object"; return false; }
  err = check_all_fields(c.raw);
  if (!err.empty()) return false;
  auto S = [&](const char *k) { const Json *v = c.raw.get(k); return v && v->type == Json::Str ? v->str : std::string(); };
  auto N = [&](const char *k, double d) { const Json *v = c.raw.get(k); return v && v->type == Json::Num ? v->num : d; };
  c.name = S("name"); c.obj_files_path = S("OBJFilesPath"); c.draco_files_path = S("DRACOFilesPath"); c.images_path = S("ImagesPath");
  c.ktx2_files_path = S("KTX2FilesPath"); c.output_directory = S("OutputDirectory"); c.audio_url = S("AudioURL"); c.abc_file_path = S("ABCFilePath");
  c.q_position = (int)N("Q_POSITION_ATTR", 11); c.q_texture = (int)N("Q_TEXTURE_ATTR", 10); c.q_normal = (int)N("Q_NORMAL_ATTR", 8);
  c.q_generic = (int)N("Q_GENERIC_ATTR", 8); c.compression_level = (int)N("DRACO_COMPRESSION_LEVEL", 7);
  c.ktx2_first_file = (int)N("KTX2_FIRST_FILE", 0); c.ktx2_file_count = (int)N("KTX2_FILE_COUNT", 0); c.ktx2_batch_size = (int)N("KTX2_BATCH_SIZE", 0);
  c.geometry_frame_rate = N("GEOMETRY_FRAME_RATE", 0); c.texture_frame_rate = N("TEXTURE_FRAME_RATE", 0);
  if (c.ktx2_batch_size <= 0) { err = "KTX2_BATCH_SIZE must be a positive integer"; return false; }
  return true;
}
std::string config_template() {                                          // scripts/Encoder.py:163-186 (same fields, same defaults)
  return "{\n  \"name\": \"\",\n  \"draco_encoder\": \"\", // unused by uvolenc (in-process HIP codec)\n  \"basisu\": \"\", // unused by uvolenc (in-process HIP codec)\n"
         "  \"ABCFilePath\": \"\",\n  \"OBJFilesPath\": \"\", // pattern with hashes. eg: OBJ/frame_[#####].obj\n  \"DRACOFilesPath\": \"\", // pattern with hashes\n"
         "  \"Q_POSITION_ATTR\": 11, // quantization bits for the position attribute, default=11.\n  \"Q_TEXTURE_ATTR\": 10, // quantization bits for the texture coordinate attribute, default=10.\n"
         "  \"Q_NORMAL_ATTR\": 8, // quantization bits for the normal vector attribute, default=8.\n  \"Q_GENERIC_ATTR\": 8, // quantization bits for any generic attribute, default=8.\n"
         "  \"DRACO_COMPRESSION_LEVEL\": 7, // compression level [0-10], most=10, least=0, default=7.\n  \"ImagesPath\": \"\", // pattern with hashes.\n"
         "  \"KTX2_FIRST_FILE\": 0, // The index of the first file in above pattern. Eg: If PNG/frame_001.png is first texture, this field should be 1\n"
         "  \"KTX2_FILE_COUNT\": 0,\n  \"KTX2_BATCH_SIZE\": 7,\n  \"KTX2FilesPath\": \"\",\n  \"GEOMETRY_FRAME_RATE\": 30,\n  \"TEXTURE_FRAME_RATE\": 30,\n  \"OutputDirectory\": \"\"\n}\n";
}

// ------------------------------------------------------------------ files
bool write_file(const std::string &path, const void *data, size_t n) { FILE *f = std::fopen(path.c_str(), "wb"); if (!f) return false; bool ok = std::fwrite(data, 1, n, f) == n; std::fclose(f); return ok; }
unsigned effective_cpus() {
  unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[64] = {0}; long long period = 0;
    if (std::fscanf(f, "%63s %lld", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) {
      const long long quota = std::atoll(a);
      if (quota > 0) hw = (unsigned)std::max<long long>(1, std::min<long long>(hw, (quota + period - 1) / period));
    }
    std::fclose(f);
  }
  return hw;
}
bool read_file(const std::string &path, std::vector<uint8_t> &d) {
  FILE *f = std::fopen(path.c_str(), "rb"); if (!f) return false;
  std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  d.resize(n > 0 ? (size_t)n : 0); bool ok = n >= 0 && std::fread(d.data(), 1, d.size(), f) == d.size(); std::fclose(f); return ok;
}
long file_size(const std::string &path) { struct stat st; return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) ? (long)st.st_size : -1; }
bool read_file_into(const std::string &path, uint8_t *dst, size_t n) {
  FILE *f = std::fopen(path.c_str(), "rb"); if (!f) return false;
  const bool ok = std::fread(dst, 1, n, f) == n && std::fgetc(f) == EOF; std::fclose(f); return ok;
}
std::vector<std::string> list_dir(const std::string &dir) {
  std::vector<std::string> v; DIR *d = opendir(dir.empty() ? "." : dir.c_str()); if (!d) return v;
  while (dirent *e = readdir(d)) { std::string n = e->d_name; if (n != "." && n != "..") v.push_back(n); }
  closedir(d); std::sort(v.begin(), v.end()); return v;
}
bool make_dirs(const std::string &dir) {
  if (dir.empty()) return true;
  std::string cur;
  for (size_t i = 0; i <= dir.size(); i++) {
    if (i == dir.size() || dir[i] == '/') { if (!cur.empty() && cur != "/") { if (mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false; } }
    if (i < dir.size()) cur += dir[i];
  }
  return true;
}

// ------------------------------------------------------------------ OBJ (what draco_encoder's OBJ front end extracts: v / vt / vn / f, polygons fanned)
// Decimal -> float without strtof (locale-aware, ~150 ns per number: 0.8 M numbers per 100 k-vertex frame).  Up to 19 significant
// digits go into a 64-bit integer; when the integer is below 2^53 and the decimal exponent within +-22, mant * 10^e (or / 10^-e) is ONE
// correctly rounded double operation on exact operands, i.e. the correctly rounded double of the decimal.  Narrowing that double
// to float equals the correctly rounded float unless the double sits within an ulp of a float rounding boundary (double
// rounding); those cases - and everything else unusual (more digits, huge exponents, inf / nan, hex) - go to strtof, so the
// result is always exactly strtof's.
static inline bool fast_float(const char *&q, const char *le, float &out) {
  static const double P10[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
  const char *p = q; bool neg = false;
  if (p < le && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
  uint64_t mant = 0; int nd = 0, e10 = 0; bool any = false;
  while (p < le && *p >= '0' && *p <= '9') { if (nd < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) nd++; } else e10++; any = true; p++; }
  if (p < le && *p == '.') { p++; while (p < le && *p >= '0' && *p <= '9') { if (nd < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) nd++; e10--; } any = true; p++; } }
  if (!any) return false;
  bool exact = nd < 19;                                                    // 19 digits may have dropped further ones
  if (p < le && (*p == 'e' || *p == 'E')) {
    const char *pe = p + 1; bool en = false; if (pe < le && (*pe == '-' || *pe == '+')) { en = *pe == '-'; pe++; }
    if (pe < le && *pe >= '0' && *pe <= '9') { int ev = 0; while (pe < le && *pe >= '0' && *pe <= '9') { if (ev < 10000) ev = ev * 10 + (*pe - '0'); pe++; } e10 += en ? -ev : ev; p = pe; }
  }
  if (p < le && ((*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z'))) return false;   // inf / nan / 0x...: not ours
  if (!exact || mant > (1ull << 53) || e10 < -22 || e10 > 22) return false;
  double d = (double)mant; d = e10 < 0 ? d / P10[-e10] : d * P10[e10];
  uint64_t bits; std::memcpy(&bits, &d, 8);
  const uint32_t low = (uint32_t)(bits & 0x1fffffffu);                    // the 29 bits float drops; boundary at 0x10000000
  if (low - 0x0ffffffeu <= 4u) return false;                               // within 2 ulp(double) of the tie: let strtof decide
  const float f = (float)d;
  if (!(std::fabs(f) >= 1.17549435e-38f) && mant != 0) return false;       // subnormal floats round differently: strtof
  out = neg ? -f : f; q = p; return true;
}
bool read_obj(const std::string &path, ObjMesh &m, std::string &err, IngestScratch *scratch) {
  IngestScratch local; IngestScratch &S = scratch ? *scratch : local;
  std::vector<uint8_t> &d = S.file; if (!read_file(path, d)) { err = "cannot read " + path; return false; }
  m.pos.clear(); m.uv.clear(); m.nrm.clear(); m.idx_pos.clear(); m.idx_uv.clear(); m.idx_nrm.clear();      // capacity is kept
  const char *p = (const char *)d.data(), *e = p + d.size();
  bool has_uv = true, has_n = true; long nfaces = 0;
  std::vector<long> &fv = S.fv, &ft = S.ft, &fn = S.fn;
  { // one cheap pass over the line starts so that every array is allocated once (a 100 k-vertex frame grew ~25 MB of vectors by
    // doubling, on 64 ingest threads at a time)
    size_t nv = 0, nt = 0, nn = 0, nfl = 0;
    for (const char *q = p; q < e;) {
      const char *ln = q; const char *nl = (const char *)std::memchr(q, '\n', (size_t)(e - q)); q = nl ? nl + 1 : e;
      while (ln < q && (*ln == ' ' || *ln == '\t')) ln++;
      if (ln + 1 < q) { if (ln[0] == 'v') { if (ln[1] == ' ' || ln[1] == '\t') nv++; else if (ln[1] == 't') nt++; else if (ln[1] == 'n') nn++; } else if (ln[0] == 'f') nfl++; }
    }
    m.pos.reserve(3 * nv); m.uv.reserve(2 * nt); m.nrm.reserve(3 * nn); m.idx_pos.reserve(3 * nfl + 64); m.idx_uv.reserve(3 * nfl + 64); m.idx_nrm.reserve(3 * nfl + 64);
  }
  auto skip_sp = [&](const char *&q) { while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) q++; };
  auto num = [&](const char *&q, const char *le, float &o) {
    skip_sp(q);
    if (fast_float(q, le, o)) return true;
    // the unusual numbers go to strtof - on a NUL-terminated copy of the token, bounded by the line end: the file buffer is not
    // terminated, and strtof skips leading white space INCLUDING '\n' (a `v 1 2` line would take its third value from the next line)
    if (q >= le) return false;
    char tok[96]; size_t nt = 0;
    while (q + nt < le && nt + 1 < sizeof(tok) && q[nt] != ' ' && q[nt] != '\t' && q[nt] != '\r' && q[nt] != '\n') { tok[nt] = q[nt]; nt++; }
    tok[nt] = 0;
    char *end; o = std::strtof(tok, &end); const bool ok = end != tok; q += ok ? (size_t)(end - tok) : 0; return ok; };
  auto integer = [&](const char *&q, const char *le, long &o) {                 // optional sign + digits (what strtol accepts here)
    const char *s0 = q; bool neg = false; if (q < le && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    const char *d0 = q; long v = 0; while (q < le && *q >= '0' && *q <= '9') { if (v < (1l << 40)) v = v * 10 + (*q - '0'); q++; }
    if (q == d0) { q = s0; return false; }
    o = neg ? -v : v; return true; };
  while (p < e) {
    const char *ln = p; const char *nl = (const char *)std::memchr(p, '\n', (size_t)(e - p)); p = nl ? nl : e;
    const char *le = p; if (p < e) p++;
    const char *q = ln; skip_sp(q);
    if (q + 1 < le && q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) { q += 1; float x, y, z; if (num(q, le, x) && num(q, le, y) && num(q, le, z)) { m.pos.push_back(x); m.pos.push_back(y); m.pos.push_back(z); } }
    else if (q + 2 < le && q[0] == 'v' && q[1] == 't' && (q[2] == ' ' || q[2] == '\t')) { q += 2; float u = 0, v = 0; num(q, le, u); num(q, le, v); m.uv.push_back(u); m.uv.push_back(v); }
    else if (q + 2 < le && q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t')) { q += 2; float x = 0, y = 0, z = 0; num(q, le, x); num(q, le, y); num(q, le, z); m.nrm.push_back(x); m.nrm.push_back(y); m.nrm.push_back(z); }
    else if (q + 1 < le && q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
      q += 1; fv.clear(); ft.clear(); fn.clear();
      for (;;) {
        skip_sp(q); if (q >= le) break;
        long a; if (!integer(q, le, a)) break;
        long b = 0, c = 0; bool hb = false, hc = false;
        if (q < le && *q == '/') { q++; if (q < le && *q != '/') hb = integer(q, le, b); if (q < le && *q == '/') { q++; hc = integer(q, le, c); } }
        const long np = (long)m.pos.size() / 3, nt = (long)m.uv.size() / 2, nn = (long)m.nrm.size() / 3;
        fv.push_back(a < 0 ? np + a : a - 1);
        ft.push_back(hb ? (b < 0 ? nt + b : b - 1) : -1);
        fn.push_back(hc ? (c < 0 ? nn + c : c - 1) : -1);
      }
      for (size_t k = 1; k + 1 < fv.size(); k++) {
        const size_t tri[3] = { 0, k, k + 1 };
        for (size_t t : tri) {
          if (fv[t] < 0 || fv[t] >= (long)m.pos.size() / 3) { err = path + ": face references a missing vertex"; return false; }
          m.idx_pos.push_back((uint32_t)fv[t]);
          if (ft[t] < 0 || ft[t] >= (long)m.uv.size() / 2) has_uv = false;
          m.idx_uv.push_back(ft[t] < 0 ? 0u : (uint32_t)ft[t]);
          if (fn[t] < 0 || fn[t] >= (long)m.nrm.size() / 3) has_n = false;
          m.idx_nrm.push_back(fn[t] < 0 ? 0u : (uint32_t)fn[t]);
        }
        nfaces++;
      }
    }
  }
  if (!nfaces || m.pos.empty()) { err = path + ": no faces"; return false; }
  if (!has_uv || m.uv.empty()) { m.uv.clear(); m.idx_uv.clear(); }
  if (!has_n || m.nrm.empty()) { m.nrm.clear(); m.idx_nrm.clear(); }
  return true;
}

// zlib-stream inflate: libdeflate when the shared library is installed (about 3x zlib's rate on 2048^2 PNG data; looked up once
// with dlopen, no build-time dependency), else zlib's uncompress
static bool inflate_zlib_stream(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
  typedef void *(*alloc_fn)(void); typedef int (*dec_fn)(void *, const void *, size_t, void *, size_t, size_t *); typedef void (*free_fn)(void *);
  struct Lib { alloc_fn alloc = nullptr; dec_fn dec = nullptr; free_fn fre = nullptr; Lib() {
    if (const char *e = std::getenv("UVOL_NO_LIBDEFLATE")) if (*e == '1') return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL); if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL); if (!h) return;
    alloc = (alloc_fn)dlsym(h, "libdeflate_alloc_decompressor"); dec = (dec_fn)dlsym(h, "libdeflate_zlib_decompress"); fre = (free_fn)dlsym(h, "libdeflate_free_decompressor");
    if (!alloc || !dec || !fre) { alloc = nullptr; dec = nullptr; fre = nullptr; } } };
  static const Lib L;
  if (L.dec) {
    static thread_local void *dc = nullptr; if (!dc) dc = L.alloc();      // one decompressor per ingest thread (freed with the process)
    size_t got = 0;
    if (dc && L.dec(dc, in, in_len, out, out_len, &got) == 0 && got == out_len) return true;                 // LIBDEFLATE_SUCCESS = 0; on any failure zlib gets its say
  }
  uLongf outl = (uLongf)out_len;
  return uncompress(out, &outl, in, (uLong)in_len) == Z_OK && outl == out_len;
}

// ------------------------------------------------------------------ PNG (8/16-bit, colour types 0/2/3/4/6, non-interlaced) via zlib
bool read_png(const std::string &path, Image &img, std::string &err, IngestScratch *scratch) {
  IngestScratch local; IngestScratch &S = scratch ? *scratch : local;
  std::vector<uint8_t> &d = S.file; if (!read_file(path, d)) { err = "cannot read " + path; return false; }
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
  if (d.size() < 33 || std::memcmp(d.data(), sig, 8)) { err = path + ": not a PNG"; return false; }
  auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
  uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0; std::vector<uint8_t> &idat = S.idat, plte, trns; idat.clear();
  for (size_t o = 8; o + 12 <= d.size();) {
    uint32_t len = be32(o); if (o + 12 + len > d.size()) break;
    const char *t = (const char *)&d[o + 4]; const uint8_t *body = &d[o + 8];
    if (!std::memcmp(t, "IHDR", 4) && len >= 13) { w = be32(o + 8); h = be32(o + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
    else if (!std::memcmp(t, "PLTE", 4)) plte.assign(body, body + len);
    else if (!std::memcmp(t, "tRNS", 4)) trns.assign(body, body + len);
    else if (!std::memcmp(t, "IDAT", 4)) { if (idat.empty()) idat.reserve(d.size()); idat.insert(idat.end(), body, body + len); }
    else if (!std::memcmp(t, "IEND", 4)) break;
    o += 12 + len;
  }
  if (w > 16384 || h > 16384) { err = path + ": image larger than 16384 x 16384"; return false; }      // the encoder's own limit; also bounds the allocations below
  if (!w || !h || interlace || (depth != 8 && depth != 16) || (ctype != 0 && ctype != 2 && ctype != 3 && ctype != 4 && ctype != 6) || (ctype == 3 && depth != 8)) { err = path + ": unsupported PNG variant"; return false; }
  const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4, bpp = ch * depth / 8;
  const size_t stride = (size_t)w * bpp;
  std::vector<uint8_t> &raw = S.raw; raw.resize((stride + 1) * h);
  if (!inflate_zlib_stream(idat.data(), idat.size(), raw.data(), raw.size())) { err = path + ": zlib inflate failed"; return false; }
  // rows are un-filtered IN PLACE in the inflated buffer (the previous row is the one just done), one tight loop per filter type
  // instead of a switch per byte
  std::vector<uint8_t> &zero = S.zero; zero.assign(stride, 0);
  img.w = w; img.h = h; img.rgba.resize((size_t)w * h * 4);
  for (uint32_t y = 0; y < h; y++) {
    uint8_t *r = &raw[(stride + 1) * y]; const int ft = r[0]; r++;
    const uint8_t *prev = y ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
    const size_t B = (size_t)bpp;
    switch (ft) {
      case 1: for (size_t x = B; x < stride; x++) r[x] = (uint8_t)(r[x] + r[x - B]); break;
      case 2: for (size_t x = 0; x < stride; x++) r[x] = (uint8_t)(r[x] + prev[x]); break;
      case 3: for (size_t x = 0; x < B && x < stride; x++) r[x] = (uint8_t)(r[x] + prev[x] / 2);
              for (size_t x = B; x < stride; x++) r[x] = (uint8_t)(r[x] + (r[x - B] + prev[x]) / 2);
              break;
      case 4: for (size_t x = 0; x < B && x < stride; x++) r[x] = (uint8_t)(r[x] + prev[x]);                    // a = c = 0: the predictor is b
              for (size_t x = B; x < stride; x++) {                        // pa = |b - c|, pb = |a - c|, pc = |a + b - 2c|: selects, no branches
                const int a = r[x - B], b2 = prev[x], c = prev[x - B], pa = std::abs(b2 - c), pb = std::abs(a - c), pc = std::abs(a + b2 - 2 * c);
                const int ab = pa <= pb ? a : b2, pab = pa <= pb ? pa : pb;
                r[x] = (uint8_t)(r[x] + (pab <= pc ? ab : c));
              }
              break;
      default: break;
    }
    const uint8_t *cur = r;
    uint8_t *o = &img.rgba[(size_t)y * w * 4]; const int s = depth / 8;
    if (ctype == 6 && depth == 8) { std::memcpy(o, cur, (size_t)w * 4); continue; }
    if (ctype == 2 && depth == 8) { for (uint32_t x = 0; x < w; x++) { o[4 * x] = cur[3 * x]; o[4 * x + 1] = cur[3 * x + 1]; o[4 * x + 2] = cur[3 * x + 2]; o[4 * x + 3] = 255; } continue; }
    for (uint32_t x = 0; x < w; x++) {
      const uint8_t *px = &cur[(size_t)x * bpp];
      o[4 * x + 3] = 255;
      switch (ctype) {
        case 0: o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = px[0]; break;
        case 2: o[4 * x] = px[0]; o[4 * x + 1] = px[s]; o[4 * x + 2] = px[2 * s]; break;
        case 3: { const size_t k = px[0]; if (3 * k + 2 < plte.size()) { o[4 * x] = plte[3 * k]; o[4 * x + 1] = plte[3 * k + 1]; o[4 * x + 2] = plte[3 * k + 2]; } if (k < trns.size()) o[4 * x + 3] = trns[k]; break; }
        case 4: o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = px[0]; o[4 * x + 3] = px[s]; break;
        default: o[4 * x] = px[0]; o[4 * x + 1] = px[s]; o[4 * x + 2] = px[2 * s]; o[4 * x + 3] = px[3 * s]; break;
      }
    }
  }
  return true;
}

int read_png_raw(const std::string &path, PngRaw &out, std::string &err, IngestScratch *scratch, bool keep_deflated) {
  IngestScratch local; IngestScratch &S = scratch ? *scratch : local;
  std::vector<uint8_t> &d = S.file; if (!read_file(path, d)) { err = "cannot read " + path; return -1; }
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
  if (d.size() < 33 || std::memcmp(d.data(), sig, 8)) { err = path + ": not a PNG"; return -1; }
  auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
  uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0; std::vector<uint8_t> &idat = S.idat; idat.clear();
  for (size_t o = 8; o + 12 <= d.size();) {
    uint32_t len = be32(o); if (o + 12 + len > d.size()) break;
    const char *t = (const char *)&d[o + 4]; const uint8_t *body = &d[o + 8];
    if (!std::memcmp(t, "IHDR", 4) && len >= 13) { w = be32(o + 8); h = be32(o + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
    else if (!std::memcmp(t, "IDAT", 4)) { if (idat.empty()) idat.reserve(d.size()); idat.insert(idat.end(), body, body + len); }
    else if (!std::memcmp(t, "IEND", 4)) break;
    o += 12 + len;
  }
  if (!w || !h || interlace || depth != 8 || (ctype != 2 && ctype != 6) || w > 8192 || h > 16384) return 0;        // read_png decides (and words the errors)
  out.w = w; out.h = h; out.ch = ctype == 2 ? 3 : 4;
  if (keep_deflated) { if (idat.size() < 6) { err = path + ": no image data"; return -1; } out.raw.assign(idat.begin(), idat.end()); return 1; }      // (uvol_inflate_png_batch_dev inflates it)
  out.raw.resize(((size_t)w * out.ch + 1) * h);
  if (!inflate_zlib_stream(idat.data(), idat.size(), out.raw.data(), out.raw.size())) { err = path + ": zlib inflate failed"; return -1; }
  return 1;
}

// ------------------------------------------------------------------ frame accounting (scripts/Encoder.py:103-154)
static void split_path(const std::string &p, std::string &dir, std::string &base) { size_t k = p.find_last_of('/'); if (k == std::string::npos) { dir = ""; base = p; } else { dir = p.substr(0, k); base = p.substr(k + 1); } }
bool check_total_frames(const std::string &drc_pat, const std::string &ktx2_pat, int batch, double geo_rate, double tex_rate, FrameCounts &out, std::string &err) {
  std::string dd, dp, kd, kp; split_path(drc_pat, dd, dp); split_path(ktx2_pat, kd, kp);
  out = FrameCounts();
  for (auto &f : list_dir(dd)) if (match_pattern_lenient(dp, f)) out.geometry_frames++;
  std::vector<std::string> segs; for (auto &f : list_dir(kd)) if (match_pattern_lenient(kp, f)) segs.push_back(f);
  if (segs.empty()) { err = "no texture segments match " + ktx2_pat; return false; }
  out.texture_segments = (long)segs.size();
  std::vector<uint8_t> last; if (!read_file((kd.empty() ? "" : kd + "/") + segs.back(), last) || last.size() < 36) { err = "cannot read last texture segment"; return false; }
  uint32_t layers; std::memcpy(&layers, &last[32], 4);                 // KTX2 layerCount (scripts/Encoder.py:128-130)
  out.texture_frames = (out.texture_segments - 1) * batch + (long)layers;
  out.compatible = (double)out.geometry_frames * tex_rate == (double)out.texture_frames * geo_rate;     // :135-137
  out.geometry_duration = out.geometry_frames / geo_rate; out.texture_duration = out.texture_frames / tex_rate;
  return true;
}

// ------------------------------------------------------------------ sharding
ShardPlan shard_plan(long n_frames, int batch, int world, int rank) {
  ShardPlan P; if (batch <= 0 || world <= 0 || rank < 0 || rank >= world || n_frames <= 0) return P;
  const long n_seg = (n_frames + batch - 1) / batch, lo = n_seg * rank / world, hi = n_seg * (rank + 1) / world;
  const long f_lo = lo * batch, f_hi = std::min(hi * batch, n_frames);
  P.first_frame = f_lo; P.n_frames = std::max(0L, f_hi - f_lo); P.first_segment = lo; P.n_segments = hi - lo;
  return P;
}

// ------------------------------------------------------------------ audio duration
bool audio_duration(const std::string &path, double &seconds, std::string &err) {
  std::vector<uint8_t> d;
  if (path.compare(0, 4, "http") == 0 || !read_file(path, d) || d.size() < 16) { err = "cannot read " + path; return false; }
  auto le32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
  auto le16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
  if (!std::memcmp(d.data(), "RIFF", 4) && !std::memcmp(d.data() + 8, "WAVE", 4)) {
    uint32_t byte_rate = 0; size_t o = 12;
    while (o + 8 <= d.size()) {
      const uint32_t len = le32(o + 4);
      if (!std::memcmp(&d[o], "fmt ", 4) && o + 8 + 16 <= d.size()) byte_rate = le32(o + 16);
      else if (!std::memcmp(&d[o], "data", 4)) { if (!byte_rate) break; seconds = (double)std::min<size_t>(len, d.size() - o - 8) / byte_rate; return true; }
      (void)le16; o += 8 + (size_t)len + (len & 1);
    }
    err = path + ": malformed WAV"; return false;
  }
  // MPEG audio: skip an ID3v2 tag, then add up the samples of every frame header found back to back
  size_t o = 0;
  if (!std::memcmp(d.data(), "ID3", 3) && d.size() > 10) o = 10 + (((size_t)d[6] & 127) << 21 | ((size_t)d[7] & 127) << 14 | ((size_t)d[8] & 127) << 7 | ((size_t)d[9] & 127));
  static const int br1[16] = { 0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 0 }, br2[16] = { 0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 0 };
  static const int sr1[4] = { 44100, 48000, 32000, 0 };
  double total = 0; long frames = 0;
  while (o + 4 <= d.size()) {
    if (d[o] != 0xFF || (d[o + 1] & 0xE0) != 0xE0) { if (frames) break; o++; continue; }
    const int ver = (d[o + 1] >> 3) & 3, layer = (d[o + 1] >> 1) & 3, bri = d[o + 2] >> 4, sri = (d[o + 2] >> 2) & 3, padb = (d[o + 2] >> 1) & 1;
    if (ver == 1 || layer != 1 || bri == 0 || bri == 15 || sri == 3) { if (frames) break; o++; continue; }      // Layer III only
    const int sr = sr1[sri] >> (ver == 3 ? 0 : (ver == 2 ? 1 : 2)), br = (ver == 3 ? br1[bri] : br2[bri]) * 1000, spf = ver == 3 ? 1152 : 576;
    const size_t flen = (size_t)(spf / 8 * br / sr + padb);
    if (flen < 4) break;
    total += (double)spf / sr; frames++; o += flen;
  }
  if (!frames) { err = path + ": neither WAV nor MPEG Layer III audio"; return false; }
  seconds = total; return true;
}

// ------------------------------------------------------------------ manifests
Json manifest_player(const Config &c, long geo_frames, long tex_segments, uint32_t tw, uint32_t th, int pad, long etc2_frames) {
  const std::string hashes = "[" + std::string((size_t)pad, '#') + "]";
  Json m = Json::object();
  m.set("version", Json::string("v2"));
  if (!c.audio_url.empty()) { Json a = Json::object(); a.set("path", Json::string(c.audio_url)); a.set("format", Json::string("mp3")); m.set("audio", a); }
  Json g = Json::object(), gt = Json::object(), gd = Json::object();
  gd.set("format", Json::string("draco")); gd.set("frameRate", Json::number(c.geometry_frame_rate)); gd.set("frameCount", Json::number((double)geo_frames, true));
  gt.set("draco", gd); g.set("targets", gt); g.set("path", Json::string("geometry_[target]/" + hashes + "[ext]")); m.set("geometry", g);
  Json t = Json::object(), tt = Json::object(), td = Json::object(), res = Json::array();
  res.arr.push_back(Json::number(tw, true)); res.arr.push_back(Json::number(th, true));
  td.set("format", Json::string("ktx2")); td.set("resolution", res); td.set("type", Json::string("baseColor")); td.set("tag", Json::string("default"));
  td.set("sequenceSize", Json::number(c.ktx2_batch_size, true)); td.set("sequenceCount", Json::number((double)tex_segments, true)); td.set("frameRate", Json::number(c.texture_frame_rate));
  tt.set("ktx2", td);
  if (etc2_frames > 0) {                    // raw ETC2 RGB (ETC1 subset) blocks, one frame per file: sequenceSize 1 (the player indexes segments, src/V2/player.ts:418-446)
    Json te = Json::object(), res2 = Json::array();
    res2.arr.push_back(Json::number(tw, true)); res2.arr.push_back(Json::number(th, true));
    te.set("format", Json::string("etc2")); te.set("resolution", res2); te.set("type", Json::string("baseColor")); te.set("tag", Json::string("default"));
    te.set("sequenceSize", Json::number(1, true)); te.set("sequenceCount", Json::number((double)etc2_frames, true)); te.set("frameRate", Json::number(c.texture_frame_rate));
    tt.set("etc2", te);
  }
  t.set("targets", tt); t.set("path", Json::string("texture_[target]_[type]_[tag]/" + hashes + "[ext]")); m.set("texture", t);
  return m;
}
Json manifest_encoder_py(const Config &c, long geo_frames, long tex_segments, const std::string &drc_rel, const std::string &ktx2_rel) {   // scripts/Encoder.py:311-328
  Json m = Json::object(); m.set("version", Json::string("v2"));
  Json g = Json::object(); g.set("format", Json::string("draco")); g.set("frameRate", Json::number(c.geometry_frame_rate)); g.set("frameCount", Json::number((double)geo_frames, true)); g.set("path", Json::string(drc_rel)); m.set("geometry", g);
  Json t = Json::object(), arr = Json::array(), e = Json::object();
  e.set("format", Json::string("ktx2")); e.set("frameRate", Json::number(c.texture_frame_rate)); e.set("sequenceCount", Json::number((double)tex_segments, true)); e.set("sequenceSize", Json::number(c.ktx2_batch_size, true)); e.set("path", Json::string(ktx2_rel));
  arr.arr.push_back(e); t.set("targets", arr); m.set("texture", t);
  if (!c.audio_url.empty()) { Json a = Json::object(); a.set("format", Json::string("mp3")); a.set("path", Json::string(c.audio_url)); m.set("audio", a); }
  return m;
}

}  // namespace uvolh

// ------------------------------------------------------------------ tiny C surface for the CPU-side tests (no HIP)
extern "C" {
static thread_local std::string g_ret;
const char *uvolh_convert_pounds(const char *s) { g_ret = uvolh::convert_pounds_to_c_style(s); return g_ret.c_str(); }
int uvolh_match_pattern(const char *p, const char *f) { return uvolh::match_pattern(p, f) ? 1 : 0; }
int uvolh_match_pattern_lenient(const char *p, const char *f) { return uvolh::match_pattern_lenient(p, f) ? 1 : 0; }
const char *uvolh_format_index(const char *p, unsigned i) { g_ret = uvolh::format_index(p, i); return g_ret.c_str(); }
const char *uvolh_check_config(const char *json_text) {
  uvolh::Json j; std::string err;
  if (!uvolh::json_parse(json_text, j, err)) { g_ret = "parse error: " + err; return g_ret.c_str(); }
  g_ret = uvolh::check_all_fields(j); return g_ret.c_str();
}
const char *uvolh_check_total_frames(const char *drc_pat, const char *ktx2_pat, int batch, double gr, double tr) {
  uvolh::FrameCounts fc; std::string err;
  if (!uvolh::check_total_frames(drc_pat, ktx2_pat, batch, gr, tr, fc, err)) { g_ret = "error: " + err; return g_ret.c_str(); }
  char b[256]; std::snprintf(b, sizeof b, "{\"geometry\": %.17g, \"texture\": %.17g, \"geometry_frames\": %ld, \"texture_frames\": %ld, \"segments\": %ld, \"compatible\": %s}",
                             fc.geometry_duration, fc.texture_duration, fc.geometry_frames, fc.texture_frames, fc.texture_segments, fc.compatible ? "true" : "false");
  g_ret = b; return g_ret.c_str();
}
const char *uvolh_manifest(const char *config_json, long geo_frames, long segments, unsigned w, unsigned h, int pad, int encoder_py_shape, const char *drc_rel, const char *ktx2_rel) {
  uvolh::Config c; std::string err;
  if (!uvolh::load_config(config_json, c, err)) { g_ret = "error: " + err; return g_ret.c_str(); }
  g_ret = uvolh::json_dump(encoder_py_shape ? uvolh::manifest_encoder_py(c, geo_frames, segments, drc_rel, ktx2_rel) : uvolh::manifest_player(c, geo_frames, segments, w, h, pad));
  return g_ret.c_str();
}
const char *uvolh_manifest_targets(const char *config_json, long geo_frames, long segments, unsigned w, unsigned h, int pad, long etc2_frames) {
  uvolh::Config c; std::string err;
  if (!uvolh::load_config(config_json, c, err)) { g_ret = "error: " + err; return g_ret.c_str(); }
  g_ret = uvolh::json_dump(uvolh::manifest_player(c, geo_frames, segments, w, h, pad, etc2_frames));
  return g_ret.c_str();
}
int uvolh_shard_plan(long n_frames, int batch, int world, int rank, long *out4) {
  const uvolh::ShardPlan P = uvolh::shard_plan(n_frames, batch, world, rank);
  out4[0] = P.first_frame; out4[1] = P.n_frames; out4[2] = P.first_segment; out4[3] = P.n_segments; return 0;
}
double uvolh_audio_duration(const char *path) { double s = 0; std::string err; return uvolh::audio_duration(path, s, err) ? s : -1.0; }
const char *uvolh_template(void) { g_ret = uvolh::config_template(); return g_ret.c_str(); }
int uvolh_read_obj_counts(const char *path, unsigned *out6) {
  uvolh::ObjMesh m; std::string err; if (!uvolh::read_obj(path, m, err)) return -1;
  out6[0] = (unsigned)m.pos.size() / 3; out6[1] = (unsigned)m.uv.size() / 2; out6[2] = (unsigned)m.nrm.size() / 3; out6[3] = (unsigned)m.idx_pos.size() / 3; out6[4] = (unsigned)m.idx_uv.size() / 3; out6[5] = (unsigned)m.idx_nrm.size() / 3; return 0;
}
// test hook: the parsed position values (read_obj's number parser must agree with strtof bit for bit)
int uvolh_read_obj_positions(const char *path, float *pos, size_t cap_floats) {
  uvolh::ObjMesh m; std::string err; if (!uvolh::read_obj(path, m, err)) return -1;
  if (m.pos.size() > cap_floats) return -2;
  std::memcpy(pos, m.pos.data(), m.pos.size() * sizeof(float)); return (int)(m.pos.size() / 3);
}
// test hook: every array read_obj produces (the device parser of csrc/obj_ingest.hip must agree with it bit for bit); counts6 = {n_pos, n_uv,
// n_nrm, faces, faces with uv indices, faces with normal indices}; buffers may be NULL (counts only)
int uvolh_read_obj_arrays(const char *path, float *pos, float *uv, float *nrm, unsigned *ip, unsigned *iu, unsigned *in_, unsigned *counts6) {
  uvolh::ObjMesh m; std::string err; if (!uvolh::read_obj(path, m, err)) return -1;
  counts6[0] = (unsigned)m.pos.size() / 3; counts6[1] = (unsigned)m.uv.size() / 2; counts6[2] = (unsigned)m.nrm.size() / 3;
  counts6[3] = (unsigned)m.idx_pos.size() / 3; counts6[4] = (unsigned)m.idx_uv.size() / 3; counts6[5] = (unsigned)m.idx_nrm.size() / 3;
  auto put = [](void *d, const void *s_, size_t n) { if (d) std::memcpy(d, s_, n); };
  put(pos, m.pos.data(), m.pos.size() * 4); put(uv, m.uv.data(), m.uv.size() * 4); put(nrm, m.nrm.data(), m.nrm.size() * 4);
  put(ip, m.idx_pos.data(), m.idx_pos.size() * 4); put(iu, m.idx_uv.data(), m.idx_uv.size() * 4); put(in_, m.idx_nrm.data(), m.idx_nrm.size() * 4);
  return 0;
}
int uvolh_read_png(const char *path, unsigned *wh, unsigned char *rgba, size_t cap) {
  uvolh::Image im; std::string err; if (!uvolh::read_png(path, im, err)) return -1;
  wh[0] = im.w; wh[1] = im.h; if (rgba && cap >= im.rgba.size()) std::memcpy(rgba, im.rgba.data(), im.rgba.size()); return 0;
}
}
