// uvol_host.hpp — host-side mirror of the reference driver scripts/Encoder.py (the caller of the hot path).
// Same project-config.json fields, same validation rules and error behaviour, same frame accounting, and the
// V2 manifest the stock player (src/V2/player.ts) reads.  C++17, no HIP in this header.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace uvolh {

// ---- minimal JSON (commentjson-compatible input: // and /* */ comments, trailing commas) ----
struct Json {
  enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
  bool b = false; double num = 0; bool is_int = false; std::string str;
  std::vector<Json> arr; std::vector<std::pair<std::string, Json>> obj;
  const Json *get(const std::string &k) const { for (auto &kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
  Json &set(const std::string &k, Json v) { for (auto &kv : obj) if (kv.first == k) { kv.second = std::move(v); return kv.second; } obj.emplace_back(k, std::move(v)); type = Obj; return obj.back().second; }
  static Json object() { Json j; j.type = Obj; return j; }
  static Json array() { Json j; j.type = Arr; return j; }
  static Json string(const std::string &s) { Json j; j.type = Str; j.str = s; return j; }
  static Json number(double v, bool integer = false) { Json j; j.type = Num; j.num = v; j.is_int = integer; return j; }
  // python truthiness, as used by `config.get(k)` tests in Encoder.py
  bool truthy() const { switch (type) { case Null: return false; case Bool: return b; case Num: return num != 0; case Str: return !str.empty(); case Arr: return !arr.empty(); default: return !obj.empty(); } }
};
bool json_parse(const std::string &text, Json &out, std::string &err);
std::string json_dump(const Json &j);

// ---- patterns (scripts/Encoder.py:16-19, :87-100; SURVEY I2: bracketed and bare forms are both accepted) ----
std::string convert_pounds_to_c_style(const std::string &s);          // export_#####.png -> export_%05u.png
bool match_pattern(const std::string &pattern, const std::string &file_name);   // exact Encoder.py semantics
bool match_pattern_lenient(const std::string &pattern, const std::string &file_name);   // also frame_[#####].obj vs frame_00001.obj
std::string format_index(const std::string &pattern, unsigned index);  // replaces the (bracketed or bare) # run

// ---- config ----
struct Config {
  Json raw;
  std::string name, obj_files_path, draco_files_path, images_path, ktx2_files_path, output_directory, audio_url, abc_file_path;
  int q_position = 11, q_texture = 10, q_normal = 8, q_generic = 8, compression_level = 7;
  int ktx2_first_file = 0, ktx2_file_count = 0, ktx2_batch_size = 0;
  double geometry_frame_rate = 0, texture_frame_rate = 0;
};
// check_all_fields (scripts/Encoder.py:45-84): returns "" when valid, else the reference's message
std::string check_all_fields(const Json &cfg);
bool load_config(const std::string &json_text, Config &c, std::string &err);
std::string config_template();                                          // `create-template` (scripts/Encoder.py:163-186)

// ---- ingest ----
struct ObjMesh { std::vector<float> pos, uv, nrm; std::vector<uint32_t> idx_pos, idx_uv, idx_nrm; };
// Buffers an ingest worker keeps between files.  A 100 k-vertex OBJ + a 2048^2 PNG need ~70 MB of temporaries; allocated afresh
// per file they are ~70 MB of new pages to fault in per frame, and 128 ingest threads of one process faulting at once serialise
// on the address-space lock (a file took 300 ms instead of 43).  read_obj / read_png also REUSE the capacity of the ObjMesh /
// Image they are given, so a caller that recycles its batch objects allocates nothing in the steady state.
struct IngestScratch { std::vector<uint8_t> file, idat, raw, zero; std::vector<long> fv, ft, fn; };
bool read_obj(const std::string &path, ObjMesh &m, std::string &err, IngestScratch *scratch = nullptr);
struct Image { uint32_t w = 0, h = 0; std::vector<uint8_t> rgba; };
bool read_png(const std::string &path, Image &img, std::string &err, IngestScratch *scratch = nullptr);
// The device un-filters (uvol_unfilter_png_batch_dev): chunk parse + zlib inflate only.  raw = height rows of (filter-type byte + width * ch
// bytes).  1: done; 0: a variant the device path does not take (16-bit, palette, grey, wider than 8192: use read_png); -1: error (err set)
struct PngRaw { uint32_t w = 0, h = 0; int ch = 0; std::vector<uint8_t> raw; };      // raw: the inflated scanlines, or (keep_deflated) the zlib stream itself
int read_png_raw(const std::string &path, PngRaw &out, std::string &err, IngestScratch *scratch = nullptr, bool keep_deflated = false);
// CPUs this process may actually use: the cgroup v2 quota (cpu.max) when there is one, else the hardware thread count.  A container
// that sees 256 hardware threads under a 16-CPU quota gets SLOWER with more than ~16 runnable threads (measured: DESIGN section 6).
unsigned effective_cpus();
bool write_file(const std::string &path, const void *data, size_t n);
bool read_file(const std::string &path, std::vector<uint8_t> &data);
long file_size(const std::string &path);                                      // -1: cannot stat
bool read_file_into(const std::string &path, uint8_t *dst, size_t n);         // exactly n bytes (the size file_size gave), into caller memory (a page-locked slab)
std::vector<std::string> list_dir(const std::string &dir);
bool make_dirs(const std::string &dir);

// ---- frame accounting (scripts/Encoder.py:103-154) ----
struct FrameCounts { long geometry_frames = 0, texture_frames = 0, texture_segments = 0; double geometry_duration = 0, texture_duration = 0; bool compatible = false; };
bool check_total_frames(const std::string &drc_path_pattern, const std::string &ktx2_path_pattern, int batch, double geo_rate, double tex_rate, FrameCounts &out, std::string &err);

// ---- sharding (the C++ mirror of shard.py plan()) ----
// contiguous blocks of WHOLE texture segments per rank; the geometry frames with the same indices go to the same rank
struct ShardPlan { long first_frame = 0, n_frames = 0, first_segment = 0, n_segments = 0; };
ShardPlan shard_plan(long n_frames, int batch, int world, int rank);

// ---- audio (scripts/Encoder.py:331-347: duration of AudioURL against the geometry / texture durations) ----
// duration in seconds of a .wav (RIFF/PCM) or .mp3 (MPEG-1/2/2.5 Layer III frame scan, ID3v2 skipped) file; false when the file
// cannot be read or is neither (remote URLs, other containers): the caller then skips the check like a missing AudioURL
bool audio_duration(const std::string &path, double &seconds, std::string &err);

// ---- manifests ----
// The shape src/V2/player.ts reads (src/Interfaces.ts:75-132): targets are objects keyed by target name.
// extra_etc2_frames > 0 adds the raw `etc2` texture target next to `ktx2` (src/Interfaces.ts:19, :60-73; src/V2/player.ts:208-222 picks
// the supported target of highest TEXTURE_FORMAT_PRIORITY, :338-356 loads one raw ETC2 RGB image per .etc2 file): one file per frame
Json manifest_player(const Config &c, long geometry_frames, long texture_segments, uint32_t tex_w, uint32_t tex_h, int pad_width, long extra_etc2_frames = 0);
// The literal dict scripts/Encoder.py:311-328 writes (kept behind --encoder-py-manifest).
Json manifest_encoder_py(const Config &c, long geometry_frames, long texture_segments, const std::string &drc_rel, const std::string &ktx2_rel);

}  // namespace uvolh
