// tools/ingest_bench.cpp <obj> <png> <threads> <files per thread>: scaling of the ingest stage's two parsers on the host (no GPU):
// every thread parses the same two files over and over with its own scratch / result objects; prints ms per file and files/s.
#include "uvol_host.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
  if (argc < 5) return 1;
  const int nt = std::atoi(argv[3]), reps = std::atoi(argv[4]);
  using clk = std::chrono::steady_clock;
  for (int what = 0; what < 2; what++) {
    std::vector<double> ms((size_t)nt, 0.0);
    const auto t0 = clk::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
      uvolh::IngestScratch S; uvolh::ObjMesh m; uvolh::Image im; std::string err;
      const auto a = clk::now();
      for (int r = 0; r < reps; r++) { if (what == 0) uvolh::read_obj(argv[1], m, err, &S); else uvolh::read_png(argv[2], im, err, &S); }
      ms[(size_t)t] = std::chrono::duration<double, std::milli>(clk::now() - a).count() / reps;
    });
    for (auto &x : th) x.join();
    const double wall = std::chrono::duration<double>(clk::now() - t0).count();
    double avg = 0; for (double v : ms) avg += v / nt;
    std::printf("%s threads %3d: %.1f ms per file per thread, %.0f files/s\n", what == 0 ? "obj" : "png", nt, avg, nt * reps / wall);
  }
}
