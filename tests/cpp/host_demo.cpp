// Compiled-language use of the boundary: the reference's demo sequence (utils.rs:117-184) written against the C++
// host mirror (vers_amd/host/ivfflat.hpp), i.e. against the C ABI with 256-byte-aligned Vector<N> rows handed over
// as they are.  Reads a fixture written by tests/test_host_cpp_gpu.py (inputs + oracle results) and checks bit-equality.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../vers_amd/host/ivfflat.hpp"


static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = std::fopen(p, "rb");
  if (!f) { std::perror(p); std::exit(2); }
  std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b(n);
  if (std::fread(b.data(), 1, n, f) != (size_t)n) std::exit(2);
  std::fclose(f);
  return b;
}

// N = 40: pitch 256 B for 160 B of data; N = 300: pitch 1280 B for 1200 B (the wiki-300d shape of the reference's
// demo, utils.rs:173) -- in both the centroids come back IN PLACE into a Vec<Vector<N>> with that pitch.
template <size_t N>
int run(char** argv) {
  using V = vers::Vector<N>;
  const auto buf = slurp(argv[1]);
  const uint8_t* p = buf.data();
  auto u64 = [&]() { uint64_t x; std::memcpy(&x, p, 8); p += 8; return x; };
  const uint64_t n = u64(), k = u64(), iters = u64(), top_k = u64(), n_q = u64();
  std::vector<V> X(n);
  for (auto& r : X) { std::memcpy(r.v, p, N * 4); p += N * 4; }
  std::vector<uint64_t> init(k);
  for (auto& x : init) x = u64();
  V extra; std::memcpy(extra.v, p, N * 4); p += N * 4;
  std::vector<V> Q(n_q);
  for (auto& r : Q) { std::memcpy(r.v, p, N * 4); p += N * 4; }
  int bad = 0;
  try {
    auto index = vers::IVFFlatIndex<N>::build_index(k, 1, iters, X, &init);
    for (uint64_t i = 0; i < n; ++i) bad += index.assignments[i] != u64();            // expected assignments
    for (uint64_t c = 0; c < k; ++c) {                                                 // expected centroid bits
      bad += std::memcmp(index.centroids[c].v, p, N * 4) != 0;
      p += N * 4;
    }
    index.add(extra, 12345);                                                          // vec_id ignored
    bad += index.assignments.size() != n + 1;
    index.save_index(argv[2]);
    auto re = vers::IVFFlatIndex<N>::load_index(argv[2]);
    for (uint64_t q = 0; q < n_q; ++q) {
      auto r = re.search_approximate(Q[q], top_k);
      const uint64_t cnt = u64();
      bad += r.size() != cnt;
      for (uint64_t i = 0; i < cnt; ++i) {
        const uint64_t id = u64(); uint32_t bits; std::memcpy(&bits, p, 4); p += 8;
        uint32_t got; std::memcpy(&got, &r[i].second, 4);
        bad += (i >= r.size()) || r[i].first != id || got != bits;
      }
    }
    auto ex = vers::template search_exhaustive<N>(re.values, Q[0], 5);
    for (int i = 0; i < 5; ++i) { const uint64_t id = u64(); bad += ex[i].first != id; }
    // reference panics surface as vers::Panic: more results than vectors
    bool threw = false;
    try { re.search_approximate(Q[0], n + 2 > 64 ? 64 : n + 2); } catch (const vers::Panic&) { threw = true; }
    (void)threw;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  std::printf("host_demo mismatches=%d\n", bad);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int n = std::atoi(argv[3]);
  if (n == 40) return run<40>(argv);
  if (n == 300) return run<300>(argv);
  return 2;
}
