// Index file cross-check between the two host mirrors (no GPU: neither save_index nor load_index touches the device).
//   index_file_demo write <path>         the C++ mirror saves the 3-vector / 2-cluster index below (N = 2)
//   index_file_demo resave <in> <out>    the C++ mirror loads a file (written by the Python mirror) and saves it again
// tests/test_index_file.py holds the expected bytes, written out by hand from the bincode 1.3 rules.
#include <cstdio>
#include <cstring>
#include <string>

#include "../../vers_amd/host/ivfflat.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  try {
    if (mode == "write") {
      vers::IVFFlatIndex<2> ix;
      ix.num_centroids = 2;
      ix.values.resize(3);
      const float v[3][2] = {{1.0f, -2.0f}, {0.5f, 0.25f}, {-0.0f, 3.0f}};
      for (int i = 0; i < 3; ++i) std::memcpy(ix.values[i].v, v[i], 8);
      ix.centroids.resize(2);
      const float c[2][2] = {{0.75f, -0.875f}, {0.0f, 3.0f}};
      for (int i = 0; i < 2; ++i) std::memcpy(ix.centroids[i].v, c[i], 8);
      ix.assignments = {0, 0, 1};
      ix.ids = {{0, 1}, {2}};
      ix.save_index(argv[2]);
      return 0;
    }
    if (mode == "resave" && argc >= 4) {
      auto ix = vers::IVFFlatIndex<2>::load_index(argv[2]);
      if (ix.num_centroids != 2 || ix.values.size() != 3 || ix.ids.size() != 2 || ix.ids[0].size() != 2) return 4;
      ix.save_index(argv[3]);
      return 0;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 3;
  }
  return 2;
}
