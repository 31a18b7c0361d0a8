// ivfflat.hpp -- C++ host-side mirror of vers's IVFFlatIndex<N> / Index<N> on top of the C ABI (include/vers_hip.h).
//
// The reference host is Rust (vers/src/indexes/{base,ivfflat}.rs); no Rust toolchain exists in this environment, so
// the compiled-language host side is written in C++ with the SAME names, argument meaning and error behaviour:
//   Vector<N>                      base.rs:14-17   (#[repr(align(256))] [f32; N])
//   IVFFlatIndex<N>::build_index   ivfflat.rs:102-136
//   Index::add / search_approximate ivfflat.rs:200-213 / 153-198
//   Index::save_index / load_index base.rs:31-58   (bincode 1.3 layout, see vers_amd/index.py)
// A reference panic becomes a thrown vers::Panic.  The five fields stay host-owned (serde); the handle is a cache.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vers_hip.h"

namespace vers {

struct Panic : std::runtime_error {
  int32_t status;
  Panic(int32_t s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void check(int32_t status) {
  if (status != VERS_OK) throw Panic(status, std::string("vers status ") + std::to_string(status) + ": " + vers_last_error());
}

template <size_t N>
struct alignas(256) Vector {  // base.rs:14-17
  float v[N];
};

template <size_t N>
class IVFFlatIndex {
 public:
  // the reference's fields, in its order (ivfflat.rs:9-15)
  size_t num_centroids = 0;
  std::vector<Vector<N>> values;
  std::vector<Vector<N>> centroids;
  std::vector<size_t> assignments;
  std::vector<std::vector<size_t>> ids;

  IVFFlatIndex() = default;
  IVFFlatIndex(const IVFFlatIndex&) = delete;
  IVFFlatIndex& operator=(const IVFFlatIndex&) = delete;
  IVFFlatIndex(IVFFlatIndex&& o) noexcept { *this = std::move(o); }
  IVFFlatIndex& operator=(IVFFlatIndex&& o) noexcept {
    if (h_) vers_ivf_destroy(h_);
    num_centroids = o.num_centroids; values = std::move(o.values); centroids = std::move(o.centroids);
    assignments = std::move(o.assignments); ids = std::move(o.ids); h_ = o.h_; device_ = o.device_; o.h_ = nullptr;
    return *this;
  }
  ~IVFFlatIndex() { if (h_) vers_ivf_destroy(h_); }

  // ivfflat.rs:102-136.  `init_indices` (optional) injects the draws of initialize_centroids (ivfflat.rs:18-27);
  // without it they are drawn like the reference does: k indices WITH replacement per attempt from an unseeded RNG.
  static IVFFlatIndex build_index(size_t num_clusters, size_t num_attempts, size_t max_iterations,
                                  const std::vector<Vector<N>>& vectors, const std::vector<uint64_t>* init_indices = nullptr,
                                  int device = 0) {
    IVFFlatIndex ix;
    ix.device_ = device;
    check(vers_ivf_create(device, (uint32_t)N, &ix.h_));
    std::vector<uint64_t> draws;
    if (init_indices) draws = *init_indices;
    else {
      std::random_device rd;
      std::mt19937_64 rng(rd());
      draws.resize(num_attempts * num_clusters);
      for (auto& x : draws) x = vectors.empty() ? 0 : rng() % vectors.size();
    }
    // centroids are written straight into the Vec<Vector<N>> with ITS pitch (sizeof(Vector<N>) = round_up(4N, 256))
    std::vector<Vector<N>> cent(num_clusters);
    std::vector<uint64_t> asg(vectors.size() ? vectors.size() : 1);
    float cost = 0; int32_t kept = 0;
    check(vers_ivf_build(ix.h_, vectors.empty() ? nullptr : vectors[0].v, vectors.size(), sizeof(Vector<N>), num_clusters,
                         num_attempts, max_iterations, draws.data(), cent.empty() ? nullptr : cent[0].v, sizeof(Vector<N>),
                         asg.data(), &cost, &kept, nullptr));
    ix.num_centroids = num_clusters;
    ix.values = vectors;  // ivfflat.rs:131 vectors.clone()
    if (kept) {
      ix.centroids = std::move(cent);
      ix.assignments.assign(asg.begin(), asg.begin() + vectors.size());
    }  // else: nothing beat +inf -> empty centroids / assignments (ivfflat.rs:109-110)
    ix.ids.assign(num_clusters, {});
    for (size_t vec_id = 0; vec_id < ix.assignments.size(); ++vec_id) ix.ids[ix.assignments[vec_id]].push_back(vec_id);  // :123-127
    return ix;
  }

  // Index::add (ivfflat.rs:200-213): vec_id is accepted and ignored, like the reference (:209).
  void add(const Vector<N>& embedding, size_t /*vec_id*/) {
    uint64_t c = 0, id = 0;
    check(vers_ivf_add(handle(), embedding.v, &c, &id));
    values.push_back(embedding);
    assignments.push_back((size_t)c);
    ids[(size_t)c].push_back((size_t)id);
  }

  // Index::search_approximate (ivfflat.rs:153-198)
  std::vector<std::pair<size_t, float>> search_approximate(const Vector<N>& query, size_t top_k) const {
    std::vector<uint64_t> oi(top_k ? top_k : 1);
    std::vector<float> od(top_k ? top_k : 1);
    uint32_t cnt = 0;
    check(vers_ivf_search(handle(), query.v, sizeof(Vector<N>), 1, (uint32_t)top_k, 0 /* reference semantics */, oi.data(),
                          od.data(), &cnt));
    std::vector<std::pair<size_t, float>> out;
    for (uint32_t i = 0; i < cnt; ++i) out.emplace_back((size_t)oi[i], od[i]);
    return out;
  }

  // Index::save_index (base.rs:31-43): bincode 1.3 default options -- LE, u64 lengths, fields in order, no tags
  void save_index(const std::string& path) const {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("save_index: cannot create " + path);
    auto u64 = [&](uint64_t x) { std::fwrite(&x, 8, 1, f); };
    u64(num_centroids);
    u64(values.size());
    for (auto& r : values) std::fwrite(r.v, sizeof(float), N, f);
    u64(centroids.size());
    for (auto& r : centroids) std::fwrite(r.v, sizeof(float), N, f);
    u64(assignments.size());
    for (size_t a : assignments) u64(a);
    u64(ids.size());
    for (auto& l : ids) { u64(l.size()); for (size_t x : l) u64(x); }
    std::fclose(f);
  }

  // Index::load_index (base.rs:45-58); the device cache is rebuilt on first use
  static IVFFlatIndex load_index(const std::string& path, int device = 0) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("load_index: cannot open " + path);
    auto u64 = [&]() { uint64_t x = 0; if (std::fread(&x, 8, 1, f) != 1) { std::fclose(f); throw std::runtime_error("Deserialization error: unexpected end of file"); } return x; };
    auto rows = [&](std::vector<Vector<N>>& m) { m.resize(u64()); for (auto& r : m) if (std::fread(r.v, sizeof(float), N, f) != N) { std::fclose(f); throw std::runtime_error("Deserialization error: unexpected end of file"); } };
    IVFFlatIndex ix;
    ix.device_ = device;
    ix.num_centroids = u64();
    rows(ix.values);
    rows(ix.centroids);
    ix.assignments.resize(u64());
    for (auto& a : ix.assignments) a = u64();
    ix.ids.resize(u64());
    for (auto& l : ix.ids) { l.resize(u64()); for (auto& x : l) x = u64(); }
    std::fclose(f);
    return ix;
  }

 private:
  mutable vers_ivf_t* h_ = nullptr;
  int device_ = 0;
  vers_ivf_t* handle() const {  // after load_index: upload the host fields once
    if (!h_) {
      check(vers_ivf_create(device_, (uint32_t)N, &h_));
      std::vector<uint64_t> a(assignments.begin(), assignments.end());
      check(vers_ivf_upload(h_, values.empty() ? nullptr : values[0].v, values.size(), sizeof(Vector<N>),
                            centroids.empty() ? nullptr : centroids[0].v, centroids.size(), sizeof(Vector<N>), a.data()));
    }
    return h_;
  }
};

// utils::search_exhaustive (utils.rs:68-82)
template <size_t N>
std::vector<std::pair<size_t, float>> search_exhaustive(const std::vector<Vector<N>>& data, const Vector<N>& query, size_t top_k,
                                                        int device = 0) {
  vers_flat_t* h = nullptr;
  check(vers_flat_create(device, (uint32_t)N, &h));
  std::vector<uint64_t> oi(top_k ? top_k : 1);
  std::vector<float> od(top_k ? top_k : 1);
  uint32_t cnt = 0;
  int32_t rc = vers_flat_upload(h, data.empty() ? nullptr : data[0].v, data.size(), sizeof(Vector<N>));
  if (rc == VERS_OK) rc = vers_flat_search(h, query.v, sizeof(Vector<N>), 1, (uint32_t)top_k, VERS_METRIC_L2SQ, oi.data(), od.data(), &cnt);
  vers_flat_destroy(h);
  check(rc);
  std::vector<std::pair<size_t, float>> out;
  for (uint32_t i = 0; i < cnt; ++i) out.emplace_back((size_t)oi[i], od[i]);
  return out;
}

}  // namespace vers
