// ExpandedMap.h -- read view of the search state with the reference's ExpandedMap interface
// (ProjectToolkit/include/ExpandedMap.h:25-73, impl/ExpandedMap_impl.h).
//
// The reference stores (g, rhs[, info]) per expanded element in hash maps.  Here the field lives
// densely in HBM; this view pages it to the host lazily in 64x64 blocks (ufm_read_field) the
// first time a block is read after a step(), so a consumer that walks a path (the reference's
// LinearInterpolationPathExtractor reads get_interp_rhs / get_rhs along ~20 steps) moves a few
// tens of KB, not the whole field.  Out-of-range and unreached elements read +inf, as in the
// reference (impl:54-85).  The engine is at a fixed point of the update operator after every
// step, so RHS == G for every element it finalised.
#ifndef UFM_EXPANDED_MAP_H
#define UFM_EXPANDED_MAP_H

#include <cmath>
#include <cstdint>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "GridTypes.h"
#include "Macros.h"
#include "ufm.h"

template <typename ElemType_, typename InfoType_>
class ExpandedMap {
 public:
  using ElemType = ElemType_;
  using InfoType = InfoType_;
  // (g, rhs) for the level-0 planners, (g, rhs, info) for the others -- as the reference's
  // ExpandedMap.h:27-29, so that std::get<2>(element.second) of a driver compiles
  using value_ = typename std::conditional<std::is_void<InfoType_>::value, std::tuple<float, float>,
                                           std::tuple<float, float, typename std::conditional<std::is_void<InfoType_>::value, int, InfoType_>::type>>::type;
  using bucket_ = std::vector<std::pair<const ElemType_, value_>>;

  /** Filled by size(): every element that holds a finite value, as (elem, (g, rhs[, info])) pairs in
   * one bucket per 256x256 block -- what the reference's `for (auto b : map.buckets)` dump iterates.
   * info (level-1/2 planners): the engine's stored back-pointer(s), ufm_read_info. */
  std::vector<bucket_> buckets;

  ExpandedMap() = default;

  float get_g(const ElemType &s) const { return value(s.x, s.y); }
  float get_rhs(const ElemType &s) const { return value(s.x, s.y); }
  std::pair<float, float> get_g_rhs(const ElemType &s) const { const float v = value(s.x, s.y); return {v, v}; }
  bool consistent(const ElemType &s) const { const auto gr = get_g_rhs(s); return gr.first == gr.second; }

  /** impl:87-101: node planners return RHS(node); the cell planner averages the four cells
   * around the node. */
  float get_interp_rhs(const Node &s) const {
    if constexpr (std::is_same<ElemType, Cell>::value) {
      const Cell p(static_cast<int>(std::floor(s.x - 0.5)), static_cast<int>(std::floor(s.y - 0.5)));
      return (get_rhs(p.bottom_cell()) + get_rhs(p) + get_rhs(p.bottom_right_cell()) + get_rhs(p.right_cell())) * 0.25f;
    } else {
      return get_rhs(s);
    }
  }

  /** impl:113-118.  Materialises `buckets` (reads the whole field once). */
  size_t size() const {
    auto *self = const_cast<ExpandedMap *>(this);
    self->buckets.clear();
    if (!handle_) return 0;
    std::vector<float> g(static_cast<size_t>(nx_) * ny_);
    if (ufm_read_field(handle_, 0, 0, nx_, ny_, g.data(), nullptr) != UFM_OK) return 0;
    const int bx = (nx_ >> 8) + 1, by = (ny_ >> 8) + 1;
    self->buckets.resize(static_cast<size_t>(bx) * by);
    std::vector<int32_t> info;
    if constexpr (!std::is_void<InfoType_>::value) {
      info.resize(static_cast<size_t>(nx_) * ny_ * 2);
      if (ufm_read_info(handle_, 0, 0, nx_, ny_, info.data()) != UFM_OK) return 0;
    }
    size_t n = 0;
    for (int x = 0; x < nx_; ++x)
      for (int y = 0; y < ny_; ++y) {
        const size_t e = static_cast<size_t>(x) * ny_ + y;
        const float v = g[e];
        if (v < INFINITY) {
          auto &bucket = self->buckets[static_cast<size_t>(x >> 8) * by + (y >> 8)];
          if constexpr (std::is_void<InfoType_>::value) bucket.emplace_back(ElemType(x, y), value_(v, v));
          else bucket.emplace_back(ElemType(x, y), value_(v, v, make_info(info[2 * e], info[2 * e + 1])));
          ++n;
        }
      }
    return n;
  }

  // ---- wiring (used by ReplannerBase) ----
  void attach(ufm_t *h) { handle_ = h; }
  ufm_t *native_handle() const { return handle_; }
  void set_dims(int nx, int ny) { nx_ = nx; ny_ = ny; invalidate(); }
  void invalidate() const { cache_.clear(); }
  void clear() noexcept { invalidate(); buckets.clear(); }

 private:
  // linear element index -> Node / Cell (negative: "none", default-constructed like the reference's Node{} / Cell{})
  ElemType elem_of(int32_t idx) const { return idx >= 0 ? ElemType(idx / ny_, idx % ny_) : ElemType(); }
  template <typename U = InfoType_>
  typename std::conditional<std::is_void<U>::value, int, U>::type make_info(int32_t a, int32_t b) const {
    if constexpr (std::is_same<U, ElemType>::value) { (void)b; return elem_of(a); }
    else if constexpr (std::is_same<U, std::pair<ElemType, ElemType>>::value) return {elem_of(a), elem_of(b)};
    else { (void)a; (void)b; return 0; }
  }
  static constexpr int kBlock = 64;
  float value(int x, int y) const {
    if (!handle_ || x < 0 || y < 0 || x >= nx_ || y >= ny_) return INFINITY;
    const int bx = x / kBlock, by = y / kBlock;
    const uint64_t key = (static_cast<uint64_t>(bx) << 32) | static_cast<uint32_t>(by);
    auto it = cache_.find(key);
    if (it == cache_.end()) {
      const int x0 = bx * kBlock, y0 = by * kBlock;
      const int w = std::min(kBlock, nx_ - x0), h = std::min(kBlock, ny_ - y0);
      std::vector<float> blk(static_cast<size_t>(kBlock) * kBlock, INFINITY);
      std::vector<float> tmp(static_cast<size_t>(w) * h);
      if (ufm_read_field(handle_, x0, y0, w, h, tmp.data(), nullptr) == UFM_OK)
        for (int i = 0; i < w; ++i)
          for (int j = 0; j < h; ++j) blk[static_cast<size_t>(i) * kBlock + j] = tmp[static_cast<size_t>(i) * h + j];
      it = cache_.emplace(key, std::move(blk)).first;
    }
    return it->second[static_cast<size_t>(x - bx * kBlock) * kBlock + (y - by * kBlock)];
  }
  ufm_t *handle_ = nullptr;
  int nx_ = 0, ny_ = 0;
  mutable std::unordered_map<uint64_t, std::vector<float>> cache_;
};

#endif  // UFM_EXPANDED_MAP_H
