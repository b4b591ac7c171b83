// A driver written against the reference's planner surface, following the call sequence of
// Tests/Planners/FDSTAR/main.cpp:77-166 (without the FIFOs): construct, configure, plan, then a
// few {patch_map, set_heuristic_multiplier, step, read the field around the start, set_start}
// rounds, finally the `tof` dump over map.buckets.  It prints what it reads; the pytest that
// builds and runs it compares the numbers with the Python binding / the oracle.
//
// usage: drop_in_driver <FD|SG|DFM> <size> <seed-dependent map file (raw uint8 size*size)> <n_patches> <patch file>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <cmath>
#include <string>
#include <type_traits>
#include <vector>

#define NO_HEURISTIC
#include "DynamicFastMarching.h"
#include "FieldDPlanner.h"
#include "ShiftedGridPlanner.h"

template <typename Planner>
int run(int size, const std::vector<uint8_t> &raster, int n_patches, const std::vector<uint8_t> &patches) {
  std::shared_ptr<uint8_t> data(new uint8_t[(size_t)size * size], std::default_delete<uint8_t[]>());
  std::memcpy(data.get(), raster.data(), raster.size());
  Planner planner{};
  typedef typename Planner::Map::ElemType Elem;
  Position next_point(8.0f, 8.0f), goal((float)(size - 8), (float)(size - 8));
  planner.reset();
  planner.set_occupancy_threshold(1);
  planner.set_heuristic_multiplier(1);
  planner.set_map(data, size, size);
  planner.set_start(next_point);
  planner.set_goal(goal);
  if (planner.step() != LOOP_OK) { std::printf("step failed %d\n", planner.last_error); return 2; }
  const auto &map = planner.get_expanded_map();
  const auto &grid = planner.get_grid();
  std::printf("plan expanded %lu g_start %.9g cost_start %.1f\n", planner.num_nodes_expanded,
              map.get_g(Elem((int)next_point.x, (int)next_point.y)), grid.get_cost(Cell(next_point)));
  const int psz = 31;
  for (int k = 0; k < n_patches; ++k) {
    const uint8_t *rec = patches.data() + (size_t)k * (psz * psz + 16);
    int32_t hdr[4];
    std::memcpy(hdr, rec, 16);   // top, left, start_x, start_y
    std::shared_ptr<uint8_t> patch(new uint8_t[psz * psz], std::default_delete<uint8_t[]>());
    std::memcpy(patch.get(), rec + 16, psz * psz);
    planner.patch_map(patch, hdr[0], hdr[1], psz, psz);
    planner.set_heuristic_multiplier(1);
    next_point = Position((float)hdr[2], (float)hdr[3]);
    planner.set_start(next_point);
    if (planner.step() != LOOP_OK) { std::printf("step failed %d\n", planner.last_error); return 2; }
    const Node sn(next_point);
    std::printf("replan %d updated %lu rhs_start %.9g interp %.9g consistent %d patched_cost %.1f\n", k,
                planner.num_nodes_updated, map.get_rhs(Elem(sn.x, sn.y)), map.get_interp_rhs(sn),
                (int)map.consistent(Elem(sn.x, sn.y)), grid.get_cost(Cell(hdr[0], hdr[1])));
  }
  // the queue as a caller can observe it (the public member ReplannerBase::priority_queue, ReplannerBase.h:154): the elements that are
  // not consistent; after a step none of them has a key below the start's (end_condition()).  NO_HEURISTIC: Key = float = min(g, rhs)
  {
    const auto &q = planner.priority_queue;
    float start_key = 0;                       // max over the start elements that have been reached (end_condition, FD impl:225-256)
    if (std::is_same<Elem, Cell>::value) { start_key = map.get_g(Elem(Cell(next_point).x, Cell(next_point).y)); }
    else {
      const Cell sc(next_point);
      for (int dx = 0; dx < 2; ++dx) for (int dy = 0; dy < 2; ++dy) { const float v = map.get_g(Elem(sc.x + dx, sc.y + dy)); if (v < INFINITY) start_key = std::fmax(start_key, v); }
    }
    bool ordered = true, inconsistent = true, field_g = true;
    int i = 0;
    float prev = -INFINITY;
    for (auto it = q.ordered_begin(); it != q.ordered_end(); ++it, ++i) {
      if (it->key < prev) ordered = false;
      prev = it->key;
      const auto gr = q.g_rhs(i);
      if (gr.first == gr.second) inconsistent = false;
      if (map.get_g(it->elem) != gr.first) field_g = false;
    }
    std::printf("queue size %d empty %d top_key %.9g start_key %.9g ordered %d inconsistent %d field_g %d top_is_first %d err %d\n", q.size(), (int)q.empty(),
                q.empty() ? INFINITY : (double)q.top_key(), (double)start_key, (int)ordered, (int)inconsistent, (int)field_g,
                (int)(q.empty() || (q.top_value().x == q.begin()->elem.x && q.top_value().y == q.begin()->elem.y)), q.last_error());
  }
  // the `tof` dump (main.cpp:139-156)
  const long long n = (long long)planner.map.size();
  double sum = 0;
  long long cnt = 0;
  for (const auto &b : planner.map.buckets)
    for (const auto &e : b) {
      sum += std::get<0>(e.second); ++cnt; (void)e.first.x;
      auto info = std::get<2>(e.second);      // the level-1/2 planners' back-pointer member (ExpandedMap.h:27-29)
      (void)info;
    }
  // ... against the same field probed element by element through the map's read accessors (what the path extractor does):
  // the elements that hold a value beyond the start's key are not final and not the same from run to run, so the dump is
  // held to the process's own field, not to another run's
  const int ex = std::is_same<Elem, Cell>::value ? size : size + 1;
  long long probed = 0;
  double psum = 0;
  for (int x = 0; x < ex; ++x)
    for (int y = 0; y < ex; ++y) {
      const float v = map.get_g(Elem(x, y));
      if (v < INFINITY) { ++probed; psum += v; }
    }
  std::printf("dump size %lld iterated %lld sum_g %.6f probed %lld probed_sum %.6f\n", n, cnt, sum, probed, psum);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s <FD|SG|DFM> <size> <map.raw> <n_patches> <patches.raw>\n", argv[0]); return 1; }
  const std::string algo = argv[1];
  const int size = std::atoi(argv[2]), n_patches = std::atoi(argv[4]);
  auto slurp = [](const char *path) {
    std::vector<uint8_t> v;
    if (FILE *f = std::fopen(path, "rb")) {
      std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
      v.resize((size_t)n);
      if (std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
      std::fclose(f);
    }
    return v;
  };
  const std::vector<uint8_t> raster = slurp(argv[3]), patches = slurp(argv[5]);
  if ((int)raster.size() != size * size) { std::fprintf(stderr, "bad map file\n"); return 1; }
  try {
    if (algo == "FD") return run<FieldDPlanner<1>>(size, raster, n_patches, patches);
    if (algo == "SG") return run<ShiftedGridPlanner<2>>(size, raster, n_patches, patches);
    if (algo == "DFM") return run<DFMPlanner<1>>(size, raster, n_patches, patches);
  } catch (const std::exception &e) { std::fprintf(stderr, "error: %s\n", e.what()); return 3; }
  return 1;
}
