// Graph.h -- host-side mirror of the reference's Graph (ProjectToolkit/include/Graph.h:20-66,
// ProjectToolkit/Graph.cpp): the cost raster shared with the caller, start / goal, validity
// tests, neighbour enumeration.  The planner keeps its own copy of the raster in HBM
// (ufm_set_map / ufm_patch_map); this object keeps the caller's raster coherent on the host
// exactly as the reference does (it mutates the shared buffer in place, Graph.cpp:36-51), so
// consumers such as a path extractor read costs without touching the device.
#ifndef UFM_GRAPH_H
#define UFM_GRAPH_H

#include <cmath>
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

#include "GridTypes.h"
#include "Macros.h"

typedef std::pair<Node, Node> Edge;

class Graph {
 public:
  std::shared_ptr<uint8_t> map_;
  std::vector<Cell> updated_cells_;

  Cell start_cell_, goal_cell_;
  Node start_node_, goal_node_;
  Position start_pos_, goal_pos_;

  int length_ = 0, width_ = 0;
  float flength_ = 0, fwidth_ = 0;
  int size_ = 0;
  int occupancy_threshold_uchar_ = 254;   // Graph.h:34

  void set_start(const Position &s) { start_pos_ = s; start_cell_ = Cell(s); start_node_ = Node(s); }   // Graph.cpp:6-10
  void set_goal(const Position &g) { goal_pos_ = g; goal_cell_ = Cell(g); goal_node_ = Node(g); }       // Graph.cpp:12-16
  void set_occupancy_threshold(float t) { occupancy_threshold_uchar_ = static_cast<int>(t * 255.0f); }  // Graph.cpp:18-20

  void init(std::shared_ptr<uint8_t> image, int width, int length) {   // Graph.cpp:22-29
    length_ = length; width_ = width;
    flength_ = static_cast<float>(length); fwidth_ = static_cast<float>(width);
    size_ = length * width;
    map_ = std::move(image);
  }
  uint8_t &get(int x, int y) { return map_.get()[x * width_ + y]; }

  // Graph.cpp:36-51: overwrite the rectangle, remember which cells changed (reset on every call)
  void update(const std::shared_ptr<uint8_t> &patch, int x, int y, int w, int h) {
    updated_cells_.clear();
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j) {
        uint8_t &mv = get(x + i, y + j);
        const uint8_t pv = patch.get()[i * w + j];
        if (mv != pv) updated_cells_.emplace_back(x + i, y + j);
        mv = pv;
      }
  }

  float get_cost(const Cell &c) const {   // Graph.cpp:262-268
    if (!is_valid(c)) return INFINITY;
    const int v = map_.get()[c.x * width_ + c.y];
    return v >= occupancy_threshold_uchar_ ? INFINITY : static_cast<float>(v);
  }

  bool is_valid(const Node &s) const { return s.x >= 0 && s.y >= 0 && s.x <= length_ && s.y <= width_; }
  bool is_valid(const Position &p) const { return p.x >= 0.0f && p.x <= flength_ && p.y >= 0.0f && p.y <= fwidth_; }
  bool is_valid(const Cell &c) const { return c.x >= 0 && c.x < length_ && c.y >= 0 && c.y < width_; }
  bool is_valid_vertex(const Position &p) const { return std::ceil(p.x) == p.x && std::ceil(p.y) == p.y && is_valid(p); }

  // enumeration orders of Graph.cpp:71-149
  std::vector<Node> neighbors_8(const Node &s, bool include_invalid = false) const {
    return filter<Node>({s.top_node(), s.top_left_node(), s.left_node(), s.bottom_left_node(), s.bottom_node(),
                         s.bottom_right_node(), s.right_node(), s.top_right_node()}, include_invalid);
  }
  std::vector<Cell> neighbors_8(const Cell &s, bool include_invalid = false) const {
    return filter<Cell>({s.top_cell(), s.top_left_cell(), s.left_cell(), s.bottom_left_cell(), s.bottom_cell(),
                         s.bottom_right_cell(), s.right_cell(), s.top_right_cell()}, include_invalid);
  }
  std::vector<Node> neighbors_4(const Node &s, bool include_invalid = false) const {
    return filter<Node>({s.top_node(), s.left_node(), s.bottom_node(), s.right_node()}, include_invalid);
  }
  std::vector<Cell> neighbors_4(const Cell &s, bool include_invalid = false) const {
    return filter<Cell>({s.top_cell(), s.left_cell(), s.bottom_cell(), s.right_cell()}, include_invalid);
  }
  std::vector<Node> neighbors_diag_4(const Node &s, bool include_invalid = false) const {
    return filter<Node>({s.top_left_node(), s.bottom_left_node(), s.top_right_node(), s.bottom_right_node()}, include_invalid);
  }

  // the (up to 8) pairs of consecutive valid neighbours around a node, Graph.cpp:202-230
  std::vector<Edge> consecutive_neighbors(const Node &s) const {
    static const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    std::vector<Node> ring;
    ring.reserve(8);
    for (int i = 0; i < 8; ++i) ring.emplace_back(s.x + dx[i], s.y + dy[i]);
    return pair_up(ring);
  }
  // ... around a position on a node or on a cell edge, Graph.cpp:151-200
  std::vector<Edge> consecutive_neighbors(const Position &p) const {
    float ix, iy;
    const float fx = std::modf(p.x, &ix), fy = std::modf(p.y, &iy);
    std::vector<Node> ring;
    auto add = [&](float a, float b) { ring.emplace_back(static_cast<int>(a), static_cast<int>(b)); };
    if (fx > 0.0f && fx < 1.0f) {          // on an edge between (ix,iy) and (ix+1,iy): two cells
      add(ix, iy); add(ix, iy - 1); add(ix + 1, iy - 1); add(ix + 1, iy); add(ix + 1, iy + 1); add(ix, iy + 1);
    } else if (fy > 0.0f && fy < 1.0f) {   // on an edge between (ix,iy) and (ix,iy+1)
      add(ix, iy); add(ix + 1, iy); add(ix + 1, iy + 1); add(ix, iy + 1); add(ix - 1, iy + 1); add(ix - 1, iy);
    } else {                               // on a node: four cells
      add(ix + 1, iy); add(ix + 1, iy + 1); add(ix, iy + 1); add(ix - 1, iy + 1);
      add(ix - 1, iy); add(ix - 1, iy - 1); add(ix, iy - 1); add(ix + 1, iy - 1);
    }
    return pair_up(ring);
  }

  // 45-degree rotations of s' around s, Graph.cpp:232-260 (ring order: top, top_right, right, ...)
  optional<Node> ccw_neighbor(const Node &s, const Node &sp) const { return rotate(s, sp, +1); }
  optional<Node> cw_neighbor(const Node &s, const Node &sp) const { return rotate(s, sp, -1); }

 private:
  template <typename E>
  std::vector<E> filter(std::vector<E> v, bool include_invalid) const {
    if (include_invalid) return v;
    std::vector<E> out;
    out.reserve(v.size());
    for (const E &e : v) if (is_valid(e)) out.push_back(e);
    return out;
  }
  std::vector<Edge> pair_up(const std::vector<Node> &ring) const {
    std::vector<Edge> out;
    const size_t n = ring.size();
    for (size_t i = 0; i < n; ++i) {
      if (!is_valid(ring[i])) continue;
      if (is_valid(ring[(i + 1) % n])) out.emplace_back(ring[i], ring[(i + 1) % n]);
      else ++i;   // the next pair starts with the invalid node: skip it (Graph.cpp:224-226)
    }
    return out;
  }
  optional<Node> rotate(const Node &s, const Node &sp, int dir) const {
    static const int rx[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, ry[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    for (int i = 0; i < 8; ++i)
      if (sp.x - s.x == rx[i] && sp.y - s.y == ry[i]) {
        const int j = (i + dir + 8) & 7;
        const Node r(s.x + rx[j], s.y + ry[j]);
        if (is_valid(r)) return r;
        return nullopt;
      }
    return nullopt;
  }
};

#endif  // UFM_GRAPH_H
