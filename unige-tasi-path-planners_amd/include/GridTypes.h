// GridTypes.h -- Position / Node / Cell value types of the planner surface.
//
// Same names, members and semantics as the reference's ProjectToolkit types
// (ProjectToolkit/include/{Position,Node,Cell}.h, ProjectToolkit/{Position,Node,Cell}.cpp),
// written header-only so that a driver needs nothing but these headers and libufm.so.
// Conventions kept from the reference: x = row (north -> south), y = column (west -> east);
// Node(Position) and Cell(Position) ROUND (Node.cpp:14-17, Cell.cpp:20-21); the default Cell is
// (-1,-1) (Cell.cpp:10) and the default Node (0,0); a Cell's corner nodes are (x,y) (x+1,y)
// (x,y+1) (x+1,y+1) (Cell.cpp:48-60); a Node's cells are (x-1,y-1) (x-1,y) (x,y-1) (x,y)
// (Node.cpp:44-50).
#ifndef UFM_GRID_TYPES_H
#define UFM_GRID_TYPES_H

#include <cmath>
#include <cstddef>
#include <functional>
#include <utility>
#include <vector>

class Node;
class Cell;

class Position {
 public:
  float x{}, y{};
  Position() = default;
  Position(float x_, float y_) : x(x_), y(y_) {}
  explicit Position(const std::pair<float, float> &o) : x(o.first), y(o.second) {}
  Position(const Node &n);   // NOLINT: implicit, as in the reference (Position.h)
  Position(const Cell &c);   // NOLINT: cell centre
  bool operator==(const Position &o) const { return x == o.x && y == o.y; }
  bool operator!=(const Position &o) const { return !(*this == o); }
  float distance(const Position &n) const { return std::hypot(x - n.x, y - n.y); }
  bool aligned(const Position &p) const { return x == p.x || y == p.y; }
};

class Node {
 public:
  int x{}, y{};
  Node() = default;
  Node(int x_, int y_) : x(x_), y(y_) {}
  explicit Node(const std::pair<int, int> &o) : x(o.first), y(o.second) {}
  explicit Node(const Position &p) : x(static_cast<int>(std::roundf(p.x))), y(static_cast<int>(std::roundf(p.y))) {}
  bool operator==(const Node &o) const { return x == o.x && y == o.y; }
  bool operator!=(const Node &o) const { return !(*this == o); }
  Node top_node() const { return {x - 1, y}; }
  Node top_left_node() const { return {x - 1, y - 1}; }
  Node top_right_node() const { return {x - 1, y + 1}; }
  Node bottom_node() const { return {x + 1, y}; }
  Node bottom_left_node() const { return {x + 1, y - 1}; }
  Node bottom_right_node() const { return {x + 1, y + 1}; }
  Node left_node() const { return {x, y - 1}; }
  Node right_node() const { return {x, y + 1}; }
  inline Cell top_left_cell() const;
  inline Cell top_right_cell() const;
  inline Cell bottom_left_cell() const;
  inline Cell bottom_right_cell() const;
  inline Cell neighbor_cell(bool bottom_TOP, bool left_RIGHT) const;
  inline std::vector<Cell> cells() const;
  float distance(const Node &n) const { return static_cast<float>(std::hypot(x - n.x, y - n.y)); }
  float distance(const Position &n) const { return std::hypot(static_cast<float>(x) - n.x, static_cast<float>(y) - n.y); }
  bool aligned(const Node &p) const { return x == p.x || y == p.y; }
};

class Cell {
 public:
  int x{-1}, y{-1};
  Cell() = default;
  Cell(int x_, int y_) : x(x_), y(y_) {}
  explicit Cell(const std::pair<int, int> &o) : x(o.first), y(o.second) {}
  explicit Cell(const Position &p) : x(static_cast<int>(std::roundf(p.x))), y(static_cast<int>(std::roundf(p.y))) {}
  bool operator==(const Cell &o) const { return x == o.x && y == o.y; }
  bool operator!=(const Cell &o) const { return !(*this == o); }
  Cell top_cell() const { return {x - 1, y}; }
  Cell top_left_cell() const { return {x - 1, y - 1}; }
  Cell top_right_cell() const { return {x - 1, y + 1}; }
  Cell bottom_cell() const { return {x + 1, y}; }
  Cell bottom_left_cell() const { return {x + 1, y - 1}; }
  Cell bottom_right_cell() const { return {x + 1, y + 1}; }
  Cell left_cell() const { return {x, y - 1}; }
  Cell right_cell() const { return {x, y + 1}; }
  Node top_left_node() const { return {x, y}; }
  Node top_right_node() const { return {x + 1, y}; }
  Node bottom_left_node() const { return {x, y + 1}; }
  Node bottom_right_node() const { return {x + 1, y + 1}; }
  Position center() const { return {static_cast<float>(x) + 0.5f, static_cast<float>(y) + 0.5f}; }
  std::vector<Node> corners() const { return {top_left_node(), top_right_node(), bottom_left_node(), bottom_right_node()}; }
  bool has_node(const Node &n) const { return (n.x == x || n.x == x + 1) && (n.y == y || n.y == y + 1); }
  float distance(const Cell &n) const { return static_cast<float>(std::hypot(x - n.x, y - n.y)); }
};

inline Position::Position(const Node &n) : x(static_cast<float>(n.x)), y(static_cast<float>(n.y)) {}
inline Position::Position(const Cell &c) : Position(c.center()) {}
inline Cell Node::top_left_cell() const { return {x - 1, y - 1}; }
inline Cell Node::top_right_cell() const { return {x - 1, y}; }
inline Cell Node::bottom_left_cell() const { return {x, y - 1}; }
inline Cell Node::bottom_right_cell() const { return {x, y}; }
inline Cell Node::neighbor_cell(bool bottom_TOP, bool left_RIGHT) const {
  return Cell(bottom_TOP ? x - 1 : x, left_RIGHT ? y : y - 1);
}
inline std::vector<Cell> Node::cells() const {
  return {top_left_cell(), top_right_cell(), bottom_left_cell(), bottom_right_cell()};
}

namespace std {
template <> struct hash<Node> {
  size_t operator()(const Node &n) const noexcept { return (static_cast<size_t>(static_cast<unsigned>(n.x)) << 32) ^ static_cast<unsigned>(n.y); }
};
template <> struct hash<Cell> {
  size_t operator()(const Cell &n) const noexcept { return (static_cast<size_t>(static_cast<unsigned>(n.x)) << 32) ^ static_cast<unsigned>(n.y); }
};
}  // namespace std

#endif  // UFM_GRID_TYPES_H
