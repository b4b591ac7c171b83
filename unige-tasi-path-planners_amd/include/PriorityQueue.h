// PriorityQueue.h -- read view of the planner's queue with the observer half of the reference's
// PriorityQueue<K, V> interface (ProjectToolkit/include/PriorityQueue.h:47-63): size(), empty(),
// top_key(), top_value(), begin()/end(), ordered_begin()/ordered_end().
//
// The reference keeps a Fibonacci heap of the elements that are not consistent (G != RHS;
// ReplannerBase::enqueue_if_inconsistent, ReplannerBase.h:110-115) and pops it in key order.  The
// engine has no heap -- the order of its relaxation is kept in tile lists and queue words on the
// device -- so what a caller can observe between two steps is derived from the
// field: ufm_read_queue lists every element whose value differs from its RHS; the keys are made
// here as calculate_key makes them (FieldDPlanner_impl.h:166-186, DynamicFastMarching_impl.h:135-155).
// After a step all of them lie at or beyond the start's key (end_condition()); which elements they
// are depends on the order of the expansions out there, in the reference as here.
// The mutators (insert, insert_or_update, remove_if_present, pop, swap) are not part of the view:
// nothing outside the device decides what is relaxed next.
#ifndef UFM_PRIORITYQUEUE_H
#define UFM_PRIORITYQUEUE_H

#include <algorithm>
#include <cstdint>
#include <functional>
#include <utility>
#include <vector>

#include "ufm.h"

template <typename K, typename V>
class PriorityQueue {
 public:
  using Key = K;
  using Value = V;
  struct ElemType {
    Value elem;
    Key key;
  };
  using IteratorType = typename std::vector<ElemType>::const_iterator;
  using OrderedIteratorType = IteratorType;

  PriorityQueue() = default;

  IteratorType begin() const { refresh(); return items_.begin(); }
  IteratorType end() const { refresh(); return items_.end(); }
  /** ascending by key, as the reference's ordered iterators walk the heap */
  OrderedIteratorType ordered_begin() const { return begin(); }
  OrderedIteratorType ordered_end() const { return end(); }

  const Key &top_key() const { refresh(); return items_.front().key; }
  const Value &top_value() const { refresh(); return items_.front().elem; }
  int size() const { refresh(); return static_cast<int>(items_.size()); }
  bool empty() const { return size() == 0; }

  /** (G, RHS) of the i-th entry in key order */
  std::pair<float, float> g_rhs(int i) const { refresh(); return values_[static_cast<size_t>(i)]; }
  /** ufm code of the last refresh (UFM_OK, or why the view is empty) */
  int last_error() const { return last_error_; }

  // ---- wiring (used by ReplannerBase) ----
  /** key_of(element, min(g, rhs)) = the planner's calculate_key */
  void attach(ufm_t *h, std::function<Key(const Value &, float)> key_of) { handle_ = h; key_of_ = std::move(key_of); stale_ = true; }
  void invalidate() const { stale_ = true; }

 private:
  void refresh() const {
    if (!stale_) return;
    stale_ = false;
    items_.clear();
    values_.clear();
    if (!handle_) return;
    int total = 0;
    last_error_ = ufm_read_queue(handle_, 0, nullptr, nullptr, &total);
    if (last_error_ != UFM_OK || total == 0) return;
    std::vector<int32_t> xy(static_cast<size_t>(total) * 2);
    std::vector<float> gr(static_cast<size_t>(total) * 2);
    int again = 0;
    last_error_ = ufm_read_queue(handle_, total, xy.data(), gr.data(), &again);
    if (last_error_ != UFM_OK) return;
    const int n = std::min(total, again);
    std::vector<int> order(static_cast<size_t>(n));
    std::vector<ElemType> raw;
    raw.reserve(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
      order[static_cast<size_t>(i)] = i;
      const Value v(xy[2 * static_cast<size_t>(i)], xy[2 * static_cast<size_t>(i) + 1]);
      raw.push_back(ElemType{v, key_of_(v, std::min(gr[2 * static_cast<size_t>(i)], gr[2 * static_cast<size_t>(i) + 1]))});
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return raw[static_cast<size_t>(a)].key < raw[static_cast<size_t>(b)].key; });
    items_.reserve(static_cast<size_t>(n));
    values_.reserve(static_cast<size_t>(n));
    for (int i : order) {
      items_.push_back(raw[static_cast<size_t>(i)]);
      values_.emplace_back(gr[2 * static_cast<size_t>(i)], gr[2 * static_cast<size_t>(i) + 1]);
    }
  }

  ufm_t *handle_ = nullptr;
  std::function<Key(const Value &, float)> key_of_;
  mutable bool stale_ = true;
  mutable int last_error_ = 0;
  mutable std::vector<ElemType> items_;
  mutable std::vector<std::pair<float, float>> values_;
};

#endif  // UFM_PRIORITYQUEUE_H
