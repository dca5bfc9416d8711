// uvol_ws.hpp — lifetime-shared device workspaces (host side).  An array is described by its size and the first / last pipeline
// phase that touches it; arrays whose lifetimes do not overlap share addresses (greedy first-fit over the live intervals, largest
// first).  Arrays that must start out zero carry phase UVOL_WS_PINNED: they form the head of the workspace (never shared), which
// one clear kernel zeroes per batch.  Used by the geometry encoder (geom_encode.hip) and the geometry decoder (geom_decode.hip).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#define UVOL_WS_PINNED (-1)
struct UvolWsItem { size_t bytes; int first, last; size_t off; };
// -> offsets in items[].off; returns the total size, *zero = size of the zero-initialised head
static inline size_t uvol_ws_place(std::vector<UvolWsItem> &items, size_t *zero, int n_phases, const char *what) {
  auto a256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  for (auto &it : items) if (it.first == UVOL_WS_PINNED) { it.off = off; off = a256(off + it.bytes); }
  *zero = off;
  std::vector<size_t> order;
  for (size_t i = 0; i < items.size(); i++) if (items[i].first != UVOL_WS_PINNED) order.push_back(i);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return items[a].bytes > items[b].bytes; });
  std::vector<size_t> placed; std::vector<std::pair<size_t, size_t>> busy;
  size_t total = off;
  for (size_t i : order) {
    UvolWsItem &it = items[i];
    busy.clear();
    for (size_t j : placed) if (items[j].first <= it.last && it.first <= items[j].last) busy.emplace_back(items[j].off, a256(items[j].off + items[j].bytes));
    std::sort(busy.begin(), busy.end());
    size_t cur = *zero;
    for (auto &b : busy) { if (cur + it.bytes <= b.first) break; cur = std::max(cur, b.second); }
    it.off = cur; total = std::max(total, a256(cur + it.bytes));
    placed.push_back(i);
  }
  static const bool dump = [] { const char *e = getenv("UVOL_WS_DUMP"); return e && *e == '1'; }();
  if (dump) {
    fprintf(stderr, "[uvol-ws] %s: zero head %.2f MB, total %.2f MB\n", what, *zero / 1e6, total / 1e6);
    for (int ph = 0; ph < n_phases; ph++) { size_t live = 0; for (auto &it : items) if (it.first != UVOL_WS_PINNED && it.first <= ph && ph <= it.last) live += a256(it.bytes); fprintf(stderr, "[uvol-ws]   phase %2d: %.2f MB live\n", ph, live / 1e6); }
  }
  return total;
}
