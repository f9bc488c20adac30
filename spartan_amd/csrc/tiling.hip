// Tiling solver of the auto-tiling pass (host code only; no kernel in this file).
// Reference: spartan/expr/operator/tiling.cc (mincost_tiling :85-170, maxedge_tiling :263-369, best_tiling
// :371-408) -- a CPython-2 extension the optimiser's AutomaticTiling pass calls with the tiling graph it built
// (optimize.py:459-1054).  The graph: nodes are (expression, tiling) alternatives, a GROUP is the set of
// alternatives of one expression of which exactly one is chosen, an edge (u, v, cost) is paid when both u and v are
// chosen (ungrouped nodes always are).  Costs here are BYTES OVER xGMI LINKS (spartan_amd/expr/tiling.py), not the
// reference's unit costs.
// Small problems are solved exactly (depth-first over the groups with the running cost as the bound -- what
// best_tiling does by enumeration); larger ones by a greedy start and single-group moves until no move helps.
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "sp_common.hpp"

namespace {

struct Pair {      // edge between alternatives of two different groups
  int g1, s1, g2, s2;
  double cost;
};

struct Problem {
  int n_groups;
  std::vector<int> size;                          // alternatives per group
  std::vector<std::vector<double>> unary;         // [g][s]: edges to always-chosen nodes
  std::vector<Pair> pairs;
  std::vector<std::vector<int>> pairs_of;         // [g] -> indices into pairs
  double constant;
};

double total_cost(const Problem& p, const std::vector<int>& c) {
  double t = p.constant;
  for (int g = 0; g < p.n_groups; ++g) t += p.unary[g][c[g]];
  for (const Pair& e : p.pairs)
    if (c[e.g1] == e.s1 && c[e.g2] == e.s2) t += e.cost;
  return t;
}

// cost added by fixing group g to slot s given the groups < g already fixed (pairs are charged at their LATER group)
double added_cost(const Problem& p, const std::vector<int>& c, int g, int s) {
  double t = p.unary[g][s];
  for (int i : p.pairs_of[g]) {
    const Pair& e = p.pairs[i];
    if (e.g1 == g && e.s1 == s && e.g2 < g && c[e.g2] == e.s2) t += e.cost;
    if (e.g2 == g && e.s2 == s && e.g1 < g && c[e.g1] == e.s1) t += e.cost;
  }
  return t;
}

void dfs(const Problem& p, int g, std::vector<int>& cur, double cost, std::vector<int>& best, double& best_cost,
         int64_t& budget) {
  if (cost >= best_cost || budget <= 0) return;
  if (g == p.n_groups) {
    best = cur;
    best_cost = cost;
    return;
  }
  --budget;
  // cheapest alternative first: good bounds early
  std::vector<std::pair<double, int>> order;
  for (int s = 0; s < p.size[g]; ++s) order.push_back({added_cost(p, cur, g, s), s});
  std::sort(order.begin(), order.end());
  for (auto& o : order) {
    cur[g] = o.second;
    dfs(p, g + 1, cur, cost + o.first, best, best_cost, budget);
  }
  cur[g] = 0;
}

}  // namespace

extern "C" int sp_tiling_solve(int32_t n_nodes, int64_t n_edges, const int32_t* edge_u, const int32_t* edge_v,
                               const double* edge_cost, int32_t n_groups, const int32_t* group_ptr,
                               const int32_t* group_nodes, int32_t* choice, double* total) {
  if (n_nodes < 0 || n_edges < 0 || n_groups < 0) SP_FAIL("sp_tiling_solve: bad sizes");
  if ((n_edges && (!edge_u || !edge_v || !edge_cost)) || (n_groups && (!group_ptr || !group_nodes || !choice)))
    SP_FAIL("sp_tiling_solve: NULL pointer");
  std::vector<int> grp(n_nodes, -1), slot(n_nodes, -1);
  Problem p;
  p.n_groups = n_groups;
  p.constant = 0;
  p.size.resize(n_groups);
  p.unary.resize(n_groups);
  p.pairs_of.resize(n_groups);
  for (int g = 0; g < n_groups; ++g) {
    p.size[g] = group_ptr[g + 1] - group_ptr[g];
    if (p.size[g] < 1) SP_FAIL("sp_tiling_solve: empty group %d", g);
    p.unary[g].assign(p.size[g], 0.0);
    for (int s = 0; s < p.size[g]; ++s) {
      const int n = group_nodes[group_ptr[g] + s];
      if (n < 0 || n >= n_nodes) SP_FAIL("sp_tiling_solve: node %d out of range", n);
      if (grp[n] >= 0) SP_FAIL("sp_tiling_solve: node %d is in two groups", n);
      grp[n] = g;
      slot[n] = s;
    }
  }
  for (int64_t i = 0; i < n_edges; ++i) {
    const int u = edge_u[i], v = edge_v[i];
    if (u < 0 || u >= n_nodes || v < 0 || v >= n_nodes) SP_FAIL("sp_tiling_solve: edge %lld out of range", (long long)i);
    if (edge_cost[i] < 0) SP_FAIL("sp_tiling_solve: negative cost");
    const int gu = grp[u], gv = grp[v];
    if (gu < 0 && gv < 0) p.constant += edge_cost[i];
    else if (gu < 0) p.unary[gv][slot[v]] += edge_cost[i];
    else if (gv < 0) p.unary[gu][slot[u]] += edge_cost[i];
    else if (gu == gv) {
      if (u == v) p.unary[gu][slot[u]] += edge_cost[i];   // (two alternatives of one group are never both chosen)
    } else {
      p.pairs_of[gu].push_back((int)p.pairs.size());
      p.pairs_of[gv].push_back((int)p.pairs.size());
      p.pairs.push_back({gu, slot[u], gv, slot[v], edge_cost[i]});
    }
  }
  // greedy start + single-group moves
  std::vector<int> best(n_groups, 0);
  for (int g = 0; g < n_groups; ++g) {
    double bc = -1;
    for (int s = 0; s < p.size[g]; ++s) {
      const double c = added_cost(p, best, g, s);
      if (bc < 0 || c < bc) {
        bc = c;
        best[g] = s;
      }
    }
  }
  double best_cost = total_cost(p, best);
  for (int sweep = 0; sweep < 64; ++sweep) {
    bool moved = false;
    for (int g = 0; g < n_groups; ++g) {
      const int keep = best[g];
      for (int s = 0; s < p.size[g]; ++s) {
        if (s == keep) continue;
        best[g] = s;
        const double c = total_cost(p, best);
        if (c < best_cost) {
          best_cost = c;
          moved = true;
        } else {
          best[g] = keep;
        }
        if (best[g] == s) break;
      }
    }
    if (!moved) break;
  }
  // exact search, bounded by the local optimum, while it is affordable
  if (n_groups <= 24) {
    std::vector<int> cur(n_groups, 0), found = best;
    double found_cost = best_cost + 1e-9 * (best_cost + 1.0);
    int64_t budget = 4000000;
    dfs(p, 0, cur, p.constant, found, found_cost, budget);
    if (found_cost < best_cost) {
      best = found;
      best_cost = total_cost(p, best);
    }
  }
  for (int g = 0; g < n_groups; ++g) choice[g] = best[g];
  if (total) *total = best_cost;
  return 0;
}
