// Lane-faithful hipGraphs.
//
// A multi-stream capture hands ROCm a DAG; at instantiation ROCm 7.2 re-assigns every node to one of
// DEBUG_HIP_FORCE_GRAPH_QUEUES (4) internal streams by a depth-first walk from the first node: a
// node takes the stream of the parent that reaches it first PLUS the index of the edge in that
// parent's edge list, modulo 4 (read off DEBUG_HIP_GRAPH_DOT_PRINT dumps, tools/graph_sched_probe.py).
// The streams the ops were captured on are forgotten, so the lane schedule the plan computed
// (plan.hip, Sched) is only a hint: e.g. weight gradients sent to their own lane ended up queued
// behind a whole column of the grid (measured with NUNET_STAMPS, profiles/r01_summary.md).
//
// nunet_graph_begin/end capture as usual, remember the lane of every node (graph_tag_tail, called by
// the scheduler after each op), then REWRITE the edge lists of the captured graph before it is
// instantiated so that the depth-first rule reproduces the lanes exactly:
//   * edges implied by other paths are dropped (transitive reduction), lane order becomes explicit
//     edges (node -> next node of its lane) and is always a node's FIRST edge, so the walk runs down a
//     lane before it looks at cross edges;
//   * a cross edge parent(lane a) -> child(lane b) is placed at an index == (b - a) mod 4, padded in
//     front with redundant edges to nodes that are already placed (later nodes of the parent's lane);
//   * the walk is simulated on the host until every node lands on its lane.
// Nothing here changes what is computed: only redundant edges are added or removed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

constexpr int Q = 4;   // ROCm's graph stream pool (DEBUG_HIP_FORCE_GRAPH_QUEUES default)

struct Capture {
  hipStream_t origin = nullptr;
  std::vector<int> lane;          // by node creation index; -1 = not tagged (caller's stream -> lane 0)
  bool active = false;
};
thread_local Capture g_cap;

}  // namespace

struct nunet_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipStream_t launch_stream;      // see nunet_graph_end: own stream -> own hardware queue
  hipEvent_t ev_in, ev_out;
  std::vector<hipGraphNode_t> node;          // creation order
  std::vector<std::pair<int, int>> elist;    // edges in insertion order (the order matters to ROCm's stream assignment)
  int nodes, edges_before, edges_after, padded, lanes_used;
};

// Called by the plan's scheduler after every op it issues on `st` while a nunet_graph capture is
// active on this thread: all nodes created since the previous call belong to `lane`.
void graph_tag_tail(hipStream_t st, int lane) {
  if (!g_cap.active) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  if (hipStreamGetCaptureInfo_v2(st, &cs, &id, &g, &deps, &ndeps) != hipSuccess || cs != hipStreamCaptureStatusActive || !g) {
    (void)hipGetLastError();
    return;
  }
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) { (void)hipGetLastError(); return; }
  // every node created since the previous call (the scheduler calls with lane -1 when it takes over
  // from the caller's stream, so launches made outside the scheduler stay on lane 0)
  if (n > g_cap.lane.size()) g_cap.lane.resize(n, lane);
}

// An event record node behind the current tail of capturing stream `st`: each replay of the graph records `ev` when the
// tail's work is done, and streams outside the graph can wait on it (hipEventRecordWithFlags(..., hipEventRecordExternal)
// returns "invalid argument" on this runtime, so the node is added through the graph API).
int graph_record_external(hipStream_t st, hipEvent_t ev) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  if (hipStreamGetCaptureInfo_v2(st, &cs, &id, &g, &deps, &ndeps) != hipSuccess || cs != hipStreamCaptureStatusActive || !g) {
    nunet_set_error("graph_record_external: stream is not capturing (%s)", hipGetErrorString(hipGetLastError()));
    return NUNET_ELAUNCH;
  }
  std::vector<hipGraphNode_t> d(deps, deps + ndeps);
  hipGraphNode_t node = nullptr;
  hipError_t e = hipGraphAddEventRecordNode(&node, g, d.data(), d.size(), ev);
  if (e == hipSuccess) e = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
  if (e != hipSuccess) { nunet_set_error("graph_record_external: %s", hipGetErrorString(e)); (void)hipGetLastError(); return NUNET_ELAUNCH; }
  return NUNET_OK;
}

bool graph_capture_active() { return g_cap.active; }

extern "C" int nunet_graph_begin(nunet_stream_t s) {
  NUNET_REQUIRE(s, "graph_begin: capture needs an explicit (non-default) stream");
  NUNET_REQUIRE(!g_cap.active, "graph_begin: a capture is already active on this thread");
  if (hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeRelaxed) != hipSuccess) {
    nunet_set_error("graph_begin: hipStreamBeginCapture failed: %s", hipGetErrorString(hipGetLastError()));
    return NUNET_ELAUNCH;
  }
  g_cap.origin = (hipStream_t)s;
  g_cap.lane.clear();
  g_cap.active = true;
  return NUNET_OK;
}

extern "C" void nunet_graph_destroy(nunet_graph* G);

namespace {

struct Rewriter {
  int n = 0;
  std::vector<int> lane;                    // target stream of every node
  std::vector<std::vector<int>> ch;         // ordered children lists (the result)
  std::vector<std::vector<uint64_t>> reach; // descendants (bitset)
  int words = 0;
  int padded = 0;
  int mode = 1;   // 1 full; 2 re-add the captured edges unchanged; 3 reduction only (creation order); 4 lane-first order, no padding

  bool reaches(int a, int b) const { return (reach[a][b >> 6] >> (b & 63)) & 1ull; }

  // returns false when no legal padding target exists
  bool run(const std::vector<std::pair<int, int>>& edges) {
    words = (n + 63) / 64;
    std::vector<std::vector<int>> kids(n);
    for (auto& e : edges) if (e.first != e.second) kids[e.first].push_back(e.second);
    if (mode == 2) { ch = kids; return true; }
    if (mode == 5 || mode == 6) {   // captured edge set; same-lane children first (5) / last (6), creation order otherwise
      ch = kids;
      for (int v = 0; v < n; ++v) {
        std::sort(ch[v].begin(), ch[v].end());
        ch[v].erase(std::unique(ch[v].begin(), ch[v].end()), ch[v].end());
        std::stable_partition(ch[v].begin(), ch[v].end(), [&](int c) { return (lane[c] == lane[v]) == (mode == 5); });
      }
      return true;
    }
    // explicit lane order: node -> next node of its lane (creation order is a topological order)
    std::vector<int> succ(n, -1), last(Q, -1);
    for (int v = 0; v < n; ++v) {
      const int l = lane[v];
      if (last[l] >= 0) { succ[last[l]] = v; kids[last[l]].push_back(v); }
      last[l] = v;
    }
    for (int v = 0; v < n; ++v) {
      std::sort(kids[v].begin(), kids[v].end());
      kids[v].erase(std::unique(kids[v].begin(), kids[v].end()), kids[v].end());
      for (int c : kids[v]) if (c <= v) return false;   // creation order must be topological
    }
    // descendants, then transitive reduction (the lane edge always stays)
    reach.assign(n, std::vector<uint64_t>(words, 0));
    for (int v = n - 1; v >= 0; --v)
      for (int c : kids[v]) {
        reach[v][c >> 6] |= 1ull << (c & 63);
        for (int w = 0; w < words; ++w) reach[v][w] |= reach[c][w];
      }
    ch.assign(n, {});
    for (int v = 0; v < n; ++v) {
      if (succ[v] >= 0 && mode != 3) ch[v].push_back(succ[v]);
      for (int c : kids[v]) {
        if (c == succ[v] && mode != 3) continue;
        bool implied = false;
        for (int o : kids[v]) if (o != c && reaches(o, c)) { implied = true; break; }
        if (lane[c] == lane[v]) implied = true;           // same lane, not the successor: implied by lane order
        if (c == succ[v]) implied = false;
        if (!implied) ch[v].push_back(c);
      }
    }
    if (mode == 3 || mode == 4) return true;
    // simulate ROCm's depth-first stream assignment; pad until every node lands on its lane
    for (int iter = 0; iter < 4 * n + 16; ++iter) {
      std::vector<int> sid(n, -1), vtime(n, -1);
      int clock = 0, bad = -1, bad_parent = -1, bad_pos = -1;
      // explicit stack of (node, next edge position, stream counter)
      struct Fr { int v, pos, s; };
      std::vector<Fr> stk;
      int root_s = 0;
      for (int r = 0; r < n && bad < 0; ++r) {
        if (sid[r] >= 0) continue;
        sid[r] = root_s; vtime[r] = clock++;
        if (sid[r] != lane[r]) { bad = r; break; }   // only the first root can be handled (lane 0)
        stk.push_back({r, 0, root_s});
        while (!stk.empty() && bad < 0) {
          Fr& f = stk.back();
          if (f.pos >= (int)ch[f.v].size()) { stk.pop_back(); continue; }
          const int c = ch[f.v][f.pos];
          const int s = f.s;
          const int pos = f.pos;
          f.pos++; f.s = (f.s + 1) % Q;
          if (sid[c] >= 0) continue;
          sid[c] = s; vtime[c] = clock++;
          if (s != lane[c]) { bad = c; bad_parent = stk.back().v; bad_pos = pos; break; }
          stk.push_back({c, 0, s});
        }
        root_s = (root_s + 1) % Q;
      }
      if (bad < 0) return true;
      if (bad_parent < 0) return false;
      // insert redundant edges in front of `bad` until its index gives the right stream
      const int p = bad_parent;
      const int need = ((lane[bad] - lane[p]) % Q + Q) % Q;
      int add = ((need - bad_pos) % Q + Q) % Q;
      if (add == 0) return false;   // cannot happen: the index decides the stream
      std::vector<int> cand;
      for (int d = p + 1; d < n && (int)cand.size() < add; ++d) {
        if (d == bad || sid[d] < 0 || !reaches(p, d)) continue;          // placed already, descendant of p
        if (std::find(ch[p].begin(), ch[p].end(), d) != ch[p].end()) continue;
        cand.push_back(d);
      }
      if ((int)cand.size() < add) return false;
      ch[p].insert(ch[p].begin() + bad_pos, cand.begin(), cand.end());
      padded += add;
    }
    return false;
  }
};

}  // namespace

extern "C" int nunet_graph_end(nunet_stream_t s, nunet_graph** out) {
  NUNET_REQUIRE(out, "graph_end: null out");
  NUNET_REQUIRE(g_cap.active && g_cap.origin == (hipStream_t)s, "graph_end: no capture active on this stream");
  g_cap.active = false;
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture((hipStream_t)s, &g) != hipSuccess || !g) {
    nunet_set_error("graph_end: hipStreamEndCapture failed: %s", hipGetErrorString(hipGetLastError()));
    return NUNET_ELAUNCH;
  }
  size_t n = 0, ne = 0;
  std::vector<hipGraphNode_t> nodes, from, to;
  bool ok = hipGraphGetNodes(g, nullptr, &n) == hipSuccess;
  if (ok) { nodes.resize(n); ok = hipGraphGetNodes(g, nodes.data(), &n) == hipSuccess; }
  if (ok) ok = hipGraphGetEdges(g, nullptr, nullptr, &ne) == hipSuccess;
  if (ok) { from.resize(ne); to.resize(ne); ok = ne == 0 || hipGraphGetEdges(g, from.data(), to.data(), &ne) == hipSuccess; }
  if (!ok) { (void)hipGetLastError(); (void)hipGraphDestroy(g); nunet_set_error("graph_end: cannot read the captured graph"); return NUNET_ELAUNCH; }

  nunet_graph* G = new nunet_graph();
  G->graph = g; G->exec = nullptr; G->launch_stream = nullptr; G->ev_in = G->ev_out = nullptr; G->nodes = (int)n; G->edges_before = (int)ne; G->edges_after = (int)ne; G->padded = 0; G->lanes_used = 1;

  static int rewrite = -1;
  if (rewrite < 0) { const char* e = getenv("NUNET_GRAPH_REWRITE"); rewrite = e ? atoi(e) : 0; }
  bool multi_lane = false;
  for (size_t k = 0; k < g_cap.lane.size() && k < n; ++k) if (g_cap.lane[k] > 0) multi_lane = true;
  if (rewrite && multi_lane && n > 2) {
    Rewriter R;
    R.n = (int)n;
    R.mode = rewrite;
    R.lane.assign(n, 0);
    for (size_t k = 0; k < n && k < g_cap.lane.size(); ++k) R.lane[k] = g_cap.lane[k] < 0 ? 0 : g_cap.lane[k] % Q;
    std::vector<std::pair<int, int>> edges;
    {
      std::vector<std::pair<hipGraphNode_t, int>> idx(n);
      for (size_t k = 0; k < n; ++k) idx[k] = {nodes[k], (int)k};
      std::sort(idx.begin(), idx.end());
      auto find = [&](hipGraphNode_t h) {
        auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(h, -1));
        return (it != idx.end() && it->first == h) ? it->second : -1;
      };
      for (size_t k = 0; k < ne; ++k) {
        const int a = find(from[k]), b = find(to[k]);
        if (a < 0 || b < 0) { ok = false; break; }
        edges.push_back({a, b});
      }
    }
    if (ok && R.run(edges)) {
      if (ne && hipGraphRemoveDependencies(g, from.data(), to.data(), ne) != hipSuccess) ok = false;
      int added = 0;
      for (size_t v = 0; v < n && ok; ++v)
        for (int c : R.ch[v]) {
          if (hipGraphAddDependencies(g, &nodes[v], &nodes[c], 1) != hipSuccess) { ok = false; break; }
          ++added;
        }
      if (!ok) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(g); delete G;
        nunet_set_error("graph_end: rewriting the edge lists failed");
        return NUNET_ELAUNCH;
      }
      G->edges_after = added; G->padded = R.padded;
      int mx = 0; for (int l : R.lane) mx = l > mx ? l : mx;
      G->lanes_used = mx + 1;
    } else if (getenv("NUNET_GRAPH_VERBOSE")) {
      fprintf(stderr, "nunet_graph: lane rewrite not applicable, graph left as captured\n");
    }
  }
  // remember the edge lists as they stand (nunet_graph_tune permutes them)
  G->node = nodes;
  G->elist.clear();
  {
    size_t ne2 = 0;
    std::vector<hipGraphNode_t> f2, t2;
    if (hipGraphGetEdges(g, nullptr, nullptr, &ne2) == hipSuccess && ne2) {
      f2.resize(ne2); t2.resize(ne2);
      if (hipGraphGetEdges(g, f2.data(), t2.data(), &ne2) == hipSuccess) {
        std::vector<std::pair<hipGraphNode_t, int>> idx(n);
        for (size_t k = 0; k < n; ++k) idx[k] = {nodes[k], (int)k};
        std::sort(idx.begin(), idx.end());
        auto find = [&](hipGraphNode_t h) {
          auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(h, -1));
          return (it != idx.end() && it->first == h) ? it->second : -1;
        };
        for (size_t k = 0; k < ne2; ++k) { const int a = find(f2[k]), b = find(t2[k]); if (a >= 0 && b >= 0) G->elist.push_back({a, b}); }
      }
    }
    (void)hipGetLastError();
  }
  if (getenv("NUNET_GRAPH_VERBOSE"))
    fprintf(stderr, "nunet_graph: %d nodes, %d -> %d edges (%d padding), %d lanes\n", G->nodes, G->edges_before, G->edges_after, G->padded, G->lanes_used);
  // ROCm maps a new stream to the least-used of its 4 hardware queues, and instantiation creates the
  // graph's 3 extra streams. Creating the launch stream right before makes these four consecutive
  // picks, i.e. four DISTINCT hardware queues (launching on the caller's stream instead let two graph
  // streams share its queue: their kernels then ran strictly one after the other).
  static int own_stream = -1;
  if (own_stream < 0) { const char* e = getenv("NUNET_GRAPH_OWN_STREAM"); own_stream = e ? atoi(e) : 1; }
  if (own_stream) {
    if (hipStreamCreateWithFlags(&G->launch_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&G->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&G->ev_out, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      nunet_set_error("graph_end: cannot create the launch stream");
      nunet_graph_destroy(G);
      return NUNET_ELAUNCH;
    }
  }
  if (hipGraphInstantiate(&G->exec, g, nullptr, nullptr, 0) != hipSuccess) {
    nunet_set_error("graph_end: hipGraphInstantiate failed: %s", hipGetErrorString(hipGetLastError()));
    nunet_graph_destroy(G);
    return NUNET_ELAUNCH;
  }
  *out = G;
  return NUNET_OK;
}


// ---- edge-order autotuning ---------------------------------------------------------------------
// ROCm decides which of its 4 streams runs a node from the ORDER in which the edges were added
// (position in the parents' edge lists; see the header of this file), and the step time of one and the
// same graph varies by +-30 % with that order. The order carries no meaning, so it is tuned like any
// other launch parameter: swap the positions of two edges that leave (or enter) the same node,
// re-instantiate, time a few replays, keep the order when it is faster.
namespace {
struct Lcg { uint64_t s; uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); } };

bool apply_edges(nunet_graph* G) {
  size_t ne = 0;
  if (hipGraphGetEdges(G->graph, nullptr, nullptr, &ne) != hipSuccess) return false;
  if (ne) {
    std::vector<hipGraphNode_t> f(ne), t(ne);
    if (hipGraphGetEdges(G->graph, f.data(), t.data(), &ne) != hipSuccess) return false;
    if (hipGraphRemoveDependencies(G->graph, f.data(), t.data(), ne) != hipSuccess) return false;
  }
  for (auto& e : G->elist)
    if (hipGraphAddDependencies(G->graph, &G->node[e.first], &G->node[e.second], 1) != hipSuccess) return false;
  return true;
}

bool reinstantiate(nunet_graph* G) {
  if (G->exec) { (void)hipGraphExecDestroy(G->exec); G->exec = nullptr; }
  return hipGraphInstantiate(&G->exec, G->graph, nullptr, nullptr, 0) == hipSuccess;
}

// milliseconds per replay: best of 3 timings of `replays` back-to-back launches
float time_replays(nunet_graph* G, hipStream_t s, int replays, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, s);
    for (int k = 0; k < replays; ++k) if (hipGraphLaunch(G->exec, s) != hipSuccess) return -1.f;
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= replays;
    best = ms < best ? ms : best;
  }
  return best;
}
}  // namespace

extern "C" int nunet_graph_tune(nunet_graph* G, int32_t iters, int32_t replays, uint32_t seed, float* base_ms, float* best_ms) {
  NUNET_REQUIRE(G && G->exec && base_ms && best_ms && iters >= 0 && replays >= 1, "graph_tune: bad args");
  hipStream_t s = G->launch_stream;
  NUNET_REQUIRE(s, "graph_tune: needs the graph's own launch stream");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); nunet_set_error("graph_tune: events"); return NUNET_ELAUNCH; }
  (void)time_replays(G, s, replays, e0, e1);   // warm
  float cur = time_replays(G, s, replays, e0, e1);
  *base_ms = cur;
  // groups of edge positions that share their source / their destination
  const int n = (int)G->node.size(), ne = (int)G->elist.size();
  std::vector<std::vector<int>> by_src(n), by_dst(n);
  for (int k = 0; k < ne; ++k) { by_src[G->elist[k].first].push_back(k); by_dst[G->elist[k].second].push_back(k); }
  std::vector<const std::vector<int>*> groups;
  for (int v = 0; v < n; ++v) { if (by_src[v].size() >= 2) groups.push_back(&by_src[v]); if (by_dst[v].size() >= 2) groups.push_back(&by_dst[v]); }
  Lcg rng{seed * 2654435761ull + 12345};
  int rc = NUNET_OK, accepted = 0;
  for (int it = 0; it < iters && !groups.empty(); ++it) {
    const std::vector<int>& grp = *groups[rng.next() % groups.size()];
    const int a = (int)(rng.next() % grp.size());
    int b = (int)(rng.next() % (grp.size() - 1)); if (b >= a) ++b;
    // positions are permuted, group membership by position stays valid only for the VALUES: swap the edge values
    std::swap(G->elist[grp[a]], G->elist[grp[b]]);
    if (!apply_edges(G) || !reinstantiate(G)) { rc = NUNET_ELAUNCH; break; }
    (void)time_replays(G, s, 1, e0, e1);
    const float t = time_replays(G, s, replays, e0, e1);
    if (t > 0.f && t < cur * 0.997f) {
      cur = t; ++accepted;
      // the swapped edges may now sit in other groups' position lists: rebuild
      for (int v = 0; v < n; ++v) { by_src[v].clear(); by_dst[v].clear(); }
      for (int k = 0; k < ne; ++k) { by_src[G->elist[k].first].push_back(k); by_dst[G->elist[k].second].push_back(k); }
    } else {
      std::swap(G->elist[grp[a]], G->elist[grp[b]]);   // revert (applied with the next candidate / at the end)
    }
  }
  if (rc == NUNET_OK && iters > 0 && (!apply_edges(G) || !reinstantiate(G))) rc = NUNET_ELAUNCH;
  if (rc == NUNET_OK) *best_ms = time_replays(G, s, replays, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc != NUNET_OK) { (void)hipGetLastError(); nunet_set_error("graph_tune: re-instantiation failed"); return rc; }
  if (getenv("NUNET_GRAPH_VERBOSE")) fprintf(stderr, "nunet_graph: tuned %d iterations, %d accepted, %.4f -> %.4f ms\n", iters, accepted, *base_ms, *best_ms);
  return NUNET_OK;
}

extern "C" int nunet_graph_launch(nunet_graph* G, nunet_stream_t s) {
  NUNET_REQUIRE(G && G->exec, "graph_launch: null graph");
  hipStream_t cs = (hipStream_t)s;
  if (!G->launch_stream) {
    if (hipGraphLaunch(G->exec, cs) != hipSuccess) { nunet_set_error("graph_launch: %s", hipGetErrorString(hipGetLastError())); return NUNET_ELAUNCH; }
    return NUNET_OK;
  }
  // ordered after what the caller queued on `s`, and `s` continues after the replay
  if (hipEventRecord(G->ev_in, cs) != hipSuccess || hipStreamWaitEvent(G->launch_stream, G->ev_in, 0) != hipSuccess ||
      hipGraphLaunch(G->exec, G->launch_stream) != hipSuccess ||
      hipEventRecord(G->ev_out, G->launch_stream) != hipSuccess || hipStreamWaitEvent(cs, G->ev_out, 0) != hipSuccess) {
    nunet_set_error("graph_launch: %s", hipGetErrorString(hipGetLastError()));
    return NUNET_ELAUNCH;
  }
  return NUNET_OK;
}

extern "C" int nunet_graph_info(const nunet_graph* G, int32_t* nodes, int32_t* edges_captured, int32_t* edges_final, int32_t* padding, int32_t* lanes) {
  NUNET_REQUIRE(G && nodes && edges_captured && edges_final && padding && lanes, "graph_info: null pointer");
  *nodes = G->nodes; *edges_captured = G->edges_before; *edges_final = G->edges_after; *padding = G->padded; *lanes = G->lanes_used;
  return NUNET_OK;
}

extern "C" void nunet_graph_destroy(nunet_graph* G) {
  if (!G) return;
  if (G->exec) (void)hipGraphExecDestroy(G->exec);
  if (G->graph) (void)hipGraphDestroy(G->graph);
  if (G->ev_in) (void)hipEventDestroy(G->ev_in);
  if (G->ev_out) (void)hipEventDestroy(G->ev_out);
  if (G->launch_stream) (void)hipStreamDestroy(G->launch_stream);
  delete G;
}
