// hipGraph capture / launch of the fused training step.
//
// A multi-stream capture hands ROCm a DAG; at instantiation ROCm 7.2 re-assigns every node to one of
// DEBUG_HIP_FORCE_GRAPH_QUEUES (4) internal streams by a depth-first walk from the first node: a node takes the
// stream of the parent that reaches it first PLUS the index of the edge in that parent's edge list, modulo 4 (read
// off DEBUG_HIP_GRAPH_DOT_PRINT dumps, tools/graph_sched_probe.py). The streams the ops were captured on are
// forgotten, so the lane schedule the plan computed (plan.hip, Sched) is a hint. Two attempts to take control of that
// walk - rewriting the captured edge lists so that it reproduces the lanes exactly, and hill-climbing the edge order
// with timed replays - were measured (round 1: slower, and -1..2 %) and are not kept; what remains is the plain
// capture, bookkeeping of the lane of every node (graph_tag_tail) for nunet_graph_info, the external event-record
// node of the data-parallel bucket-0 signal, and a launch stream of the graph's own (its own hardware queue).
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

  {
    int mx = 0;
    for (size_t k = 0; k < g_cap.lane.size() && k < n; ++k) if (g_cap.lane[k] > mx) mx = g_cap.lane[k];
    G->lanes_used = mx + 1;
  }
  // remember the edge lists as captured
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
  // ROCm maps a new stream to the least-used of its 4 hardware queues, and instantiation creates the
  // graph's 3 extra streams. Creating the launch stream right before makes these four consecutive
  // picks, i.e. four DISTINCT hardware queues (launching on the caller's stream instead let two graph
  // streams share its queue: their kernels then ran strictly one after the other).
  {
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
