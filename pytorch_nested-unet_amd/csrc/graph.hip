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


// ---------------------------------------------------------------------------------------------------------
// Segmented step (see common.h): a recorded program of {launch single-stream graph, record event, wait event} over real
// streams. Recording takes two passes of the same step body:
//   dry  - nothing is launched (g_dry_run); every cross-stream wait the lane scheduler asks for marks its event as NEEDED;
//   real - streams capture lazily (the first launch after a cut begins a capture on that stream); a cross-stream wait or the
//          record of a NEEDED event cuts the stream's open segment (end capture, instantiate, emit its launch) and is emitted
//          as an instruction; records of events nobody waits on across streams are dropped (same-stream order is implicit).
// ---------------------------------------------------------------------------------------------------------
thread_local bool g_dry_run = false;

// ---------------------------------------------------------------------------------------------------------
// Flag-synchronised lanes (recording mode 2). Every lane stream captures ONE single-stream graph for the whole step - the form
// ROCm replays as a batch of pre-built packets - and the cross-lane dependencies become device-side flags instead of graph edges
// or events between graph launches:
//   record of an event some other lane waits for  ->  flags[f] = step       (one thread, release, agent scope)
//   wait for an event recorded on another lane    ->  poll flags[f] >= step (one thread per flag, s_sleep between polls)
// All the records and waits a lane collects between two of its kernels ride in ONE lane_sync_kernel (the command processor's
// ~2 us per dispatch is shared by all queues: every node counts). `step` is the lane's own replay counter, bumped by the lane's
// first sync kernel, so flags never need a reset and a value left by the previous replay cannot satisfy a wait. Data visibility rides on the ordinary kernel boundaries: the signal
// kernel starts after its lane's producer kernel has completed (release at the end of a kernel), the consumer kernel starts after
// the wait kernel has exited (acquire at the start of a kernel); only the flag itself is accessed with agent-scope atomics.
// The lanes must sit on different hardware queues (a wait kernel ahead of the signal it waits for in the SAME queue would never
// end): the plan picks them by measured overlap (seg_pick_lanes) and the recording is refused when fewer than the lanes it
// uses are distinct. A wait that is not satisfied within LANE_WAIT_TIMEOUT_S sets the program's error word (host-pinned) and
// exits, so a broken schedule fails loudly at the next launch instead of hanging the GPU.
// ---------------------------------------------------------------------------------------------------------
constexpr unsigned long long LANE_WAIT_TIMEOUT_TICKS = 400000000ull;   // s_memrealtime runs at 100 MHz: 4 s
constexpr int LANE_SYNC_MAX = 12;          // signals / waits per sync kernel (more: a second kernel)
struct LaneSyncP { unsigned* flags; unsigned* ctr; unsigned* err; unsigned* derr; int bump, nsig, nwait; unsigned short sig[LANE_SYNC_MAX], wait[LANE_SYNC_MAX]; };
__global__ __launch_bounds__(64) void lane_sync_kernel(LaneSyncP p) {
  // (one wave: lane 0 reads - and, in the lane's first sync kernel, bumps - the replay counter, readfirstlane hands it round)
  const int t = threadIdx.x;
  unsigned v = 0u;
  if (t == 0) { v = *p.ctr; if (p.bump) { ++v; *p.ctr = v; } }
  const unsigned step = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  if (t < p.nsig) __hip_atomic_store(p.flags + p.sig[t], step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (t >= 32 && t - 32 < p.nwait) {
    const unsigned* flag = p.flags + p.wait[t - 32];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // (once one wait of the program has timed out, every later one gives up at its first poll: a broken schedule costs ONE
    //  timeout, not one per sync kernel; the device-side copy of the error word keeps that check off the PCIe bus)
    unsigned long long limit = __hip_atomic_load(p.derr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0ull : LANE_WAIT_TIMEOUT_TICKS;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - step) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (__builtin_amdgcn_s_memrealtime() - t0 > limit) {
        __hip_atomic_store(p.derr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.err, (unsigned)p.wait[t - 32] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
    (void)__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

namespace {
struct SegInstr { int op; hipStream_t st; hipEvent_t ev; hipGraphExec_t exec; };   // op 0 launch, 1 record, 2 wait
struct FlagOf { int f; hipStream_t st; };
struct SegRec {
  bool dry = false;
  // mode 2 (flag-synchronised lanes)
  bool flags_mode = false;
  unsigned* dflags = nullptr;                // [MAXF] flags, then [MAXL] lane counters (device memory, zeroed)
  unsigned* herr = nullptr;                  // host-pinned error word
  int nflags = 0, nwaits = 0, nsignals = 0;
  std::vector<hipStream_t> lanes;            // index -> stream (its counter is dflags[MAXF + index])
  std::vector<std::pair<hipEvent_t, FlagOf>> evflag;        // latest record of every needed event
  std::vector<std::pair<hipStream_t, int>> waited;          // (stream, flag) already waited for in this recording
  struct Pend { hipStream_t st; std::vector<int> sig, wait; bool bumped; };
  std::vector<Pend> pend;                                   // per lane: what its next sync kernel carries
  int nsync = 0;
  static constexpr int MAXF = 2048, MAXL = 16;
  Pend& pend_of(hipStream_t st) {
    for (auto& p : pend) if (p.st == st) return p;
    pend.push_back(Pend{st, {}, {}, false}); return pend.back();
  }
  void begin_capture(hipStream_t st) {
    if (is_cap(st)) return;
    const hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { fail("segment begin capture", e); return; }
    capturing.push_back(st);
  }
  // the lane's pending records and waits as one kernel on its capture, in front of whatever is launched on it next
  void flush(hipStream_t st) {
    Pend& p = pend_of(st);
    while (!p.sig.empty() || !p.wait.empty()) {
      if (lane_of(st) >= MAXL) { fail("too many lanes", hipErrorInvalidValue); return; }
      begin_capture(st);
      if (failed) return;
      LaneSyncP a; memset(&a, 0, sizeof(a));
      a.flags = dflags; a.ctr = ctr_of(st); a.err = herr; a.derr = dflags + MAXF + MAXL; a.bump = p.bumped ? 0 : 1; p.bumped = true;
      while (a.nsig < LANE_SYNC_MAX && !p.sig.empty()) { a.sig[a.nsig++] = (unsigned short)p.sig.front(); p.sig.erase(p.sig.begin()); }
      // (waits go out only with the LAST signals: a wait must not precede a record that was made before it)
      if (p.sig.empty()) while (a.nwait < LANE_SYNC_MAX && !p.wait.empty()) { a.wait[a.nwait++] = (unsigned short)p.wait.front(); p.wait.erase(p.wait.begin()); }
      hipLaunchKernelGGL(lane_sync_kernel, dim3(1), dim3(64), 0, st, a);
      ++nsync;
    }
  }
  int lane_of(hipStream_t st) {
    for (size_t k = 0; k < lanes.size(); ++k) if (lanes[k] == st) return (int)k;
    lanes.push_back(st); return (int)lanes.size() - 1;
  }
  unsigned* ctr_of(hipStream_t st) { return dflags + MAXF + lane_of(st); }
  hipStream_t main = nullptr;
  std::vector<hipEvent_t> needed;            // sorted after the dry pass
  std::vector<hipStream_t> capturing;        // streams with an open segment
  std::vector<SegInstr> prog;
  std::vector<hipGraph_t> graphs;
  std::vector<hipGraphExec_t> execs;
  bool failed = false;
  char err[160] = {0};
  bool is_cap(hipStream_t st) const { return std::find(capturing.begin(), capturing.end(), st) != capturing.end(); }
  bool is_needed(hipEvent_t e) const { return std::binary_search(needed.begin(), needed.end(), e); }
  void fail(const char* what, hipError_t e) { if (!failed) { failed = true; snprintf(err, sizeof(err), "%s: %s", what, hipGetErrorString(e)); } (void)hipGetLastError(); }
  void cut(hipStream_t st) {
    auto it = std::find(capturing.begin(), capturing.end(), st);
    if (it == capturing.end()) return;
    capturing.erase(it);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (e != hipSuccess || !g) { fail("segment end capture", e); return; }
    size_t n = 0;
    (void)hipGraphGetNodes(g, nullptr, &n);
    if (n == 0) { (void)hipGraphDestroy(g); return; }          // nothing was launched since the cut
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { fail("segment instantiate", e); (void)hipGraphDestroy(g); return; }
    graphs.push_back(g); execs.push_back(x);
    prog.push_back(SegInstr{0, st, nullptr, x});
  }
  void touch(hipStream_t st) {
    if (dry) return;
    if (flags_mode) { flush(st); begin_capture(st); return; }
    begin_capture(st);
  }
  // mode 2: the record of a needed event / a cross-lane wait, collected for the lane's next sync kernel
  void flag_signal(hipEvent_t ev, hipStream_t st) {
    if (nflags >= MAXF) { fail("too many cross-lane events", hipErrorInvalidValue); return; }
    const int f = nflags++;
    bool found = false;
    for (auto& p : evflag) if (p.first == ev) { p.second = FlagOf{f, st}; found = true; break; }
    if (!found) evflag.push_back({ev, FlagOf{f, st}});
    Pend& p = pend_of(st);
    // a record made after waits were collected: those waits belong in front of it (flush them first)
    if (!p.wait.empty()) flush(st);
    pend_of(st).sig.push_back(f);
    ++nsignals;
  }
  void flag_wait(hipStream_t st, hipEvent_t ev) {
    const FlagOf* fo = nullptr;
    for (auto& p : evflag) if (p.first == ev) { fo = &p.second; break; }
    if (!fo) { fail("wait for an event that was not recorded inside the recording", hipErrorInvalidValue); return; }
    if (fo->st == st) return;                                   // same lane: stream order
    for (auto& w : waited) if (w.first == st && w.second == fo->f) return;
    waited.push_back({st, fo->f});
    pend_of(st).wait.push_back(fo->f);
    ++nwaits;
  }
};
thread_local SegRec* g_seg = nullptr;
thread_local std::vector<hipEvent_t> g_seg_needed;      // result of the last dry pass
}  // namespace

struct nunet_seg {
  std::vector<SegInstr> prog;
  std::vector<hipGraph_t> graphs;
  std::vector<hipGraphExec_t> execs;
  hipStream_t main;
  hipEvent_t ev_in, ev_out;
  int n_launch, n_record, n_wait, n_nodes;
  unsigned* dflags; unsigned* herr;          // mode 2: flags + lane counters (device), error word (host-pinned)
};

bool seg_active() { return g_seg != nullptr; }
bool seg_wait(hipStream_t st, hipEvent_t ev) {
  SegRec* r = g_seg;
  if (!r) return false;
  if (r->dry) { r->needed.push_back(ev); return true; }
  if (r->flags_mode) { r->flag_wait(st, ev); return true; }
  hipEvent_t e = ev;
  // one wait per (stream, event) between two cuts of the stream is enough
  for (size_t k = r->prog.size(); k-- > 0;) {
    const SegInstr& i = r->prog[k];
    if (i.st != st) continue;
    if (i.op == 2 && i.ev == e) return true;
    if (i.op == 0) break;
  }
  r->cut(st);
  r->prog.push_back(SegInstr{2, st, e, nullptr});
  return true;
}
bool seg_record(hipEvent_t ev, hipStream_t st) {
  SegRec* r = g_seg;
  if (!r) return false;
  if (r->dry || !r->is_needed(ev)) return true;
  if (r->flags_mode) { r->flag_signal(ev, st); return true; }
  // (merging the records lazily - a lane cut only when another lane asks for one of its events - gives 33 instead of 55 segments
  //  per step but 2.64 instead of 2.33 ms: consumers then wait for the producer lane's whole tail; measured, not kept)
  r->cut(st);
  r->prog.push_back(SegInstr{1, st, ev, nullptr});
  return true;
}
void seg_touch(hipStream_t st) { if (g_seg) g_seg->touch(st); }

extern "C" int nunet_seg_begin(nunet_stream_t s, int32_t mode) {
  NUNET_REQUIRE(s, "seg_begin: recording needs an explicit (non-default) stream");
  NUNET_REQUIRE(!g_seg && !g_cap.active, "seg_begin: a recording / capture is already active on this thread");
  NUNET_REQUIRE(mode >= 0 && mode <= 2, "seg_begin: mode %d", mode);
  SegRec* r = new SegRec();
  r->dry = mode == NUNET_SEG_DRY; r->main = (hipStream_t)s;
  if (mode == NUNET_SEG_FLAGS) {
    r->flags_mode = true;
    const size_t bytes = sizeof(unsigned) * (SegRec::MAXF + SegRec::MAXL + 1);      // flags | lane counters | device copy of the error word
    if (hipMalloc((void**)&r->dflags, bytes) != hipSuccess || hipMemset(r->dflags, 0, bytes) != hipSuccess ||
        hipHostMalloc((void**)&r->herr, sizeof(unsigned), hipHostMallocDefault) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      nunet_set_error("seg_begin: cannot allocate the lane flags: %s", hipGetErrorString(hipGetLastError()));
      if (r->dflags) (void)hipFree(r->dflags);
      delete r;
      return NUNET_ELAUNCH;
    }
    *r->herr = 0u;
  }
  if (!r->dry) { r->needed = g_seg_needed; std::sort(r->needed.begin(), r->needed.end()); }
  g_seg = r;
  g_dry_run = r->dry;
  r->touch(r->main);            // launches the caller makes on its own stream belong to the program too
  if (r->failed) { nunet_set_error("seg_begin: %s", r->err); g_seg = nullptr; g_dry_run = false; delete r; return NUNET_ELAUNCH; }
  return NUNET_OK;
}

extern "C" void nunet_seg_destroy(nunet_seg* G);
extern "C" int nunet_seg_end(nunet_stream_t s, nunet_seg** out) {
  SegRec* r = g_seg;
  NUNET_REQUIRE(r && r->main == (hipStream_t)s, "seg_end: no recording active on this stream");
  g_seg = nullptr; g_dry_run = false;
  if (r->dry) {
    g_seg_needed = r->needed;
    delete r;
    if (out) *out = nullptr;
    return NUNET_OK;
  }
  if (r->flags_mode) { for (size_t k = 0; k < r->pend.size(); ++k) r->flush(r->pend[k].st); }
  while (!r->capturing.empty()) r->cut(r->capturing.back());
  nunet_seg* G = new nunet_seg();
  G->prog = r->prog; G->graphs = r->graphs; G->execs = r->execs; G->main = r->main; G->ev_in = G->ev_out = nullptr;
  G->dflags = r->dflags; G->herr = r->herr;
  G->n_launch = G->n_record = G->n_wait = G->n_nodes = 0;
  for (const SegInstr& i : G->prog) { if (i.op == 0) ++G->n_launch; else if (i.op == 1) ++G->n_record; else ++G->n_wait; }
  if (r->flags_mode) {
    G->n_record = r->nsignals; G->n_wait = r->nwaits;
    // the caller's lane is launched first: it heads the dependency order, the others start with a wait for one of its flags
    for (size_t k = 0; k < G->prog.size(); ++k) if (G->prog[k].st == G->main) { std::swap(G->prog[0], G->prog[k]); break; }
  }
  for (hipGraph_t g : G->graphs) { size_t n = 0; (void)hipGraphGetNodes(g, nullptr, &n); G->n_nodes += (int)n; }
  const bool failed = r->failed;
  if (failed) nunet_set_error("seg_end: %s", r->err);
  delete r;
  if (failed || !out || hipEventCreateWithFlags(&G->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&G->ev_out, hipEventDisableTiming) != hipSuccess) {
    if (!failed) nunet_set_error("seg_end: cannot create the hand-over events");
    nunet_seg_destroy(G);
    return NUNET_ELAUNCH;
  }
  *out = G;
  return NUNET_OK;
}

// Replay: the recorded program on its own streams, ordered after what the caller queued on `s`; `s` continues after it.
extern "C" int nunet_seg_launch(nunet_seg* G, nunet_stream_t s) {
  NUNET_REQUIRE(G, "seg_launch: null program");
  hipStream_t cs = (hipStream_t)s;
  hipError_t e = hipSuccess;
  if (G->herr && *G->herr) {
    nunet_set_error("seg_launch: a cross-lane wait of an earlier replay timed out (flag %u): two lanes share a hardware queue, or a lane died", *G->herr - 1u);
    return NUNET_ELAUNCH;
  }
  if (cs != G->main || G->dflags) { e = hipEventRecord(G->ev_in, cs); if (e == hipSuccess && cs != G->main) e = hipStreamWaitEvent(G->main, G->ev_in, 0); }
  for (size_t k = 0; k < G->prog.size() && e == hipSuccess; ++k) {
    const SegInstr& i = G->prog[k];
    // (flag mode: a side lane does not start polling before the caller's earlier work is done)
    if (i.op == 0 && G->dflags && i.st != G->main) e = hipStreamWaitEvent(i.st, G->ev_in, 0);
    if (e != hipSuccess) break;
    if (i.op == 0) e = hipGraphLaunch(i.exec, i.st);
    else if (i.op == 1) e = hipEventRecord(i.ev, i.st);
    else e = hipStreamWaitEvent(i.st, i.ev, 0);
  }
  if (e == hipSuccess && cs != G->main) { e = hipEventRecord(G->ev_out, G->main); if (e == hipSuccess) e = hipStreamWaitEvent(cs, G->ev_out, 0); }
  if (e != hipSuccess) { nunet_set_error("seg_launch: %s", hipGetErrorString(e)); (void)hipGetLastError(); return NUNET_ELAUNCH; }
  return NUNET_OK;
}

extern "C" int nunet_seg_info(const nunet_seg* G, int32_t* launches, int32_t* records, int32_t* waits, int32_t* kernel_nodes) {
  NUNET_REQUIRE(G && launches && records && waits && kernel_nodes, "seg_info: null pointer");
  *launches = G->n_launch; *records = G->n_record; *waits = G->n_wait; *kernel_nodes = G->n_nodes;
  return NUNET_OK;
}

extern "C" void nunet_seg_destroy(nunet_seg* G) {
  if (!G) return;
  for (hipGraphExec_t x : G->execs) (void)hipGraphExecDestroy(x);
  for (hipGraph_t g : G->graphs) (void)hipGraphDestroy(g);
  if (G->ev_in) (void)hipEventDestroy(G->ev_in);
  if (G->ev_out) (void)hipEventDestroy(G->ev_out);
  if (G->dflags) { (void)hipDeviceSynchronize(); (void)hipFree(G->dflags); }
  if (G->herr) (void)hipHostFree(G->herr);
  delete G;
}
