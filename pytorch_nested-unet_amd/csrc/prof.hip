// prof.hip — per-dispatch begin/end timing (hipExtLaunchKernelGGL start/stop events) aggregated per kernel class (see common.h).
#include <string.h>

#include <vector>

#include "common.h"

thread_local bool g_prof_on = false;
thread_local int g_prof_alg_cin = 0;

struct ProfRec { int cls; double flops, bytes; std::vector<hipEvent_t> ev; };   // ev: start/stop pairs of the scope's kernels
static thread_local std::vector<ProfRec> g_recs;
static thread_local std::vector<size_t> g_open;       // stack of open scopes (indices into g_recs)
static thread_local std::vector<hipEvent_t> g_pool;
static thread_local size_t g_pool_used = 0;

static hipEvent_t get_event() {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    g_pool.push_back(e);
  }
  return g_pool[g_pool_used++];
}
void nunet_prof_push(int cls, double flops, double bytes) {
  ProfRec r; r.cls = cls; r.flops = flops; r.bytes = bytes;
  g_open.push_back(g_recs.size());
  g_recs.push_back(r);
}
void nunet_prof_pop() { if (!g_open.empty()) g_open.pop_back(); }
void nunet_prof_kernel_events(hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (g_open.empty()) return;
  ProfRec& r = g_recs[g_open.back()];
  *e0 = get_event(); *e1 = get_event();
  r.ev.push_back(*e0); r.ev.push_back(*e1);
}

static const char* kNames[PC_COUNT] = {
    "conv3x3_fwd_dgrad<BM256,BN32>", "conv3x3_fwd_dgrad<BM128,BN64>", "conv3x3_fwd_dgrad<BM128,BN32>", "conv3x3_wgrad(Cout=32)", "conv3x3_wgrad(Cout>=64)",
    "bn_relu_fwd(+pool)", "bn_relu_bwd_reduce", "bn_relu_bwd_apply", "upsample2x_fwd", "upsample2x_bwd",
    "maxpool2x2", "head_1x1", "pack_weights", "wgrad_slab_reduce/unpack", "bce_dice+iou", "sgd_step", "layout"};

extern "C" int nunet_profile_begin(void) {
  g_recs.clear(); g_open.clear();
  g_pool_used = 0;
  g_prof_on = true;
  return NUNET_OK;
}
// Host-synchronising by design: waits for the recorded events, then aggregates.
extern "C" int nunet_profile_end(nunet_prof_entry* out, int32_t max_entries, int32_t* n_out) {
  g_prof_on = false;
  NUNET_REQUIRE(out && n_out && max_entries >= PC_COUNT, "profile_end: need room for %d entries", PC_COUNT);
  for (int c = 0; c < PC_COUNT; ++c) {
    memset(&out[c], 0, sizeof(out[c]));
    strncpy(out[c].name, kNames[c], sizeof(out[c].name) - 1);
  }
  for (size_t i = 0; i < g_recs.size(); ++i) {
    const ProfRec& r = g_recs[i];
    double ms_sum = 0.0;
    for (size_t k = 0; k + 1 < r.ev.size(); k += 2) {
      if (hipEventSynchronize(r.ev[k + 1]) != hipSuccess) { nunet_set_error("profile_end: event sync failed"); return NUNET_ELAUNCH; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, r.ev[k], r.ev[k + 1]);     // begin -> end of that one dispatch
      ms_sum += ms;
    }
    out[r.cls].launches += 1; out[r.cls].ms += ms_sum; out[r.cls].flops += r.flops; out[r.cls].bytes += r.bytes;
  }
  *n_out = PC_COUNT;
  g_recs.clear();
  g_pool_used = 0;
  return NUNET_OK;
}

// Diagnostic for tools/graph_sched_probe.py: a kernel of `tag` workgroups whose first wave spins
// for `us` microseconds of the 100 MHz wall clock (bounded: always exits). The workgroup count
// identifies the launch in a rocprofv3 kernel trace.
__global__ void debug_spin_kernel(int ticks) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < (long long)ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int nunet_debug_spin(int32_t us, int32_t tag, nunet_stream_t s) {
  NUNET_REQUIRE(us >= 0 && us <= 2000 && tag >= 1 && tag <= 4096, "debug_spin: us in [0,2000], tag in [1,4096]");
  NUNET_LAUNCH(debug_spin_kernel, dim3(tag), dim3(64), 0, (hipStream_t)s, us * 100);
  return nunet_check_launch("debug_spin");
}
