/*
 * nunet_diag.h - diagnostic and test hooks of libnunet.so. NOT part of the product boundary (include/nunet.h):
 * nothing in the package's training / evaluation path calls these; tools/ and tests/ do.
 */
#ifndef NUNET_DIAG_H
#define NUNET_DIAG_H

#include "nunet.h"

#ifdef __cplusplus
extern "C" {
#endif

/* use caller-owned streams as lanes (n >= 1, cycled over the 10 lanes) instead of the plan's own (tools/graph_sched_probe.py) */
int nunet_plan_set_lanes(nunet_plan* p, nunet_stream_t* lanes, int32_t n);
/* Diagnostic: with NUNET_STAMPS=1 in the environment every op the plan schedules is followed by a
 * 1-thread kernel that stores the 100 MHz wall clock; this reads them back (synchronises) for the
 * last forward (pass 0) / backward (pass 1), labels one per line ("L<lane> B<i><j>.<op>").
 * Works inside hipGraph replays, where a profiler's dispatch overhead would distort the timeline. */
int nunet_plan_stamps_read(nunet_plan* plan, int32_t pass, uint64_t* ticks, int32_t cap, int32_t* n_out, char* labels, int32_t label_bytes);
/* Diagnostic: a 1-thread kernel on `stream` stores the chip-wide 100 MHz clock to *dst when it runs (works inside a hipGraph:
 * when did this point of the graph execute, relative to another stamp). */
int nunet_debug_stamp(uint64_t* dst, nunet_stream_t stream);
/* Diagnostic (tools/graph_sched_probe.py): `tag` workgroups, the first spins `us` microseconds. */
int nunet_debug_spin(int32_t us, int32_t tag, nunet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NUNET_DIAG_H */
