// libposegan_hip: error reporting, version, HIP-event timing helpers.
#include "common.h"
#include <cstdlib>

namespace pg {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int& last_info() {
  static thread_local int v = 0;
  return v;
}
int& deterministic_flag() {
  static int v = [] { const char* e = getenv("PG_DETERMINISTIC"); return (e != nullptr && e[0] != '0') ? 1 : 0; }();
  return v;
}
}  // namespace pg

// PG_DETERMINISTIC at run time (the environment variable sets the initial value): see common.h
extern "C" int pg_set_deterministic(int32_t on) { pg::deterministic_flag() = on ? 1 : 0; return 0; }
extern "C" int pg_get_deterministic(void) { return pg::deterministic_flag(); }

extern "C" int pg_last_launch_info(void) { return pg::last_info(); }
extern "C" const char* pg_last_error(void) { return pg::err_buf(); }
extern "C" int pg_version(void) { return 100; }

extern "C" int pg_event_create(void** ev) {
  PG_REQUIRE(ev != nullptr, "pg_event_create: null");
  hipEvent_t e;
  PG_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return 0;
}
extern "C" int pg_event_record(void* ev, void* stream) {
  PG_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return 0;
}
extern "C" int pg_event_elapsed_ms(void* start, void* stop, float* ms) {
  PG_REQUIRE(ms != nullptr, "pg_event_elapsed_ms: null");
  PG_HIP(hipEventSynchronize((hipEvent_t)stop));
  PG_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}
extern "C" int pg_event_destroy(void* ev) {
  PG_HIP(hipEventDestroy((hipEvent_t)ev));
  return 0;
}

// Test aid (tests/test_gpu_round3.py: stream-ordering stress of the data-parallel reducer): ONE lane busy-waits for about
// `microseconds` (bounded to 50 ms) on `stream`, delaying everything enqueued behind it on that stream.
namespace pg {
__global__ void debug_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();            // constant 100 MHz counter
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace pg
extern "C" int pg_debug_spin(int32_t microseconds, void* stream) {
  PG_REQUIRE(microseconds >= 0 && microseconds <= 50000, "pg_debug_spin: 0..50000 us");
  PG_KLAUNCH(pg::debug_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
  PG_LAUNCH_OK("pg_debug_spin");
  return 0;
}

// ------------------------------------------------------------------------------------------------ launch tape (round 3)
// See common.h (PG_KLAUNCH).  A tape is recorded by ONE thread between pg_tape_begin and pg_tape_end while it runs one
// training iteration the usual way; pg_tape_replay re-issues every recorded enqueue (kernel launches, memsets, stream waits,
// device copies) with the argument blocks captured at record time.  The caller guarantees what a HIP graph needs as well:
// every buffer of the iteration is persistent, per-iteration scalars live in device memory (pg_dropout_mask_ctr, pg_adam_ctr,
// pg_counter_add), inputs are copied into static tensors before a replay.
#include <cstdlib>
#include <vector>
namespace pg {
struct TapeOp { void (*thunk)(void*); void* closure; void (*del)(void*); const char* where; int line; };
struct Tape { std::vector<TapeOp> ops; std::vector<hipEvent_t> events; };
static thread_local Tape* g_rec = nullptr;
Tape* tape_recording() { return g_rec; }
void tape_push(Tape* t, void (*thunk)(void*), void* closure, void (*del)(void*), const char* where, int line) {
  t->ops.push_back(TapeOp{thunk, closure, del, where, line});
}
}  // namespace pg

extern "C" int pg_tape_begin(void) {
  PG_REQUIRE(pg::g_rec == nullptr, "pg_tape_begin: this thread is already recording");
  pg::g_rec = new pg::Tape();
  return 0;
}
extern "C" int pg_tape_end(void** tape, int64_t* n_ops) {
  PG_REQUIRE(pg::g_rec != nullptr && tape != nullptr, "pg_tape_end: not recording");
  *tape = pg::g_rec;
  if (n_ops) *n_ops = (int64_t)pg::g_rec->ops.size();
  pg::g_rec = nullptr;
  return 0;
}
extern "C" int pg_tape_replay(void* tape) {
  PG_REQUIRE(tape != nullptr && pg::g_rec == nullptr, "pg_tape_replay: null tape / replay while recording");
  pg::Tape* t = static_cast<pg::Tape*>(tape);
  static const bool dbg = getenv("PG_TAPE_DEBUG") != nullptr;      // synchronise after every op and name the one that fails
  if (dbg) {
    int i = 0;
    for (const pg::TapeOp& op : t->ops) {
      fprintf(stderr, "[tape] op %d  %s:%d\n", i++, op.where, op.line);
      op.thunk(op.closure);
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) PG_FAIL(2, "pg_tape_replay: op %d (%s:%d): %s", i - 1, op.where, op.line, hipGetErrorString(e));
    }
    return 0;
  }
  for (const pg::TapeOp& op : t->ops) op.thunk(op.closure);
  PG_LAUNCH_OK("pg_tape_replay");
  return 0;
}
extern "C" int pg_tape_destroy(void* tape) {
  if (!tape) return 0;
  pg::Tape* t = static_cast<pg::Tape*>(tape);
  if (pg::g_rec == t) pg::g_rec = nullptr;
  for (const pg::TapeOp& op : t->ops) op.del(op.closure);
  for (hipEvent_t e : t->events) (void)hipEventDestroy(e);
  delete t;
  return 0;
}

// `waiter` waits for everything enqueued on `waited` so far (an event recorded now).  While recording, the event belongs to the
// tape and the record / wait pair is replayed; otherwise a cached per-thread event pool is used.
extern "C" int pg_stream_wait(void* waiter, void* waited) {
  if (waiter == waited) return 0;
  hipEvent_t ev;
  pg::Tape* t = pg::tape_recording();
  if (t != nullptr) {
    PG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    t->events.push_back(ev);
  } else {
    static thread_local std::vector<hipEvent_t> pool;
    static thread_local size_t next = 0;
    if (pool.size() < 64) {
      PG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      pool.push_back(ev);
    } else {
      ev = pool[next++ % pool.size()];
    }
  }
  hipStream_t a = (hipStream_t)waiter, b = (hipStream_t)waited;
  PG_HIP(hipEventRecord(ev, b));
  PG_HIP(hipStreamWaitEvent(a, ev, 0));
  if (t != nullptr) pg::tape_record([=]() { (void)hipEventRecord(ev, b); (void)hipStreamWaitEvent(a, ev, 0); });
  return 0;
}

extern "C" int pg_zero(void* ptr, int64_t bytes, void* stream) {
  PG_REQUIRE(ptr && bytes > 0, "pg_zero: bad arguments");
  PG_MEMSET_ASYNC(ptr, 0, (size_t)bytes, (hipStream_t)stream);
  return 0;
}
extern "C" int pg_copy(void* dst, const void* src, int64_t bytes, void* stream) {
  PG_REQUIRE(dst && src && bytes > 0, "pg_copy: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  PG_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, st));
  if (pg::tape_recording() != nullptr)
    pg::tape_record([=]() { (void)hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, st); });
  return 0;
}
namespace pg {
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* src, int R, int C, float* dst, int ld) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // dst[c][r] = src[r][c]
  if (i >= R * C) return;
  const int c = i / R, r = i - c * R;
  dst[(long)c * ld + r] = src[(long)r * C + c];
}
}  // namespace pg
// dst[c][r] = src[r][c] for an R x C row-major fp32 matrix; dst rows are `ld` floats apart (small matrices: the output
// convolution's weight viewed as [27][cin] -> [cin][32])
extern "C" int pg_transpose_f32(const float* src, int32_t R, int32_t C, float* dst, int32_t ld, void* stream) {
  PG_REQUIRE(src && dst && R > 0 && C > 0 && ld >= R, "pg_transpose_f32: bad arguments");
  PG_KLAUNCH(pg::transpose_f32_kernel, dim3((R * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, R, C, dst, ld);
  PG_LAUNCH_OK("pg_transpose_f32");
  return 0;
}
