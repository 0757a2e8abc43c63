// libposegan_hip: error reporting, version, HIP-event timing helpers.
#include "common.h"

namespace pg {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int& last_info() {
  static thread_local int v = 0;
  return v;
}
}  // namespace pg

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
  hipLaunchKernelGGL(pg::debug_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
  PG_LAUNCH_OK("pg_debug_spin");
  return 0;
}
