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
