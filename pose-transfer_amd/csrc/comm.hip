// Data-parallel gradient exchange behind the C ABI (pg_comm_{init,allreduce_bucket,destroy}; §8e).
// New capability — the reference is single-process (no collective anywhere in /root/reference).
//
// One process per GPU; gradients of the flat arena are SUM-all-reduced over RCCL (xGMI) in buckets that the host issues
// as soon as the backward pass has completed them, on a communication stream the CALLER owns and orders with HIP events
// (the library never synchronises).  RCCL is resolved at run time with dlopen (librccl.so.1: the copy PyTorch already
// mapped when there is one), so the library itself builds, links and loads on a box without RCCL / without a GPU.
// bf16 buckets: pg_pack_bf16 converts a finished fp32 gradient range to bf16 (RNE) into a caller-provided staging buffer,
// the all-reduce moves half the bytes (164 MB instead of 328 MB for the generator) and pg_adam_ex reads the bf16 sums
// directly (fp32 moments / master weights).
#include <dlfcn.h>

#include "common.h"

namespace pg {

typedef struct { char internal[128]; } nccl_uid_t;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
typedef int (*fn_get_uid)(nccl_uid_t*);
typedef int (*fn_init_rank)(nccl_comm_t*, int, nccl_uid_t, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_destroy)(nccl_comm_t);
typedef const char* (*fn_err)(int);
typedef int (*fn_count)(nccl_comm_t, int*);

struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_all_reduce all_reduce = nullptr;
  fn_destroy destroy = nullptr; fn_err err = nullptr; fn_count count = nullptr, user_rank = nullptr;
};

static Rccl& rccl() { static Rccl r; return r; }

static int load_rccl() {
  Rccl& r = rccl();
  if (r.h) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {          // prefer an already-mapped copy (PyTorch's), then the system one
    r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (r.h) break;
  }
  for (int i = 0; i < 3 && !r.h; ++i) r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!r.h) PG_FAIL(4, "pg_comm: cannot load librccl (%s)", dlerror());
  r.get_uid = (fn_get_uid)dlsym(r.h, "ncclGetUniqueId");
  r.init_rank = (fn_init_rank)dlsym(r.h, "ncclCommInitRank");
  r.all_reduce = (fn_all_reduce)dlsym(r.h, "ncclAllReduce");
  r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy");
  r.err = (fn_err)dlsym(r.h, "ncclGetErrorString");
  r.count = (fn_count)dlsym(r.h, "ncclCommCount");
  r.user_rank = (fn_count)dlsym(r.h, "ncclCommUserRank");
  if (!r.get_uid || !r.init_rank || !r.all_reduce || !r.destroy) { r.h = nullptr; PG_FAIL(4, "pg_comm: librccl lacks an entry point"); }
  return 0;
}

struct Comm { nccl_comm_t c; int rank, world; };

#define PG_NCCL(call, what)                                                                        \
  do {                                                                                             \
    int rc__ = (call);                                                                             \
    if (rc__ != 0) PG_FAIL(5, "%s: %s", what, rccl().err ? rccl().err(rc__) : "RCCL error");       \
  } while (0)

__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* src, unsigned* dst, long n2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
    const float2 v = reinterpret_cast<const float2*>(src)[i];
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y));
    dst[i] = r;
  }
}

}  // namespace pg

using namespace pg;

extern "C" int pg_comm_unique_id(void* out128) {
  PG_REQUIRE(out128 != nullptr, "pg_comm_unique_id: null buffer");
  if (int rc = load_rccl()) return rc;
  nccl_uid_t id;
  PG_NCCL(rccl().get_uid(&id), "ncclGetUniqueId");
  memcpy(out128, &id, sizeof(id));
  return 0;
}

extern "C" int pg_comm_init(const void* unique_id128, int32_t rank, int32_t world, void** comm) {
  PG_REQUIRE(unique_id128 && comm && world >= 1 && rank >= 0 && rank < world, "pg_comm_init: bad arguments");
  if (int rc = load_rccl()) return rc;
  nccl_uid_t id;
  memcpy(&id, unique_id128, sizeof(id));
  Comm* c = new Comm{nullptr, rank, world};
  int rc = rccl().init_rank(&c->c, world, id, rank);
  if (rc != 0) { delete c; PG_FAIL(5, "ncclCommInitRank: %s", rccl().err ? rccl().err(rc) : "RCCL error"); }
  *comm = c;
  return 0;
}

// in-place SUM all-reduce of `count` elements (dtype 0 = fp32, 1 = bf16) on `stream`
extern "C" int pg_comm_allreduce_bucket(void* comm, void* buf, int64_t count, int32_t dtype, void* stream) {
  PG_REQUIRE(comm && buf && count > 0 && (dtype == 0 || dtype == 1), "pg_comm_allreduce_bucket: bad arguments");
  Comm* c = reinterpret_cast<Comm*>(comm);
  PG_NCCL(rccl().all_reduce(buf, buf, (size_t)count, dtype == 0 ? 7 /* ncclFloat32 */ : 9 /* ncclBfloat16 */, 0 /* ncclSum */,
                            c->c, (hipStream_t)stream), "ncclAllReduce");
  return 0;
}

extern "C" int pg_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (c->c && rccl().destroy) rccl().destroy(c->c);
  delete c;
  return 0;
}

// rank / size as RCCL reports them for this communicator (not the values pg_comm_init was called with)
extern "C" int pg_comm_ranks(void* comm, int32_t* rank, int32_t* world) {
  PG_REQUIRE(comm && rank && world, "pg_comm_ranks: bad arguments");
  Comm* c = reinterpret_cast<Comm*>(comm);
  PG_REQUIRE(rccl().count && rccl().user_rank, "pg_comm_ranks: librccl lacks ncclCommCount / ncclCommUserRank");
  int r = -1, w = -1;
  PG_NCCL(rccl().count(c->c, &w), "ncclCommCount");
  PG_NCCL(rccl().user_rank(c->c, &r), "ncclCommUserRank");
  *rank = r; *world = w;
  return 0;
}

extern "C" int pg_pack_bf16(const float* src, void* dst, int64_t n, void* stream) {
  PG_REQUIRE(src && dst && n > 0 && n % 2 == 0 && ((size_t)src & 7) == 0 && ((size_t)dst & 3) == 0,
             "pg_pack_bf16: need an even count and 8 / 4-byte aligned pointers");
  long blocks = (n / 2 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  PG_KLAUNCH(pack_bf16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src, (unsigned*)dst, (long)(n / 2));
  PG_LAUNCH_OK("pg_pack_bf16");
  return 0;
}
