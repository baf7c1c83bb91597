/* A stand-in for librccl.so used ONLY by tests/test_comm_stub_cpu.py: the nccl* entry points the library binds
 * libtecogan_hip.so binds with dlopen (csrc/tg_comm.hip), implemented over POSIX shared memory for
 * HOST buffers, so that the id exchange / init order / collective semantics of tg_comm_* can run
 * with world sizes 2 and 8 in a container that has no GPU.  Not a product file; never shipped. */
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define CHUNK (1 << 16)
#define MAXW 16
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct {
  int arrived, generation, world, calls;
  float slot[MAXW][CHUNK];
} shared_t;
typedef struct { shared_t* sh; int world, rank; char name[64]; } comm_t;

static void name_of(const ncclUniqueId* id, char* out) {
  unsigned long long v;
  memcpy(&v, id->internal, 8);
  snprintf(out, 64, "/tgstub_%016llx", v);
}

static void barrier(comm_t* c) {
  shared_t* s = c->sh;
  int gen = __atomic_load_n(&s->generation, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&s->arrived, 1, __ATOMIC_ACQ_REL) == c->world) {
    __atomic_store_n(&s->arrived, 0, __ATOMIC_RELEASE);
    __atomic_add_fetch(&s->generation, 1, __ATOMIC_ACQ_REL);
  } else {
    struct timespec ts = {0, 200000};
    long spins = 0;
    while (__atomic_load_n(&s->generation, __ATOMIC_ACQUIRE) == gen) {
      nanosleep(&ts, NULL);
      if (++spins > 150000) { fprintf(stderr, "rccl_stub: rank %d stuck in a barrier\n", c->rank); abort(); }
    }
  }
}

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return 4;
  memset(id, 0, sizeof(*id));
  unsigned long long v = ((unsigned long long)getpid() << 32) ^ (unsigned long long)clock() ^ (unsigned long long)time(NULL);
  memcpy(id->internal, &v, 8);
  memcpy(id->internal + 8, "tg-stub-id", 10);
  return 0;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
  if (!comm || world < 1 || world > MAXW || rank < 0 || rank >= world) return 4;     /* ncclInvalidArgument */
  if (memcmp(id.internal + 8, "tg-stub-id", 10) != 0) return 5;                      /* an id this library did not issue */
  comm_t* c = (comm_t*)calloc(1, sizeof(comm_t));
  name_of(&id, c->name);
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(shared_t)) != 0) return 2;
  c->sh = (shared_t*)mmap(NULL, sizeof(shared_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->sh == MAP_FAILED) return 2;
  c->world = world; c->rank = rank;
  if (rank == 0) c->sh->world = world;
  barrier(c);                                       /* collective: every rank must call */
  if (c->sh->world != world) return 4;
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  comm_t* c = (comm_t*)comm;
  if (!c) return 4;
  barrier(c);
  munmap(c->sh, sizeof(shared_t));
  if (c->rank == 0) shm_unlink(c->name);
  free(c);
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
  comm_t* c = (comm_t*)comm;
  (void)stream;
  if (!c || dtype != 7 || op != 0) return 4;        /* ncclFloat32, ncclSum only */
  const float* s = (const float*)send; float* r = (float*)recv;
  for (size_t o = 0; o < count; o += CHUNK) {
    size_t n = count - o < CHUNK ? count - o : CHUNK;
    memcpy(c->sh->slot[c->rank], s + o, n * sizeof(float));
    barrier(c);
    for (size_t i = 0; i < n; ++i) {
      float acc = 0.f;
      for (int k = 0; k < c->world; ++k) acc += c->sh->slot[k][i];
      r[o + i] = acc;
    }
    barrier(c);
  }
  if (c->rank == 0) c->sh->calls++;
  return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
  comm_t* c = (comm_t*)comm;
  (void)stream;
  if (!c || dtype != 7) return 4;
  const float* s = (const float*)send; float* r = (float*)recv;
  for (size_t o = 0; o < count; o += CHUNK) {
    size_t n = count - o < CHUNK ? count - o : CHUNK;
    memcpy(c->sh->slot[c->rank], s + o, n * sizeof(float));
    barrier(c);
    for (int k = 0; k < c->world; ++k) memcpy(r + (size_t)k * count + o, c->sh->slot[k], n * sizeof(float));
    barrier(c);
  }
  return 0;
}

int ncclCommCount(void* comm, int* count) {
  comm_t* c = (comm_t*)comm;
  if (!c || !count) return 4;
  *count = c->sh->world;                            /* what rank 0 registered, not what this rank asked for */
  return 0;
}

int ncclCommUserRank(void* comm, int* rank) {
  comm_t* c = (comm_t*)comm;
  if (!c || !rank) return 4;
  *rank = c->rank;
  return 0;
}

const char* ncclGetErrorString(int code) {
  switch (code) { case 0: return "success"; case 2: return "stub: system error"; case 4: return "stub: invalid argument";
                  case 5: return "stub: unknown unique id"; default: return "stub: error"; }
}
