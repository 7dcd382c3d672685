/* TEST INFRASTRUCTURE: a stand-in for librccl.so that implements the four entry points okvis_amd/csrc/dist_capi.hip binds
 * (ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy, + ncclGetErrorString) over a Unix-domain socket, so that
 * okvis_ba_gather_records forms a communicator of MORE THAN ONE rank on a box with one GPU (or none): the rank order of the
 * gathered records, the padding records, the id-file hand-over and its time-outs are exercised before the first real 8-GPU run
 * (tests/test_rccl_stub.py, tests/test_gpu_rccl_stub.py; OKVIS_BA_RCCL_LIB selects it).  Star topology: rank 0 listens on the
 * socket whose path IS the unique id, the other ranks connect and say who they are; an all-gather is "everybody sends to rank 0,
 * rank 0 sends the concatenation back".  Device buffers are staged through the host with hipMemcpy. */
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct Comm {
  int rank, world;
  int listen_fd;
  int* fd; /* rank 0: fd[r] of every other rank; others: fd[0] = the connection to rank 0 */
  char path[108];
} Comm;
typedef Comm* ncclComm_t;

static int send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) return -1;
    c += k;
    n -= (size_t)k;
  }
  return 0;
}
static int recv_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    ssize_t k = recv(fd, c, n, 0);
    if (k <= 0) return -1;
    c += k;
    n -= (size_t)k;
  }
  return 0;
}
static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static double init_timeout(void) {
  const char* e = getenv("STUB_RCCL_TIMEOUT_S");
  return e ? atof(e) : 20.0;
}

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/tmp/okvis_stub_rccl_%d_%lx", (int)getpid(), (unsigned long)(now_s() * 1e6));
  return 0;
}

int ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  Comm* c = (Comm*)calloc(1, sizeof(Comm));
  if (!c) return 1;
  c->rank = rank;
  c->world = world;
  c->listen_fd = -1;
  c->fd = (int*)calloc((size_t)(world > 0 ? world : 1), sizeof(int));
  for (int r = 0; r < world; ++r) c->fd[r] = -1;
  strncpy(c->path, id.internal, sizeof(c->path) - 1);
  struct sockaddr_un a;
  memset(&a, 0, sizeof(a));
  a.sun_family = AF_UNIX;
  strncpy(a.sun_path, c->path, sizeof(a.sun_path) - 1);
  const double t0 = now_s(), limit = init_timeout();
  if (world == 1) {
    *out = c;
    return 0;
  }
  if (rank == 0) {
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    unlink(c->path);
    if (c->listen_fd < 0 || bind(c->listen_fd, (struct sockaddr*)&a, sizeof(a)) != 0 || listen(c->listen_fd, world) != 0) return 2;
    struct timeval tv = {1, 0};
    setsockopt(c->listen_fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    for (int got = 0; got < world - 1;) {
      if (now_s() - t0 > limit) return 3; /* a rank never came */
      int fd = accept(c->listen_fd, NULL, NULL);
      if (fd < 0) continue;
      int r = -1;
      if (recv_all(fd, &r, sizeof(r)) != 0 || r <= 0 || r >= world || c->fd[r] >= 0) {
        close(fd);
        return 4;
      }
      c->fd[r] = fd;
      ++got;
    }
    for (int r = 1; r < world; ++r) { /* everybody is here: release them */
      int ok = 1;
      if (send_all(c->fd[r], &ok, sizeof(ok)) != 0) return 5;
    }
  } else {
    for (;;) {
      int fd = socket(AF_UNIX, SOCK_STREAM, 0);
      if (fd >= 0 && connect(fd, (struct sockaddr*)&a, sizeof(a)) == 0) {
        c->fd[0] = fd;
        break;
      }
      if (fd >= 0) close(fd);
      if (now_s() - t0 > limit) return 3;
      usleep(2000);
    }
    int ok = 0;
    if (send_all(c->fd[0], &rank, sizeof(rank)) != 0 || recv_all(c->fd[0], &ok, sizeof(ok)) != 0 || ok != 1) return 5;
  }
  *out = c;
  return 0;
}

/* dtype 1 = ncclUint8 (the only one dist_capi.hip uses): count = bytes */
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t c, hipStream_t st) {
  if (dtype != 1 || !c) return 1;
  const size_t n = count, total = n * (size_t)c->world;
  char* h = (char*)malloc(total ? total : 1);
  if (!h) return 1;
  int rc = 0;
  if (hipStreamSynchronize(st) != hipSuccess) rc = 6;
  if (!rc && hipMemcpy(h + n * (size_t)c->rank, send, n, hipMemcpyDeviceToHost) != hipSuccess) rc = 6;
  if (!rc && c->world > 1) {
    if (c->rank == 0) {
      for (int r = 1; r < c->world && !rc; ++r)
        if (recv_all(c->fd[r], h + n * (size_t)r, n) != 0) rc = 7;
      for (int r = 1; r < c->world && !rc; ++r)
        if (send_all(c->fd[r], h, total) != 0) rc = 7;
    } else {
      if (send_all(c->fd[0], h + n * (size_t)c->rank, n) != 0 || recv_all(c->fd[0], h, total) != 0) rc = 7;
    }
  }
  if (!rc && hipMemcpy(recv, h, total, hipMemcpyHostToDevice) != hipSuccess) rc = 6;
  free(h);
  return rc;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return 0;
  for (int r = 0; r < c->world; ++r)
    if (c->fd[r] >= 0) close(c->fd[r]);
  if (c->listen_fd >= 0) {
    close(c->listen_fd);
    unlink(c->path);
  }
  free(c->fd);
  free(c);
  return 0;
}

const char* ncclGetErrorString(int e) {
  switch (e) {
    case 0: return "ok";
    case 3: return "stub rccl: a rank did not arrive in time";
    case 6: return "stub rccl: HIP copy failed";
    case 7: return "stub rccl: socket transfer failed";
    default: return "stub rccl: error";
  }
}
