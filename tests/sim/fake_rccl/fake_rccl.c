/* TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so on the CPU tier: the five entry points
 * cvxpygen_amd/sharding.py::RcclGather binds, with point-to-point transfers carried by files in /dev/shm
 * (the emulator's "device memory" is host memory of each rank's process) and every call appended to the
 * log named by CPG_FAKE_RCCL_LOG, so that tests can check the send / receive schedule of the gather.
 * Transfers are executed at ncclGroupEnd in the order they were queued, sends first. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <stdint.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct { int rank, world; char tag[40]; unsigned seq_to[64], seq_from[64]; } fake_comm;
typedef struct { int send; void *buf; size_t bytes; int peer; fake_comm *c; } op_t;
static op_t ops[4096]; static int n_ops = 0, in_group = 0;

static void logf_(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void logf_(const char *fmt, ...) {
    const char *p = getenv("CPG_FAKE_RCCL_LOG"); if (!p) return;
    FILE *f = fopen(p, "a"); if (!f) return;
    va_list ap; va_start(ap, fmt); vfprintf(f, fmt, ap); va_end(ap); fclose(f);
}
const char *ncclGetErrorString(int rc) { return rc ? "fake rccl error" : "ok"; }
int ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal + 1, 100, "fk%d_%ld", (int)getpid(), (long)random());   /* byte 0 is NUL on purpose */
    id->internal[127] = 7;
    logf_("getuid pid=%d\n", (int)getpid()); return 0; }
int ncclCommInitRank(void **comm, int world, ncclUniqueId id, int rank) {
    if (id.internal[127] != 7) return 1;                      /* the id must arrive whole */
    { const char *fr = getenv("CPG_FAKE_RCCL_FAIL_RANK"); if (fr && atoi(fr) == rank) return 2; }   /* test hook: this rank cannot create its communicator */
    fake_comm *c = (fake_comm *)calloc(1, sizeof(fake_comm)); c->rank = rank; c->world = world;
    snprintf(c->tag, sizeof(c->tag), "%s", id.internal + 1); *comm = c;
    logf_("init rank=%d world=%d\n", rank, world); return 0; }
int ncclCommDestroy(void *comm) { free(comm); return 0; }
int ncclGroupStart(void) { in_group = 1; return 0; }
static int run(op_t *o) {
    char path[256]; fake_comm *c = o->c;
    if (o->send) {
        snprintf(path, sizeof(path), "/dev/shm/fakerccl_%s_%d_%d_%u", c->tag, c->rank, o->peer, c->seq_to[o->peer]++);
        char tmp[300]; snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE *f = fopen(tmp, "wb"); if (!f) return 1; fwrite(o->buf, 1, o->bytes, f); fclose(f); rename(tmp, path);
    } else {
        snprintf(path, sizeof(path), "/dev/shm/fakerccl_%s_%d_%d_%u", c->tag, o->peer, c->rank, c->seq_from[o->peer]++);
        for (int t = 0; access(path, F_OK) != 0; t++) { if (t > 600000) return 1; usleep(100); }
        FILE *f = fopen(path, "rb"); if (!f) return 1; size_t got = fread(o->buf, 1, o->bytes, f); fclose(f); unlink(path);
        if (got != o->bytes) return 1;
    }
    return 0;
}
static int queue(int send, void *buf, size_t count, int dtype, int peer, void *comm, void *stream) {
    (void)stream;
    fake_comm *c = (fake_comm *)comm;
    if (dtype != 1 || n_ops >= 4096) return 1;               /* the gather moves bytes (ncclUint8) */
    logf_("%s rank=%d peer=%d bytes=%zu\n", send ? "send" : "recv", c->rank, peer, count);
    op_t o = {send, buf, count, peer, c}; 
    if (!in_group) return run(&o);
    ops[n_ops++] = o; return 0; }
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream) { return queue(1, (void *)buf, count, dtype, peer, comm, stream); }
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, void *stream) { return queue(0, buf, count, dtype, peer, comm, stream); }
int ncclGroupEnd(void) {
    int rc = 0;
    for (int k = 0; k < n_ops; k++) if (ops[k].send) rc |= run(&ops[k]);
    for (int k = 0; k < n_ops; k++) if (!ops[k].send) rc |= run(&ops[k]);
    n_ops = 0; in_group = 0; logf_("groupend\n"); return rc; }
