/* A stand-in for librccl.so that does no communication: every collective of a one-rank communicator
 * is the identity, so ncclAllReduce only writes a line (count, data type, operation) to the file named
 * by RCCL_STUB_LOG.  tests/test_gpu_parity.py puts it ahead of the real library on LD_LIBRARY_PATH to
 * read the SEQUENCE of collectives rt_allreduce issues: ranks that issue different sequences deadlock. */
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>

static void note(const char* line, size_t count, int type, int op)
{
    const char* path = getenv("RCCL_STUB_LOG");
    if (path == NULL) return;
    FILE* f = fopen(path, "a");
    if (f == NULL) return;
    fprintf(f, "%s %zu %d %d\n", line, count, type, op);
    fclose(f);
}

int ncclAllReduce(const void* send, void* recv, size_t count, int type, int op, void* comm, void* stream)
{
    (void)comm;
    (void)stream;
    note("allreduce", count, type, op);
    return send == recv ? 0 : 5; /* in place only; 5 = ncclInvalidUsage */
}

int ncclCommInitAll(void** comms, int n, const int* devices)
{
    static int tokens[64];
    (void)devices;
    if (n != 1) return 5; /* one rank is all this stand-in can do */
    comms[0] = &tokens[0];
    note("comm_init_all", (size_t)n, 0, 0);
    return 0;
}

int ncclCommDestroy(void* comm)
{
    (void)comm;
    return 0;
}

const char* ncclGetErrorString(int code)
{
    return code == 0 ? "no error" : "stand-in RCCL: unsupported use";
}
