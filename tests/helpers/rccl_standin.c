/* TEST INFRASTRUCTURE: a recording stand-in for librccl.so (ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv), loaded
   by the engine through PPP_RCCL_LIB in tests/test_gpu_parity.py::test_gather_waypoints_multi_rank_pattern_through_a_recording_rccl.
   Every call is appended to the file named by PPP_RCCL_STANDIN_LOG; PPP_RCCL_STANDIN_FAIL=recv|send makes that call
   return ncclResult 5.  Nothing is transferred: the test checks the call pattern and the root's own block. */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void logf_(const char *fmt, const void *buf, size_t count, int dtype, int peer, const void *comm, const void *stream)
{
    const char *p = getenv("PPP_RCCL_STANDIN_LOG");
    if (!p) return;
    FILE *f = fopen(p, "a");
    if (!f) return;
    fprintf(f, fmt, (unsigned long long)(size_t)buf, count, dtype, peer, (unsigned long long)(size_t)comm, (unsigned long long)(size_t)stream);
    fclose(f);
}
static int fails(const char *what) { const char *p = getenv("PPP_RCCL_STANDIN_FAIL"); return p && strcmp(p, what) == 0; }

int ncclGroupStart(void) { logf_("group_start\n", 0, 0, 0, 0, 0, 0); return 0; }
int ncclGroupEnd(void) { logf_("group_end\n", 0, 0, 0, 0, 0, 0); return 0; }
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream)
{
    logf_("send buf=%llu count=%zu dtype=%d peer=%d comm=%llu stream=%llu\n", buf, count, dtype, peer, comm, stream);
    return fails("send") ? 5 : 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, void *stream)
{
    logf_("recv buf=%llu count=%zu dtype=%d peer=%d comm=%llu stream=%llu\n", buf, count, dtype, peer, comm, stream);
    return fails("recv") ? 5 : 0;
}
