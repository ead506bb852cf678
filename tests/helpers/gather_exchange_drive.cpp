// TEST INFRASTRUCTURE: drives ppp_gather_exchange (polishpathplanning_amd/csrc/ppp_gather.h, the send/recv pattern of
// ppp_gather_waypoints) with recording stand-ins for ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv and the
// root's device copy, and prints every call as one line.  No GPU, no RCCL.
//   usage: drive <rank> <nranks> <root> <fail_at> <count_0> ... <count_{nranks-1}>
//          drive self <fail_at> <rows>      (ppp_gather_self_loop: the one-rank pre-flight, send to self + recv from self)
//   fail_at: -1 none, 0 group_start, k > 0: the k-th send/recv returns ncclResult 5, 100: group_end returns 3, 200: the copy fails
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "ppp_gather.h"

static int g_fail_at = -1, g_xfers = 0;
static float *g_recv0 = nullptr;
static const float *g_send0 = nullptr;

static int group_start() { printf("group_start\n"); return g_fail_at == 0 ? 2 : 0; }
static int group_end() { printf("group_end\n"); return g_fail_at == 100 ? 3 : 0; }
static int do_send(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream)
{
    ++g_xfers;
    printf("send off=%td count=%zu dtype=%d peer=%d comm=%p stream=%p\n", (const float *)buf - g_send0, count, dtype, peer, comm, stream);
    return g_fail_at == g_xfers ? 5 : 0;
}
static int do_recv(void *buf, size_t count, int dtype, int peer, void *comm, void *stream)
{
    ++g_xfers;
    printf("recv off=%td count=%zu dtype=%d peer=%d comm=%p stream=%p\n", (float *)buf - g_recv0, count, dtype, peer, comm, stream);
    return g_fail_at == g_xfers ? 5 : 0;
}
static int local_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    printf("copy off=%td bytes=%zu src_off=%td stream=%p\n", (float *)dst - g_recv0, bytes, (const float *)src - g_send0, stream);
    return g_fail_at == 200 ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc == 4 && std::string(argv[1]) == "self") {
        g_fail_at = atoi(argv[2]);
        const size_t rows = (size_t)atoll(argv[3]);
        std::vector<float> send(6 * (rows + 1)), recv(6 * (rows + 1));
        g_send0 = send.data(); g_recv0 = recv.data();
        PppGatherOps ops{group_start, group_end, do_send, do_recv, local_copy};
        int nres = -1;
        const int rc = ppp_gather_self_loop(ops, rows, send.data(), recv.data(), (void *)0x1234, (void *)0x5678, &nres);
        printf("rc=%d nccl=%d\n", rc, nres);
        return 0;
    }
    if (argc < 5) return 2;
    const int rank = atoi(argv[1]), nranks = atoi(argv[2]), root = atoi(argv[3]);
    g_fail_at = atoi(argv[4]);
    if (argc != 5 + nranks) return 2;
    std::vector<size_t> counts;
    size_t total = 0;
    for (int r = 0; r < nranks; ++r) { counts.push_back((size_t)atoll(argv[5 + r])); total += counts.back(); }
    std::vector<float> send(6 * (counts[rank] + 1)), recv(6 * (total + 1));
    g_send0 = send.data(); g_recv0 = recv.data();
    PppGatherOps ops{group_start, group_end, do_send, do_recv, local_copy};
    int nres = -1;
    const int rc = ppp_gather_exchange(ops, rank, nranks, root, counts.data(), send.data(), recv.data(), (void *)0x1234, (void *)0x5678, &nres);
    printf("rc=%d nccl=%d\n", rc, nres);
    return 0;
}
