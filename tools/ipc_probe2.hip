// ipc_probe2.hip -- repeated record / wait cycles of interprocess events between two processes on one GPU, alternating the
// streams on both sides: where does hipStreamWaitEvent on an opened IPC event stop working?
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e = (x);                                                                            \
        if (e != hipSuccess) {                                                                         \
            printf("[%s] HIP error %s (%d) at line %d, iteration %d\n", who, hipGetErrorString(e), (int)e, __LINE__, it); \
            fflush(stdout);                                                                            \
            _exit(3);                                                                                  \
        }                                                                                              \
    } while (0)

__global__ void k_fill(double *p, int n, double a)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = a;
}
struct Msg {
    hipIpcMemHandle_t mem;
    hipIpcEventHandle_t ev;
};
static void xfer(int fd_w, int fd_r, const Msg &mine, Msg &theirs)
{
    if (write(fd_w, &mine, sizeof(mine)) != (ssize_t)sizeof(mine)) _exit(9);
    if (read(fd_r, &theirs, sizeof(theirs)) != (ssize_t)sizeof(theirs)) _exit(9);
}
static void token(int fd_w, int fd_r)
{
    char c = 'x';
    if (write(fd_w, &c, 1) != 1) _exit(9);
    if (read(fd_r, &c, 1) != 1) _exit(9);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int two_streams = argc > 2 ? atoi(argv[2]) : 1;
    const int n = argc > 3 ? atoi(argv[3]) : 1 << 16;
    int a2b[2], b2a[2];
    if (pipe(a2b) || pipe(b2a)) return 1;
    pid_t pid = fork();
    const bool child = pid == 0;
    const char *who = child ? "child" : "parent";
    int it = -1;
    const int fd_w = child ? b2a[1] : a2b[1], fd_r = child ? a2b[0] : b2a[0];
    CK(hipSetDevice(0));
    hipStream_t st[2];
    CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    double *mine = nullptr, *theirs = nullptr, *tmp = nullptr;
    CK(hipMalloc(&mine, (size_t)8 * n));
    CK(hipMalloc(&tmp, (size_t)8 * n));
    hipEvent_t ev_mine, ev_theirs;
    CK(hipEventCreateWithFlags(&ev_mine, hipEventDisableTiming | hipEventInterprocess));
    CK(hipEventRecord(ev_mine, st[0]));
    Msg m, o;
    CK(hipIpcGetMemHandle(&m.mem, mine));
    CK(hipIpcGetEventHandle(&m.ev, ev_mine));
    xfer(fd_w, fd_r, m, o);
    printf("[%s] opening the other's memory\n", who); fflush(stdout);
    CK(hipIpcOpenMemHandle((void **)&theirs, o.mem, hipIpcMemLazyEnablePeerAccess));
    printf("[%s] opened\n", who); fflush(stdout);
    CK(hipIpcOpenEventHandle(&ev_theirs, o.ev));
    std::vector<double> h(n);
    int bad = 0;
    for (it = 0; it < iters; ++it) {
        hipStream_t q = st[two_streams ? (it / 3) & 1 : 0];
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, q, mine, n, (double)(it + (child ? 1000 : 0)));
        CK(hipEventRecord(ev_mine, q));
        token(fd_w, fd_r);  // both have issued their record
        CK(hipStreamWaitEvent(q, ev_theirs, 0));
        CK(hipMemcpyAsync(tmp, theirs, (size_t)8 * n, hipMemcpyDeviceToDevice, q));
        CK(hipMemcpyAsync(h.data(), tmp, (size_t)8 * n, hipMemcpyDeviceToHost, q));
        CK(hipStreamSynchronize(q));
        const double want = (double)(it + (child ? 0 : 1000));
        for (int i = 0; i < n; i += 997) bad += (h[i] != want);
        token(fd_w, fd_r);  // both have read: the buffers may change
    }
    printf("[%s] %d iterations, %d wrong values\n", who, iters, bad);
    fflush(stdout);
    if (child) _exit(bad ? 2 : 0);
    int status = 0;
    waitpid(pid, &status, 0);
    return (bad == 0 && WEXITSTATUS(status) == 0) ? 0 : 1;
}
